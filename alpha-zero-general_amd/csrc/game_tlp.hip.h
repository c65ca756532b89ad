// game_tlp.hip.h -- The Little Prince ("Make me a planet") env step on the device plugin interface:
// thelittleprince/TLPLogicNumba.py (Board :95-412), 3-5 players (a fifth player's terminal result rides in RecHdr.Qs, forest.hip.h rec_es).
//
// State int8 [18 n + 1][15] (copy_state :147-156): row 0 = round_and_state (col 0 round, col 1 current player, col 2 bitfield of
// who can still play this turn (player p = bit 128 >> p), cols 3..12 bitfield of the 80 cards still in the deck, MSB first);
// rows 1..n market; rows n+1..2n players_score (one column per attribute, the FACE_DOWN column carries the volcano penalty);
// rows 2n+1.. players_cards, 16 slots per player.  A card row = 14 attribute counts + the card type (25 centre, 50 uphill edge,
// 75 downhill edge, 100 + character for corners).  Action = card * n + player_delta (:20-33).
//
// STOCHASTIC = true: make_move ignores random_seed; the market refill (:366-392) draws 1 + n uniforms through
// my_random_choice_and_normalize (:46-50) = searchsorted(cumsum(mask / mask.sum()), u, 'right'), inside MCTS simulations too, so
// no edge of this game is ever memoised (kernels.hip.h k_select replays the env step with the tree's counter stream).
// RANDOM_SYM = true: get_symmetries (:177-272) shuffles players / market cards / planet slots with np.random.shuffle and drops
// duplicate states; the RNG contract defines shuffle as Fisher-Yates from the top (j = floor(u (i + 1)), i = len-1 .. 1) on the
// stream the caller names (include/azg.h azg_env_symmetries_ex).
//
// The rules are dependent scalar work on a few hundred bytes: make_move / init run on lane 0 over the LDS state, the valid-move
// mask is one action per lane, swap_players and the symmetric forms are row maps applied by all lanes.
#pragma once
#include "azg_common.hip.h"
#include "tlp_tables.h"

namespace azg {

template <int NP>
struct TLPDev {
    static constexpr int P = NP;
    static constexpr int ROWS = 18 * NP + 1, COLS = 15;
    static constexpr int S = ROWS * COLS;
    static constexpr int SP = RoundUp16<S>::value;
    static constexpr int A = NP * NP;
    static constexpr int AW = 1;
    static constexpr bool STOCHASTIC = true;
    static constexpr bool RANDOM_SYM = true;      // symmetric forms are built by lane 0 (k_env_symmetries_built) and draw randomness
    static constexpr bool SYM_DEDUP = true;       // _add_to_list_no_duplicate :239-244
    enum { FACE_DOWN, BAOBAB, VOLCANO, SUNSET, ROSE, LAMPPOST, BOX, BIG_STAR, FOX, ELEPHANT, SNAKE, SHEEP_WHITE, SHEEP_GREY,
           SHEEP_BROWN, CARD_TYPE };
    enum { NONE, VAIN_MAN, GEOGRAPHER, ASTRONOMER, KING, LAMPLIGHTER, HUNTER, DRUNKARD, BUSINESSMAN_W, BUSINESSMAN_G, BUSINESSMAN_B,
           GARDENER, TURKISH, LITTLE_PRINCE };
    enum { R_MARKET = 1, R_SCORE = NP + 1, R_CARDS = 2 * NP + 1 };

    __device__ static __forceinline__ int8_t* row(int8_t* st, int r) { return st + r * COLS; }
    __device__ static __forceinline__ const int8_t* row(const int8_t* st, int r) { return st + r * COLS; }
    __device__ static __forceinline__ int ctype(const int8_t* st, int r) { return st[r * COLS + CARD_TYPE]; }
    __device__ static __forceinline__ bool can_play(const int8_t* st, int p) { return (((uint8_t)st[2]) >> (7 - p)) & 1; }
    // slots_in_planet :76-88, 4 bits per slot: centre {5,6,9,10}, uphill {1,7,8,14}, downhill {2,4,11,13}, corner {0,3,12,15}
    __device__ static __forceinline__ int slot(int grp, int k) {
        const uint32_t tab = grp == 0 ? 0xA965u : (grp == 1 ? 0xE871u : (grp == 2 ? 0xDB42u : 0xFC30u));
        return (int)((tab >> (4 * k)) & 15u);
    }

    __device__ static int get_score(const int8_t* st, int p) {                // :104-105
        int t = 0;
#pragma unroll
        for (int k = 0; k < COLS; k++) t += st[(R_SCORE + p) * COLS + k];
        return t;
    }
    __device__ static __forceinline__ int get_round(const int8_t* st) { return st[0]; }
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return (int)(uint8_t)st[0]; }     // the round only grows
    __device__ static __forceinline__ bool move_uses_seed(int) { return true; }   // (unused: STOCHASTIC edges are never memoised)

    // my_random_choice_and_normalize :46-50 over the set bits of `mask` (bit i = entry i of `len`)
    __device__ static int choice(Rng& rng, uint32_t mask, int len) {
        const int k = __popc(mask);
        const double u = rng.u01();
        double c = 0.0;
        for (int i = 0; i < len; i++) {
            c += (((mask >> i) & 1u) ? 1.0 : 0.0) / (double)k;
            if (c > u) return i;
        }
        return len - 1;
    }

    __device__ static void fill_market_if_needed(int8_t* st, Rng& rng) {      // :366-392
        for (int i = 0; i < NP; i++)
            if (ctype(st, R_MARKET + i) != 0) return;
        bool all_full = true;
        for (int i = 0; i < 16 * NP; i++)
            if (!(ctype(st, R_CARDS + i) > 0)) { all_full = false; break; }
        if (all_full) return;
        const uint32_t room = (ctype(st, R_CARDS + 10) == 0 ? 1u : 0u) | (ctype(st, R_CARDS + 14) == 0 ? 2u : 0u) |
                              (ctype(st, R_CARDS + 13) == 0 ? 4u : 0u) | (ctype(st, R_CARDS + 15) == 0 ? 8u : 0u);
        const int grp = choice(rng, room, 4);
        for (int i = 0; i < NP; i++) {
            uint32_t avail = 0;
            for (int c = 0; c < 20; c++) {
                const int bit = 20 * grp + c;
                avail |= (uint32_t)((((uint8_t)st[3 + (bit >> 3)]) >> (7 - (bit & 7))) & 1) << c;
            }
            const int idx = choice(rng, avail, 20);
            const uint32_t w = TLP_CARD_ATTR[20 * grp + idx];
            int8_t* m = row(st, R_MARKET + i);
            for (int k = 0; k < 14; k++) m[k] = (int8_t)((w >> (2 * k)) & 3u);
            m[CARD_TYPE] = (int8_t)(grp == 3 ? 100 + TLP_CORNER_CHAR[idx] : 25 * (grp + 1));
            const int bit = 20 * grp + idx;
            st[3 + (bit >> 3)] = (int8_t)(((uint8_t)st[3 + (bit >> 3)]) & ~(128u >> (bit & 7)));
        }
        st[2] = (int8_t)(uint8_t)(0xFF00u >> NP);
    }

    __device__ static void take_card(int8_t* st, int i, int p) {               // :283-300
        const int ct = ctype(st, R_MARKET + i);
        const int grp = ct == 25 ? 0 : (ct == 50 ? 1 : (ct == 75 ? 2 : 3));
        int best = 16 * NP - 1;                                                // players_cards[-1] as written; unreachable in play
        for (int k = 0; k < 4; k++)
            if (ctype(st, R_CARDS + 16 * p + slot(grp, k)) == 0) { best = 16 * p + slot(grp, k); break; }
        int8_t* dst = row(st, R_CARDS + best);
        int8_t* src = row(st, R_MARKET + i);
        for (int k = 0; k < COLS; k++) { dst[k] = src[k]; src[k] = 0; }
        int baobabs = 0;
        for (int c = 0; c < 16; c++) baobabs += st[(R_CARDS + 16 * p + c) * COLS + BAOBAB];
        if (baobabs >= 3)
            for (int c = 0; c < 16; c++) {
                int8_t* r = row(st, R_CARDS + 16 * p + c);
                if (r[BAOBAB] >= 1) {
                    for (int k = 0; k < CARD_TYPE; k++) r[k] = 0;
                    r[FACE_DOWN] = 1;
                }
            }
    }

    __device__ static void update_score(int8_t* st, int p) {                   // :303-364
        int sum[COLS];
#pragma unroll
        for (int k = 0; k < COLS; k++) sum[k] = 0;
        for (int c = 0; c < 16; c++) {
            const int8_t* r = row((const int8_t*)st, R_CARDS + 16 * p + c);
#pragma unroll
            for (int k = 0; k < COLS; k++) sum[k] += r[k];
        }
        int8_t* sc = row(st, R_SCORE + p);
        for (int k = 0; k < COLS; k++) sc[k] = 0;
        for (int k4 = 0; k4 < 4; k4++) {
            const int ct = ctype(st, R_CARDS + 16 * p + slot(3, k4));
            const int ch = ct - 100 > 0 ? ct - 100 : 0;
            if (ch == NONE) continue;
            switch (ch) {
            case VAIN_MAN: sc[SNAKE] = (int8_t)(sc[SNAKE] + 4 * sum[SNAKE]); break;
            case GEOGRAPHER:
                for (int c = 0; c < 16; c++)
                    if (c != 0 && c != 3 && c != 12 && c != 15 && st[(R_CARDS + 16 * p + c) * COLS + VOLCANO] == 0)
                        sc[VOLCANO] = (int8_t)(sc[VOLCANO] + 1);
                break;
            case ASTRONOMER: sc[SUNSET] = (int8_t)(sc[SUNSET] + 2 * sum[SUNSET]); break;
            case KING: sc[ROSE] = (int8_t)(sc[ROSE] + (sum[ROSE] == 1 ? 14 : (sum[ROSE] == 2 ? 7 : 0))); break;   // [0, 14, 7, 0][min(n, 3)]
            case LAMPLIGHTER: sc[LAMPPOST] = (int8_t)(sc[LAMPPOST] + sum[LAMPPOST]); break;
            case HUNTER:
                sc[SNAKE] = (int8_t)(sc[SNAKE] + (sum[SNAKE] > 0 ? 3 : 0));
                sc[ELEPHANT] = (int8_t)(sc[ELEPHANT] + (sum[ELEPHANT] > 0 ? 3 : 0));
                if (sum[SHEEP_WHITE] > 0) sc[SHEEP_WHITE] = (int8_t)(sc[SHEEP_WHITE] + 3);
                else if (sum[SHEEP_GREY] > 0) sc[SHEEP_GREY] = (int8_t)(sc[SHEEP_GREY] + 3);
                else if (sum[SHEEP_BROWN] > 0) sc[SHEEP_BROWN] = (int8_t)(sc[SHEEP_BROWN] + 3);
                break;
            case DRUNKARD: sc[BAOBAB] = (int8_t)(sc[BAOBAB] + 3 * sum[FACE_DOWN]); break;
            case BUSINESSMAN_W: sc[SHEEP_WHITE] = (int8_t)(sc[SHEEP_WHITE] + 2 * sum[SHEEP_WHITE]); break;
            case BUSINESSMAN_G: sc[SHEEP_GREY] = (int8_t)(sc[SHEEP_GREY] + 3 * sum[SHEEP_GREY]); break;
            case BUSINESSMAN_B: sc[SHEEP_BROWN] = (int8_t)(sc[SHEEP_BROWN] + 5 * sum[SHEEP_BROWN]); break;
            case GARDENER: sc[BAOBAB] = (int8_t)(sc[BAOBAB] + 7 * sum[BAOBAB]); break;
            case TURKISH: sc[BIG_STAR] = (int8_t)(sc[BIG_STAR] + sum[BIG_STAR]); break;
            case LITTLE_PRINCE:
                if (sum[SHEEP_WHITE] > 0) sc[SHEEP_WHITE] = (int8_t)(sc[SHEEP_WHITE] + 3);
                if (sum[SHEEP_GREY] > 0) sc[SHEEP_GREY] = (int8_t)(sc[SHEEP_GREY] + 3);
                if (sum[SHEEP_BROWN] > 0) sc[SHEEP_BROWN] = (int8_t)(sc[SHEEP_BROWN] + 3);
                sc[BOX] = (int8_t)(sc[BOX] + sum[BOX]);
                break;
            default: break;
            }
            // volcano penalty of every player, kept in the FACE_DOWN column of the score rows (:351-357)
            int nb[NP], mx = -1000;
            for (int q = 0; q < NP; q++) {
                nb[q] = 0;
                for (int c = 0; c < 16; c++) nb[q] += st[(R_CARDS + 16 * q + c) * COLS + VOLCANO];
                mx = nb[q] > mx ? nb[q] : mx;
            }
            for (int q = 0; q < NP; q++) st[(R_SCORE + q) * COLS + FACE_DOWN] = (int8_t)(nb[q] == mx ? -mx : 0);
        }
    }

    // Board.valid_moves restricted to one action (:119-133)
    __device__ static bool valid_action(const int8_t* st, int a, int player) {
        const int card = a / NP, d = a % NP, p = (player + d) % NP;
        const uint32_t bits = ((uint8_t)st[2]) & (0xFF00u >> NP) & 0xFFu;
        const uint32_t others = bits & ~(128u >> player);
        const bool who = p == player ? others == 0 : ((others >> (7 - p)) & 1u) != 0;
        return who && ctype(st, R_MARKET + card) != 0;
    }
    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
        const uint64_t m = __ballot(l < A && valid_action(st, l < A ? l : 0, player));
        if (l == 0) mask_lds[0] = m;
    }

    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        return lane0_make_move<TLPDev<NP>>(st, move, player, seed, rng);
    }
    // Board.make_move :135-145 -- lane 0 only
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        (void)seed;
        const int card = move / NP, next = (player + move % NP) % NP;
        take_card(st, card, player);
        update_score(st, player);
        fill_market_if_needed(st, rng);
        st[2] = (int8_t)(((uint8_t)st[2]) & ~(128u >> player));               // _player_cant_play_again_this_turn :407-410
        st[0] = (int8_t)(st[0] + 1);
        st[1] = (int8_t)next;
        return next;
    }

    // Board.check_end_game :158-166 (uniform)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        if (st[0] < 16 * NP) {
#pragma unroll
            for (int p = 0; p < NP; p++) out[p] = 0.f;
            return false;
        }
        int sc[NP], mx = -1000, cnt = 0;
#pragma unroll
        for (int p = 0; p < NP; p++) { sc[p] = (int)(int8_t)get_score(st, p); mx = sc[p] > mx ? sc[p] : mx; }
#pragma unroll
        for (int p = 0; p < NP; p++) cnt += sc[p] == mx;
#pragma unroll
        for (int p = 0; p < NP; p++) out[p] = sc[p] == mx ? (cnt == 1 ? 1.f : 0.01f) : -1.f;
        return true;
    }

    // Board.swap_players :170-182: score row / planet of player i come from player (i + k) mod n; current player and the
    // who-can-play bits follow
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        for (int i = lane_id(); i < S; i += 64) tmp[i] = st[i];
        wave_sync();
        for (int i = lane_id(); i < S; i += 64) {
            const int r = i / COLS, c = i - r * COLS;
            int src = r;
            if (r >= R_CARDS) src = R_CARDS + ((r - R_CARDS) + 16 * k) % (16 * NP);
            else if (r >= R_SCORE) src = R_SCORE + ((r - R_SCORE) + k) % NP;
            int8_t v = tmp[src * COLS + c];
            if (i == 1) v = (int8_t)((tmp[1] - k + NP) % NP);
            if (i == 2) {
                uint32_t b = 0;
                for (int p = 0; p < NP; p++) b |= ((((uint8_t)tmp[2]) >> (7 - (p + k) % NP)) & 1u) ? (128u >> p) : 0u;
                v = (int8_t)(uint8_t)b;
            }
            st[i] = v;
        }
        wave_sync();
    }

    // init_game :107-117 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        st[2] = (int8_t)(uint8_t)(0xFF00u >> NP);
        for (int i = 3; i < 13; i++) st[i] = -1;
        fill_market_if_needed(st, rng);
    }

    // ---- get_symmetries :177-272.  Candidate 0 is the identity; 1..n shuffle the players who already played / who have not
    // (two shuffles, the current player stays); n+1..2n shuffle the market cards and, per player and card type, the planet slots
    // (pi and valids come back unpermuted for these, as written :223,237).  Every form is a row map of the input because the
    // chained shuffles act on disjoint rows.  lane 0 draws and fills row_src[ROWS] / act_src[A]; the kernel applies them and
    // drops states equal to a form already kept (_add_to_list_no_duplicate :239-244).
    static constexpr int NSYM_CAND = 2 * NP + 1;
    __device__ static void shuffle(Rng& rng, int* a, int len) {
        for (int i = len - 1; i > 0; i--) {
            int j = (int)(rng.u01() * (double)(i + 1));
            j = j > i ? i : j;
            const int t = a[i]; a[i] = a[j]; a[j] = t;
        }
    }
    __device__ static void sym_random_maps(const int8_t* st, int cand, int16_t* row_src, int16_t* act_src, Rng& rng) {
        for (int r = 0; r < ROWS; r++) row_src[r] = (int16_t)r;
        for (int a = 0; a < A; a++) act_src[a] = (int16_t)a;
        int list[16], sh[16];
        if (cand <= NP) {
            const int cur = st[1];
            for (int pass = 0; pass < 2; pass++) {                             // players who played, then those who have not
                int len = 0;
                for (int i = 0; i < NP; i++)
                    if (i != cur && can_play(st, i) == (pass == 1)) { list[len] = i; sh[len] = i; len++; }
                shuffle(rng, sh, len);
                for (int i = 0; i < len; i++) {
                    const int o = list[i], w = sh[i];
                    row_src[R_SCORE + w] = (int16_t)(R_SCORE + o);
                    for (int c = 0; c < 16; c++) row_src[R_CARDS + 16 * w + c] = (int16_t)(R_CARDS + 16 * o + c);
                    for (int c = 0; c < NP; c++) act_src[c * NP + w] = (int16_t)(c * NP + o);
                }
            }
            return;
        }
        int len = 0;
        for (int i = 0; i < NP; i++)
            if (ctype(st, R_MARKET + i) != 0) { list[len] = i; sh[len] = i; len++; }
        shuffle(rng, sh, len);
        for (int i = 0; i < len; i++) row_src[R_MARKET + sh[i]] = (int16_t)(R_MARKET + list[i]);
        for (int p = 0; p < NP; p++)
            for (int ct = 1; ct <= 4; ct++) {
                len = 0;
                for (int i = 0; i < 16; i++)
                    if (ctype(st, R_CARDS + 16 * p + i) / 25 == ct) { list[len] = i; sh[len] = i; len++; }
                shuffle(rng, sh, len);
                for (int i = 0; i < len; i++) row_src[R_CARDS + 16 * p + sh[i]] = (int16_t)(R_CARDS + 16 * p + list[i]);
            }
    }
    // lane 0: `cand` (a copy of the input state) becomes form c
    __device__ static bool sym_build(const int8_t* st, int c, int8_t* cand, int16_t* act_src, Rng& rng, const uint8_t* valids) {
        (void)valids;
        int16_t row_src[ROWS];
        if (c == 0) {
            for (int a = 0; a < A; a++) act_src[a] = (int16_t)a;
            return true;
        }
        sym_random_maps(st, c, row_src, act_src, rng);
        for (int r = 0; r < ROWS; r++)
            if (row_src[r] != r)
                for (int k = 0; k < COLS; k++) cand[r * COLS + k] = st[row_src[r] * COLS + k];
        return true;
    }
    // (the deterministic interface is unused: k_env_symmetries takes the RANDOM_SYM path)
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int c) { return c == 0; }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int, int i) { return st[i]; }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int, int a) { return a; }
};

}  // namespace azg
