// game_botanik.hip.h -- Botanik env step on the device plugin interface: botanik/BotanikLogicNumba.py
// (Board :143-646, helpers :648-787) with the shipped constants (BotanikConstants.py: MACHINE_SIZE = 7, two players).
//
// State int8 [66][5][7] (copy_state :165-189), a "card" = 7 bytes {colour, flowers, type, N, E, S, W}:
//   block 0 misc: byte 0 round, 1 status within the round, 2 main player, 3..4 = 1, 7 + p score of p, 21 + c / 28 + c high / low
//           byte of the 13-bit mask of the cards of colour c still in the deck (card 0 = bit 12)
//   block 1 arrival zone (3 cards), 2 / 3 registers of P0 / P1 (5), 4 middle row (5), 5 freed cards (P0: 0..1, P1: 2..3)
//   from byte 210: six [7][7][7] arrays in 350-byte slabs: machine P0 / P1, optim_neighbors P0 / P1 ([0] = cell is a candidate,
//           [3..6] = has a neighbour N/E/S/W), optim_needpipes P0 / P1 ([3..6] = that neighbour has a pipe towards me)
// 428 actions (:60-88): arrival card -> own register slot 0..14, -> middle row slot 15..29, swap mecabot with middle slot 30..34,
// freed card k on cell yx with orientation o = 35 + 4 (49 k + yx) + o, throw the freed cards away 427.
//
// STOCHASTIC = true: make_move ignores random_seed, every card drawn from the deck (_draw_cards :414-438) takes one uniform through
// my_random_choice (:112-115), inside MCTS simulations too.  BUILT_SYM: the 11-14 symmetric forms (:286-409) are built by lane 0
// (mirror of either machine, swap of the freed cards, permutations of the arrival zone / the register columns, colour rolls).
// The quirks of the source are kept: _compute_open_pipes scans 5 x 5 cells (:675-676), _swap_freed strides by 25 (:331-332), the
// connected-component scoring follows the directed label equivalences exactly (:715-787).
//
// make_move / init run on lane 0 over the LDS state (the scoring walk is an explicit-stack depth-first search), the valid-move mask
// is one action per lane in 7 ballots, swap_players is a byte map applied by all lanes.
#pragma once
#include "azg_common.hip.h"

namespace azg {

struct BotanikDev {
    static constexpr int P = 2;
    static constexpr int ROWS = 330, COLS = 7;
    static constexpr int S = 2310;
    static constexpr int SP = RoundUp16<S>::value;
    static constexpr int A = 428;
    static constexpr int AW = (A + 63) / 64;
    static constexpr bool STOCHASTIC = true;
    static constexpr bool RANDOM_SYM = true;      // symmetric forms are built by lane 0 (k_env_symmetries_built); no draw is consumed
    static constexpr bool SYM_DEDUP = false;
    enum { EMPTY = 0, SOURCE = 1 };
    enum { PIPE2_ANGLE = 0, PIPE2_STRAIGHT = 1, PIPE3 = 2, PIPE4 = 3, PLANT = 4, VEGET = 5, MECABOT = 6 };
    enum { TO_REGISTER = 0, OTHERP_EXPAND = 1, OTHERP_SWAP = 2, MAINPL_EXPAND = 3, MAINPL_SWAP = 4 };
    enum { NORTH = 3, EAST = 4, SOUTH = 5, WEST = 6 };
    enum { MS = 7, MM = 49 };
    enum { B_ARRIVAL = 35, B_REG = 70, B_MIDDLE = 140, B_FREED = 175, B_MACH = 210, B_NEIGH = 910, B_NEEDP = 1610 };

    template <class T> __device__ static __forceinline__ T* arrival(T* st, int i) { return st + B_ARRIVAL + 7 * i; }
    template <class T> __device__ static __forceinline__ T* reg(T* st, int p, int i) { return st + B_REG + 35 * p + 7 * i; }
    template <class T> __device__ static __forceinline__ T* middle(T* st, int i) { return st + B_MIDDLE + 7 * i; }
    template <class T> __device__ static __forceinline__ T* freed(T* st, int i) { return st + B_FREED + 7 * i; }
    template <class T> __device__ static __forceinline__ T* cell(T* base, int y, int x) { return base + (y * MS + x) * 7; }
    __device__ static __forceinline__ bool is_empty(const int8_t* c) { return c[0] == EMPTY; }
    __device__ static __forceinline__ bool is_mecabot(const int8_t* c) { return c[2] == MECABOT; }
    __device__ static __forceinline__ void copy7(int8_t* d, const int8_t* s) {
#pragma unroll
        for (int i = 0; i < 7; i++) d[i] = s[i];
    }
    __device__ static __forceinline__ void zero7(int8_t* d) {
#pragma unroll
        for (int i = 0; i < 7; i++) d[i] = 0;
    }
    // np_all_cards[colour][k][1..6] (BotanikConstants.py:57-80), packed: flowers 2 bits, type 3 bits, N E S W 1 bit each
    __device__ static __forceinline__ uint32_t generic(int k) {
        // k = 0..12: flowers {0,0,1,0,0,1,0,0,1,0,3,3,0}, type {0,0,0,1,1,1,2,2,2,3,4,5,6}, pipes NESW {6,6,6,A,A,A,7,7,7,F,2,2,0}
        const uint32_t fl = (0xF10410u >> (2 * k)) & 3u;               // flowers of card k
        const uint32_t ty = (uint32_t)((0x6543222111000ull >> (4 * k)) & 15ull);
        const uint32_t pipes = (uint32_t)((0x022F777AAA666ull >> (4 * k)) & 15ull);   // N = bit 3 ... W = bit 0
        return fl | (ty << 2) | (pipes << 5);
    }
    __device__ static void card_of(int colour, int k, int8_t* out) {
        const uint32_t g = generic(k);
        out[0] = (int8_t)(colour + 2);
        out[1] = (int8_t)(g & 3u);
        out[2] = (int8_t)((g >> 2) & 7u);
        out[3] = (int8_t)((g >> 8) & 1u); out[4] = (int8_t)((g >> 7) & 1u); out[5] = (int8_t)((g >> 6) & 1u); out[6] = (int8_t)((g >> 5) & 1u);
    }

    __device__ static __forceinline__ int get_score(const int8_t* st, int p) { return st[7 + p]; }
    __device__ static __forceinline__ int get_round(const int8_t* st) { return st[0]; }
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return (int)(uint8_t)st[0]; }     // the round only grows
    __device__ static __forceinline__ bool move_uses_seed(int) { return true; }   // (unused: STOCHASTIC edges are never memoised)

    // _draw_cards(1) :414-438; false when the deck is empty (no uniform consumed)
    __device__ static bool draw_card(int8_t* st, Rng& rng, int8_t* out) {
        uint32_t m[5];
        int k = 0;
        for (int c = 0; c < 5; c++) {
            m[c] = (((uint32_t)(uint8_t)st[21 + c]) << 8) | (uint8_t)st[28 + c];
            k += __popc(m[c] & 0x1FFFu);
        }
        if (k == 0) return false;
        // my_random_choice(:112-115): the first index whose running sum of p = mask / k exceeds u.  A card that is gone adds 0.0 / k = +0.0,
        // which leaves the sum as it is, so only the k cards still in the deck take a step; 1.0 / k is the same quotient every time
        const double u = rng.u01();
        const double inv = 1.0 / (double)k;
        double acc = 0.0;
        int pick = 64;
        for (int c = 0; c < 5 && pick == 64; c++) {
            uint32_t left = m[c] & 0x1FFFu;
            while (left) {
                const int bit = 31 - __clz((int)left);                          // card j = 12 - bit: ascending j
                left &= ~(1u << bit);
                acc += inv;
                if (acc > u) { pick = 13 * c + (12 - bit); break; }
            }
        }
        const int c = pick / 13, j = pick - 13 * c;
        const uint32_t nm = m[c] & ~(1u << (12 - j));
        st[21 + c] = (int8_t)(uint8_t)(nm >> 8);
        st[28 + c] = (int8_t)(uint8_t)(nm & 255u);
        card_of(c, j, out);
        return true;
    }
    __device__ static void draw_arrival(int8_t* st, Rng& rng) {                // :440-443
        int8_t cards[3][7];
        for (int i = 0; i < 3; i++)
            if (!draw_card(st, rng, cards[i])) return;
        for (int i = 0; i < 3; i++) copy7(arrival(st, i), cards[i]);
    }

    __device__ static void update_optims(int8_t* st, int p, int y, int x) {    // :615-627
        int8_t *mach = st + B_MACH + 350 * p, *nei = st + B_NEIGH + 350 * p, *need = st + B_NEEDP + 350 * p;
        for (int o = 0; o < 4; o++) {
            const int ny = y + (o == 0 ? -1 : (o == 2 ? 1 : 0)), nx = x + (o == 1 ? 1 : (o == 3 ? -1 : 0));
            if (ny < 0 || ny >= MS || nx < 0 || nx >= MS) continue;
            const int opp = (o + 2) % 4 + 3;
            cell(nei, ny, nx)[0] = (int8_t)(is_empty(cell(mach, ny, nx)) ? 1 : 0);
            cell(nei, ny, nx)[opp] = 1;
            cell(need, ny, nx)[opp] = (int8_t)(cell(mach, y, x)[3 + o] > 0 ? 1 : 0);
        }
        zero7(cell(nei, y, x));
        zero7(cell(need, y, x));
    }

    __device__ static int open_pipes(const int8_t* mach) {                     // _compute_open_pipes :672-686 (5 x 5 scan, as written)
        int n = 0;
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) {
                const int8_t* c = cell(mach, y, x);
                if (is_empty(c)) continue;
                if (y > 0 && is_empty(cell(mach, y - 1, x)) && c[NORTH] > 0) n++;
                if (is_empty(cell(mach, y, x + 1)) && c[EAST] > 0) n++;          // x < 6 and y < 6 always hold here
                if (is_empty(cell(mach, y + 1, x)) && c[SOUTH] > 0) n++;
                if (x > 0 && is_empty(cell(mach, y, x - 1)) && c[WEST] > 0) n++;
            }
        return n;
    }
    // the same count with one of the 5 x 5 cells per lane (wave-uniform result)
    __device__ static int wave_open_pipes(const int8_t* mach) {
        const int l = lane_id(), y = l / 5, x = l - 5 * y;
        int n = 0;
        if (l < 25) {
            const int8_t* c = cell(mach, y, x);
            if (!is_empty(c)) {
                n += (y > 0 && is_empty(cell(mach, y - 1, x)) && c[NORTH] > 0) ? 1 : 0;
                n += (is_empty(cell(mach, y, x + 1)) && c[EAST] > 0) ? 1 : 0;
                n += (is_empty(cell(mach, y + 1, x)) && c[SOUTH] > 0) ? 1 : 0;
                n += (x > 0 && is_empty(cell(mach, y, x - 1)) && c[WEST] > 0) ? 1 : 0;
            }
        }
        return __popcll(__ballot(n & 1)) + 2 * __popcll(__ballot(n & 2)) + 4 * __popcll(__ballot(n & 4));
    }
    // _check_card_on_machine :688-713 for one orientation
    __device__ static bool check_card(const int8_t* card, int y, int x, const int8_t* need, const int8_t* nei, int initial_open, int orient) {
        if (card[2] == PIPE2_STRAIGHT && orient >= 2) return false;
        if (card[2] == PIPE4 && orient >= 1) return false;
        int card_pipes = 0, closed = 0;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int oc = card[3 + ((i - orient + 4) & 3)];                    // np.roll(card[NORTH:], orient)
            const int pwn = oc * nei[3 + i];
            ok = ok && pwn == need[3 + i];
            const bool inb = i == 0 ? y > 0 : (i == 1 ? x < MS - 1 : (i == 2 ? y < MS - 1 : x > 0));
            card_pipes += inb ? oc : 0;
            closed += pwn;
        }
        return ok && initial_open - closed + (card_pipes - closed) > 0;
    }

    // ---- _compute_score :715-787: the recursive walk as an explicit stack (frame = cell + index of its next neighbour) ----
    __device__ static int neighbours(const int8_t* mach, int cur, int* nbc) {   // cells my pipes point at, in N E S W order
        const int y = cur / MS, x = cur - y * MS;
        const int8_t* c = mach + cur * 7;
        int nn = 0;
        if (y > 0 && c[NORTH] > 0) nbc[nn++] = cur - MS;
        if (x < MS - 1 && c[EAST] > 0) nbc[nn++] = cur + 1;
        if (y < MS - 1 && c[SOUTH] > 0) nbc[nn++] = cur + MS;
        if (x > 0 && c[WEST] > 0) nbc[nn++] = cur - 1;
        return nn;
    }
    // The walk's tables live in LDS behind the state (MOVE_SCRATCH): as private arrays with run-time indices they were scratch MEMORY, a
    // global-memory round trip per label / frame access of the one lane that walks.
    static constexpr int MOVE_SCRATCH = 768;
    __device__ static int compute_score(const int8_t* mach, int8_t* scratch) {
        uint64_t visited = 0;
        uint64_t* equiv = (uint64_t*)scratch;                   // [MM + 1]
        short* ncards = (short*)(scratch + 400);                // [MM + 1]
        short* nflow = (short*)(scratch + 500);                 // [MM + 1]
        int8_t* labels = scratch + 600;                         // [MM]
        uint8_t* fr_cell = (uint8_t*)scratch + 656;             // [MM]
        uint8_t* fr_next = (uint8_t*)scratch + 712;             // [MM]
        int n = 0, sp = 0, nbc[4];
        for (int i = 0; i < MM; i++) labels[i] = 99;
        int cur = (MS / 3) * MS + MS / 2;
        bool enter = true;
        for (;;) {
            if (enter) {                                                        // first pass of _dfs on `cur` (:739-766)
                const int8_t* c = mach + cur * 7;
                visited |= 1ull << cur;
                const int nn = neighbours(mach, cur, nbc);
                int best = 99;
                for (int i = 0; i < nn; i++)
                    if (mach[nbc[i] * 7] == c[0]) best = labels[nbc[i]] < best ? labels[nbc[i]] : best;
                if (best == 99) {
                    best = n;
                    equiv[n] = 1ull << n; ncards[n] = 1; nflow[n] = c[1];
                    n++;
                } else {
                    for (int i = 0; i < nn; i++)
                        if (mach[nbc[i] * 7] == c[0] && labels[nbc[i]] != 99) equiv[labels[nbc[i]]] |= 1ull << best;
                    ncards[best] = (short)(ncards[best] + 1); nflow[best] = (short)(nflow[best] + c[1]);
                }
                labels[cur] = (int8_t)best;
                fr_cell[sp] = (uint8_t)cur; fr_next[sp] = 0;
                sp++;
                enter = false;
            }
            if (sp == 0) break;
            // second pass (:768-771): resume the top frame at its next neighbour that holds a card and was not visited yet
            const int f = sp - 1, fc = fr_cell[f];
            const int nn = neighbours(mach, fc, nbc);
            bool descended = false;
            for (int i = fr_next[f]; i < nn; i++) {
                const int t = nbc[i];
                if (mach[t * 7] != EMPTY && !((visited >> t) & 1ull)) {
                    fr_next[f] = (uint8_t)(i + 1);
                    cur = t;
                    enter = descended = true;
                    break;
                }
            }
            if (!descended) sp--;
        }
        // _score_sum :776-787: areas reachable through the directed equivalences, each label counted once
        uint64_t seen = 0;
        int total = 0;
        for (int a = 1; a < n; a++) {
            int cards = 0, flowers = 0;
            uint64_t todo = 1ull << a;
            while (todo) {
                const int i = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1;
                if ((seen >> i) & 1ull) continue;
                seen |= 1ull << i;
                cards += ncards[i]; flowers += nflow[i];
                todo |= equiv[i] & ~seen;
            }
            total += cards >= 3 ? cards + flowers : flowers;
        }
        return total;
    }

    __device__ static void free_card_if_needed(int8_t* st, int slot) {         // :505-547
        const int mc = middle(st, slot)[0], mt = middle(st, slot)[2];
        for (int p = 0; p < 2; p++) {
            int8_t* r = reg(st, p, slot);
            if (is_empty(r) || r[0] == mc || r[2] == mt) continue;
            const int ns = is_empty(freed(st, 2 * p)) ? 0 : 1;                 // both busy: slot 1 is overwritten, as written
            copy7(freed(st, 2 * p + ns), r);
            zero7(r);
            const bool is_main = p == st[2];
            int status;
            if (is_mecabot(freed(st, 2 * p + ns))) {
                status = is_main ? MAINPL_SWAP : OTHERP_SWAP;
                if (ns != 0) {                                                 // the mecabot goes first
                    int8_t t[7];
                    copy7(t, freed(st, 2 * p + 1));
                    copy7(freed(st, 2 * p + 1), freed(st, 2 * p));
                    copy7(freed(st, 2 * p), t);
                }
            } else {
                status = is_main ? MAINPL_EXPAND : OTHERP_EXPAND;
            }
            if (status > st[1]) st[1] = (int8_t)status;
        }
    }
    __device__ static void next_status(int8_t* st) {                           // :592-604, 632-644
        const int mainpl = st[2], otherp = 1 - mainpl;
        if (is_mecabot(freed(st, 2 * mainpl))) st[1] = MAINPL_SWAP;
        else if (!is_empty(freed(st, 2 * mainpl))) st[1] = MAINPL_EXPAND;
        else if (is_mecabot(freed(st, 2 * otherp))) st[1] = OTHERP_SWAP;
        else if (!is_empty(freed(st, 2 * otherp))) st[1] = OTHERP_EXPAND;
        else st[1] = TO_REGISTER;
    }

    // Board.valid_moves restricted to one action (:191-201, 445-486); action 427 is decided by valid_mask
    __device__ static bool valid_action(const int8_t* st, int a, int player, int nb_open) {
        const int status = st[1];
        if (a < 30) {
            if (status != TO_REGISTER) return false;
            const int i = (a % 15) / 5, k = a % 5;
            const int8_t* card = arrival(st, i);
            if (is_empty(card)) return false;
            if (a >= 15) return true;
            return is_empty(reg(st, player, k)) && (middle(st, k)[0] == card[0] || middle(st, k)[2] == card[2]);
        }
        if (a < 35) return (status == MAINPL_SWAP || status == OTHERP_SWAP) && middle(st, a - 30)[2] != MECABOT;
        if (a >= A - 1 || !(status == MAINPL_EXPAND || status == OTHERP_EXPAND)) return false;
        const int k = (a - 35) / (4 * MM), rem = (a - 35) - k * 4 * MM, yx = rem >> 2, o = rem & 3;
        const int8_t* card = freed(st, 2 * player + k);
        if (is_empty(card)) return false;
        const int8_t* nei = st + B_NEIGH + 350 * player + yx * 7;
        if (!nei[0]) return false;
        return check_card(card, yx / MS, yx % MS, st + B_NEEDP + 350 * player + yx * 7, nei, nb_open, o);
    }
    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
        const int status = st[1];
        const bool expand = status == MAINPL_EXPAND || status == OTHERP_EXPAND;
        const int nb_open = expand ? wave_open_pipes(st + B_MACH + 350 * player) : 0;
        uint64_t any = 0;
#pragma unroll 1
        for (int k = 0; k < AW; k++) {
            const int a = k * 64 + l;
            uint64_t m = __ballot(a < A && valid_action(st, a < A ? a : 0, player, nb_open));
            any |= m;
            if (k == AW - 1 && expand && any == 0) m |= 1ull << ((A - 1) & 63);     // nothing fits: throw the freed cards away
            if (l == 0) mask_lds[k] = m;
        }
    }

    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        return lane0_make_move<BotanikDev>(st, move, player, seed, rng);
    }
    // Board.make_move :203-230 -- lane 0 only
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        (void)seed;
        if (move < 15) {                                                        // _move_to_register :488-495
            copy7(reg(st, player, move % 5), arrival(st, move / 5));
            zero7(arrival(st, move / 5));
        } else if (move < 30) {                                                 // _move_to_middle_row_and_unlink :497-503
            const int ci = (move - 15) / 5, slot = (move - 15) % 5;
            copy7(middle(st, slot), arrival(st, ci));
            zero7(arrival(st, ci));
            free_card_if_needed(st, slot);
        } else if (move < 35) {                                                 // _swap_mecabot :549-567
            const int slot = move - 30;
            int8_t t[7];
            copy7(t, freed(st, 2 * player));
            copy7(freed(st, 2 * player), middle(st, slot));
            copy7(middle(st, slot), t);
            if (st[1] == MAINPL_SWAP) st[1] = MAINPL_EXPAND;
            else if (st[1] == OTHERP_SWAP) st[1] = OTHERP_EXPAND;
            free_card_if_needed(st, slot);
        } else if (move < A - 1) {                                              // _expand_machine :569-604
            const int ci = (move - 35) / (4 * MM), rem = (move - 35) - ci * 4 * MM, slot = rem >> 2, orient = rem & 3;
            const int y = slot / MS, x = slot - y * MS;
            int8_t* dst = st + B_MACH + 350 * player + slot * 7;
            const int8_t* card = freed(st, 2 * player + ci);
            dst[0] = card[0]; dst[1] = card[1]; dst[2] = card[2];
            for (int i = 0; i < 4; i++) dst[3 + i] = card[3 + ((i - orient + 4) & 3)];
            zero7(freed(st, 2 * player + ci));
            update_optims(st, player, y, x);
            if (ci == 0 && !is_empty(freed(st, 2 * player + 1))) {
                copy7(freed(st, 2 * player), freed(st, 2 * player + 1));
                zero7(freed(st, 2 * player + 1));
            }
            st[7 + player] = (int8_t)compute_score(st + B_MACH + 350 * player, st + SP);
            next_status(st);
        } else {                                                                // _throw_cards_away :629-644
            zero7(freed(st, 2 * player)); zero7(freed(st, 2 * player + 1));
            next_status(st);
        }
        const int status = st[1];
        int mainpl = st[2];
        if (status == TO_REGISTER) {
            if (is_empty(arrival(st, 0)) && is_empty(arrival(st, 1)) && is_empty(arrival(st, 2))) draw_arrival(st, rng);
            st[0] = (int8_t)(st[0] + 1);
            mainpl = 1 - mainpl;
            st[2] = (int8_t)mainpl;
            return mainpl;
        }
        return (status == MAINPL_EXPAND || status == MAINPL_SWAP) ? mainpl : 1 - mainpl;
    }

    // Board.check_end_game :235-252 (uniform)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        bool live = false;
        for (int z = 21; z < 35; z++) live = live || st[z] != 0;
        for (int i = 0; i < 3; i++) live = live || !is_empty(arrival(st, i));
        for (int i = 0; i < 4; i++) live = live || !is_empty(freed(st, i));
        if (live) { out[0] = 0.f; out[1] = 0.f; return false; }
        int a = st[7], b = st[8];
        if (a == b) {
            a = b = 0;
            for (int i = 0; i < MM; i++) { a += st[B_MACH + 7 * i] != 0; b += st[B_MACH + 350 + 7 * i] != 0; }
        }
        out[0] = a > b ? 1.f : (a < b ? -1.f : 0.01f);
        out[1] = a > b ? -1.f : (a < b ? 1.f : 0.01f);
        return true;
    }

    // Board.swap_players :254-284 (k == 1): registers, freed cards, scores and the three machine arrays trade places (343 of the
    // 350 bytes of every slab), the status maps 1 <-> 3, 2 <-> 4 and the main player flips
    // In place, one exchange per lane and step: the two players' bytes are disjoint pairs -- registers 35, freed cards 14, three pairs of
    // slabs (both start on even offsets: 171 two-byte words + 1 byte each), the scores -- 566 exchanges = 9 steps of the wave, no copy.
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        (void)tmp;
        if (k != 1) return;
        constexpr int NPAIR = 35 + 14 + 1 + 3 * 172;
        for (int i = lane_id(); i < NPAIR; i += 64) {
            int a, b;
            bool wide = false;
            if (i < 35) { a = B_REG + i; b = a + 35; }
            else if (i < 49) { a = B_FREED + (i - 35); b = a + 14; }
            else if (i == 49) { a = 7; b = 8; }
            else {
                const int j = i - 50, pair = j / 172, w = j - pair * 172;
                a = B_MACH + 700 * pair + 2 * w; b = a + 350;
                wide = w < 171;
            }
            if (wide) {
                const uint16_t x = *(const uint16_t*)(st + a), y = *(const uint16_t*)(st + b);
                *(uint16_t*)(st + a) = y; *(uint16_t*)(st + b) = x;
            } else {
                const int8_t x = st[a], y = st[b];
                st[a] = y; st[b] = x;
            }
        }
        if (lane_id() == 0) {
            if (st[1] > TO_REGISTER) st[1] = (int8_t)((st[1] + 1) % 4 + 1);
            st[2] = (int8_t)(1 - st[2]);
        }
        wave_sync();
    }

    // init_game :152-163, 606-613 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        for (int c = 0; c < 5; c++) { st[21 + c] = 0x1F; st[28 + c] = (int8_t)0xFF; }
        for (int i = 0; i < 5; i++) draw_card(st, rng, middle(st, i));
        draw_arrival(st, rng);
        for (int p = 0; p < 2; p++) {
            int8_t* src = st + B_MACH + 350 * p + ((MS / 3) * MS + MS / 2) * 7;
            src[0] = SOURCE; src[5] = 1;
        }
        st[3] = 1; st[4] = 1;
        for (int p = 0; p < 2; p++) update_optims(st, p, MS / 3, MS / 2);
    }

    // ---- get_symmetries :286-409: lane 0 turns `cand` (a copy of the input state) into form c and fills act_src (out[a] = in[act_src[a]])
    static constexpr int NSYM_CAND = 14;
    __device__ static void mirror_machine(int8_t* mach) {                      // :293-305
        for (int y = 0; y < MS; y++)
            for (int x = 0; x < (MS + 1) / 2; x++) {
                int8_t *a = cell(mach, y, x), *b = cell(mach, y, MS - 1 - x);
                if (MS - 1 - x != x) {
                    for (int i = 0; i < 7; i++) { const int8_t t = a[i]; a[i] = b[i]; b[i] = t; }
                    const int8_t w = b[EAST]; b[EAST] = b[WEST]; b[WEST] = w;
                }
                const int8_t w = a[EAST]; a[EAST] = a[WEST]; a[WEST] = w;
            }
    }
    __device__ static void roll_colors(int8_t* cards, int n, int nroll) {      // :356-367
        for (int i = 0; i < n; i++) {
            const int c = cards[7 * i];
            if (c != EMPTY && c != SOURCE) cards[7 * i] = (int8_t)((c - 2 + nroll) % 5 + 2);
        }
    }
    __device__ static bool sym_build(const int8_t* st, int c, int8_t* cand, int16_t* act_src, Rng& rng, const uint8_t* valids) {
        (void)rng; (void)valids;
        for (int a = 0; a < A; a++) act_src[a] = (int16_t)a;
        if (c == 0) return true;
        if (c == 1) {                                                           // mirror of machine 0 + policy / valids (:307-324)
            mirror_machine(cand + B_MACH);
            for (int y = 0; y < MS; y++)
                for (int x = 0; x < (MS + 1) / 2; x++)
                    for (int ci = 0; ci < 2; ci++) {
                        const bool angle = freed(st, ci)[2] == PIPE2_ANGLE;
                        for (int o = 0; o < 4; o++) {
                            const int no = angle ? (o ^ 1) : ((4 - o) & 3);      // [1,0,3,2] / [0,3,2,1]
                            const int a1 = 35 + 4 * (MM * ci + MS * y + x), a2 = 35 + 4 * (MM * ci + MS * y + MS - 1 - x);
                            act_src[a2 + no] = (int16_t)(a1 + o);
                            act_src[a1 + no] = (int16_t)(a2 + o);
                        }
                    }
            return true;
        }
        if (c == 2) { mirror_machine(cand + B_MACH + 350); return true; }
        if (c == 3) {                                                           // _swap_freed :326-337 (stride 25, as written)
            if (is_empty(freed(st, 0)) || is_empty(freed(st, 1))) return false;
            for (int yx = 0; yx < MM; yx++)
                for (int o = 0; o < 4; o++) {
                    const int a1 = 35 + 4 * yx + o, a2 = 35 + 4 * (25 + yx) + o;
                    act_src[a2] = (int16_t)a1;
                    act_src[a1] = (int16_t)a2;
                }
            for (int i = 0; i < 7; i++) { const int8_t t = freed(cand, 0)[i]; freed(cand, 0)[i] = freed(cand, 1)[i]; freed(cand, 1)[i] = t; }
            return true;
        }
        if (c < 7) {                                                            // _permute_arrival :339-345: {0,2,1} {1,0,2} {2,1,0}
            const int q = c - 4;
            for (int i = 0; i < 3; i++) {
                const int ni = q == 0 ? (i == 0 ? 0 : 3 - i) : (q == 1 ? (i == 2 ? 2 : 1 - i) : 2 - i);
                copy7(arrival(cand, ni), arrival(st, i));
                for (int j = 0; j < 5; j++) {
                    act_src[5 * ni + j] = (int16_t)(5 * i + j);
                    act_src[5 * ni + 15 + j] = (int16_t)(5 * i + 15 + j);
                }
            }
            return true;
        }
        if (c < 12) {                                                           // _permute_registers :347-354
            // permutations_registers (BotanikConstants.py:47-53), 3 bits per entry: {0,3,2,4,1} {1,0,3,2,4} {2,4,1,0,3} {3,2,4,1,0} {4,1,0,3,2}
            const uint32_t perm = c == 7 ? 0x1898u : (c == 8 ? 0x44C1u : (c == 9 ? 0x3062u : (c == 10 ? 0x0313u : 0x260Cu)));
            for (int i = 0; i < 5; i++) {
                const int ni = (int)((perm >> (3 * i)) & 7u);
                copy7(reg(cand, 0, ni), reg(st, 0, i));
                copy7(reg(cand, 1, ni), reg(st, 1, i));
                copy7(middle(cand, ni), middle(st, i));
                for (int z = 0; z < 7; z++) act_src[z * 5 + ni] = (int16_t)(z * 5 + i);
            }
            return true;
        }
        const int nroll = c == 12 ? 2 : 4;                                      // colour rolls :393-403
        roll_colors(cand + B_ARRIVAL, 5, nroll); roll_colors(cand + B_REG, 5, nroll); roll_colors(cand + B_REG + 35, 5, nroll);
        roll_colors(cand + B_MIDDLE, 5, nroll); roll_colors(cand + B_FREED, 5, nroll);
        roll_colors(cand + B_MACH, MM, nroll); roll_colors(cand + B_MACH + 350, MM, nroll);
        return true;
    }
    // (the deterministic per-byte interface is unused: k_env_symmetries takes the built path)
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int c) { return c == 0; }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int, int i) { return st[i]; }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int, int a) { return a; }
};

}  // namespace azg
