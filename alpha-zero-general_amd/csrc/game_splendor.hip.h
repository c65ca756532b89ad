// game_splendor.hip.h -- Splendor (2-4 players) env step for one wavefront, state staged in LDS.
//
// Semantics follow splendor/SplendorLogicNumba.py `Board` (line numbers cited); the byte layout of the state is the
// reference's int8[(32+10n+n*n)][7] (copy_state :207-219).  valid_mask() is lane-parallel and branch-free (one action per lane, two
// ballots for the 81 actions); wave_make_move() keeps the branchy rule arithmetic on the scalar unit (see below).
#pragma once
#include "azg_common.hip.h"
#include "splendor_tables.h"

namespace azg {

template <int NP>
struct SplendorDev {
    static constexpr bool STOCHASTIC = false;   // the env step is a function of (state, action, random_seed): edges are memoised
    static constexpr bool RANDOM_SYM = false;   // get_symmetries draws no randomness
    static constexpr int P = NP;
    static constexpr int NN = NP + 1;                       // nobles in play  :145
    static constexpr int ROWS = 32 + 10 * NP + NP * NP;     // observation_size :90-92
    static constexpr int COLS = 7;
    static constexpr int S = ROWS * COLS;
    static constexpr int SP = RoundUp16<S>::value;          // padded stride in HBM / LDS (zero tail)
    static constexpr int A = 81;                            // action_size :94-96
    static constexpr int AW = 2;                            // 64-bit words of the valid mask
    static constexpr int R_NOBLES = 31, R_GEMS = 32 + NP, R_PNOB = 32 + 2 * NP, R_PCARDS = 32 + 3 * NP + NP * NP,
                         R_RES = 32 + 4 * NP + NP * NP;
    static constexpr int GOLD = 5, PTS = 6;
    static constexpr int MAX_MOVES = 62 * NP;               // :146

    __device__ static __forceinline__ const int8_t* row(const int8_t* st, int r) { return st + r * COLS; }
    __device__ static __forceinline__ int8_t* row(int8_t* st, int r) { return st + r * COLS; }
    __device__ static __forceinline__ int sum5(const int8_t* r) { return r[0] + r[1] + r[2] + r[3] + r[4]; }
    __device__ static __forceinline__ int sum7(const int8_t* r) { return sum5(r) + r[5] + r[6]; }

    // sum(max(cost - gems - cards, 0)) with int8 wrap (:350, :363)
    __device__ static __forceinline__ int missing(const int8_t* cost, const int8_t* gems, const int8_t* cards) {
        int s = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            int8_t d = (int8_t)((int8_t)(cost[c] - gems[c]) - cards[c]);
            s += d > 0 ? d : 0;
        }
        return s;
    }

    // Board.valid_moves :180-188, wave-cooperative and branch-free: lane l answers action l (first ballot) and action
    // 64+l (second ballot).  The player's rows are wave-uniform bytes; only the 30 card actions read a per-lane row.
    // Lane 0 writes the AW mask words (caller syncs).
    __device__ static __forceinline__ void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
        const int8_t* bankp = row(st, 0);
        const int8_t* gemsp = row(st, R_GEMS + player);
        const int8_t* cardsp = row(st, R_PCARDS + player);
        const int8_t* lastres = row(st, R_RES + 6 * player + 5);
        int bank[5], gems[7], cards[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { bank[c] = bankp[c]; cards[c] = cardsp[c]; }
#pragma unroll
        for (int c = 0; c < 7; c++) gems[c] = gemsp[c];
        const bool empty_slot = sum5(lastres) == 0;                                   // :376
        const int r = l < 12 ? 1 + 2 * l : l < 24 ? 1 + 2 * (l - 12) : l < 27 ? 25 + 2 * (l - 24)
                    : l < 30 ? R_RES + 6 * player + 2 * (l - 27) : 0;
        const int8_t* cr = row(st, r);
        int cost[5];
#pragma unroll
        for (int c = 0; c < 5; c++) cost[c] = cr[c];
        int gsum = 0;
#pragma unroll
        for (int c = 0; c < 7; c++) gsum += gems[c];
        int miss = 0, csum = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const int8_t d = (int8_t)((int8_t)(cost[c] - gems[c]) - cards[c]);        // int8 wrap :363
            miss += d > 0 ? d : 0;
            csum += cost[c];
        }
        const bool nz = csum != 0;
        const bool ok_buy = miss <= gems[GOLD] && nz;                                 // :359-368, :402-412
        const bool ok_res = empty_slot && nz;                                         // :375-380
        constexpr uint32_t g3[5] = SPL_GEMS3_COL_INIT, g2[5] = SPL_GEMS2_COL_INIT;   // bit i of gN[c] = table[i][c]
        const int i3 = (l - 30) & 31;                                                 // lanes 30..54
        const int ig = (l >= 60 ? l - 60 : l + 4) & 15;                               // lanes 60..63 (a=l) and 0..10 (a=64+l)
        bool ok3 = true, okg = true;
        int k3 = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const int t3 = (int)((g3[c] >> i3) & 1u), t2 = (int)((g2[c] >> ig) & 1u);
            ok3 = ok3 && ((int8_t)(bank[c] - t3) >= 0);                               // :422-427
            okg = okg && ((int8_t)(gems[c] - t2) >= 0);                               // :446-449
            k3 += t3;
        }
        ok3 = ok3 && (gsum + k3 <= 10);
        const int cs = l - 55, cg = l - 11;                                           // identical-gem colours
        int bank_cs = bank[0], gems_cg = gems[0];
#pragma unroll
        for (int c = 1; c < 5; c++) { bank_cs = cs == c ? bank[c] : bank_cs; gems_cg = cg == c ? gems[c] : gems_cg; }
        const bool ok_same = bank_cs >= 4 && gsum + 2 <= 10;                          // :429-434
        const bool v0 = l < 12 ? ok_buy : l < 27 ? ok_res : l < 30 ? ok_buy : l < 55 ? ok3 : l < 60 ? ok_same : okg;
        const bool v1 = l < 11 ? okg : l < 16 ? gems_cg >= 2 : l == 16;               // :451-453, pass :187
        const uint64_t m0 = __ballot(v0);
        const uint64_t m1 = __ballot(v1);
        if (l == 0) { mask_lds[0] = m0; mask_lds[1] = m1; }
    }

    // _get_deck_card :306-336 (lane 0).  Writes 14 card bytes; returns false when the deck is empty.
    __device__ static bool get_deck_card(int8_t* st, int tier, long long seed, Rng& rng, int8_t* card14) {
        int8_t* cnt = row(st, 25 + 2 * tier);
        int8_t* bits = row(st, 26 + 2 * tier);
        int total = sum5(cnt);
        if (total == 0) return false;
        int color = 0, card_index = 0;
        if (seed == 0) {                                                   // true random :311-315
            double u = rng.u01(), acc = 0.0;
            int k = 0;
            for (; k < 5; k++) { acc += (double)cnt[k] / (double)total; if (acc > u) break; }
            if (k >= 5) { for (k = 4; k > 0 && cnt[k] == 0; k--) {} }
            color = k;
            uint32_t b = (uint8_t)bits[color];
            int nb = __popc(b);
            double u2 = rng.u01();
            acc = 0.0;
            int idx = -1, last = 0;
            for (int i = 0; i < 8; i++) {
                int set = (b >> (7 - i)) & 1;
                if (set) last = i;
                acc += (double)set / (double)nb;
                if (acc > u2) { idx = i; break; }
            }
            card_index = idx < 0 ? last : idx;
        } else {                                                           // seeded universe draw :316-323
            int n = 0;
            long long seedv = 0, pw = 1;
            for (int c = 0; c < 5; c++) {
                uint32_t b = (uint8_t)bits[c];
                n += __popc(b);
                seedv += (long long)b * pw;
                pw *= 32;
            }
            long long x = 4594591LL * (seed + seedv);
            long long r = x % n;
            if (r < 0) r += n;                                             // Python floor-mod
            // r-th candidate in colour-major, MSB-first order
            int rem = (int)r;
            for (int c = 0; c < 5; c++) {
                uint32_t b = (uint8_t)bits[c];
                int pc = __popc(b);
                if (rem < pc) {
                    color = c;
                    for (int i = 0; i < 8; i++)
                        if ((b >> (7 - i)) & 1) { if (rem == 0) { card_index = i; break; } rem--; }
                    break;
                }
                rem -= pc;
            }
        }
        uint32_t b = (uint8_t)bits[color];
        b &= ~(0x80u >> card_index);
        bits[color] = (int8_t)(uint8_t)b;                                  // int8 wrap :327
        cnt[color] -= 1;
#pragma unroll
        for (int i = 0; i < 14; i++) card14[i] = SPL_CARDS[tier][color][card_index][i];
        return true;
    }

    __device__ static void fill_new_card(int8_t* st, int tier, int index, long long seed, Rng& rng) {   // :338-342
        int8_t* dst = row(st, 1 + 8 * tier + 2 * index);
        int8_t card[14];
        bool got = get_deck_card(st, tier, seed, rng, card);
#pragma unroll
        for (int i = 0; i < 14; i++) dst[i] = got ? card[i] : (int8_t)0;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // Board.make_move :190-205 for the whole wave.  The env step is wave-uniform, branchy byte arithmetic: run by one lane
    // it costs a full VALU instruction per byte operation (plus an LDS round trip per access).  Here the state is pulled
    // into registers (lane i holds dwords i, 64+i, ... of the padded state), rows are fetched with v_readlane into SGPR
    // pairs (7 packed bytes) and all the rule arithmetic stays on the scalar unit; modified rows go back with
    // a compare+select.  Only the 14-byte card moves (table -> board slot, board slot -> reserve, reserve shift) are done as
    // lane-parallel byte copies in LDS after the registers have been stored back.
    // ------------------------------------------------------------------------------------------------------------------
    static constexpr int SPW = SP / 4, NW = (SPW + 63) / 64;
    struct SW {
        uint32_t w[NW];
        __device__ __forceinline__ void load(const int8_t* st) {
#pragma unroll
            for (int k = 0; k < NW; k++) { const int i = lane_id() + 64 * k; w[k] = i < SPW ? ((const uint32_t*)st)[i] : 0u; }
        }
        __device__ __forceinline__ void store(int8_t* st) const {
#pragma unroll
            for (int k = 0; k < NW; k++) { const int i = lane_id() + 64 * k; if (i < SPW) ((uint32_t*)st)[i] = w[k]; }
        }
        __device__ __forceinline__ uint32_t rd(int d) const {                      // d wave-uniform
            uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)w[0], d & 63);
#pragma unroll
            for (int k = 1; k < NW; k++) {
                const uint32_t vk = (uint32_t)__builtin_amdgcn_readlane((int)w[k], d & 63);
                v = (d >> 6) == k ? vk : v;
            }
            return v;
        }
        __device__ __forceinline__ void wr(int d, uint32_t v) {
#pragma unroll
            for (int k = 0; k < NW; k++) w[k] = (lane_id() + 64 * k == d) ? v : w[k];   // v_cmp + v_cndmask (no writelane builtin)
        }
        __device__ __forceinline__ uint64_t row(int r) const {                     // 7 bytes of row r, packed little-endian
            const int o = 7 * r, d = o >> 2, sh = (o & 3) * 8;
            uint64_t v = (((uint64_t)rd(d + 1) << 32) | rd(d)) >> sh;
            if (sh > 8) v |= (uint64_t)rd(d + 2) << (64 - sh);
            return v & 0x00FFFFFFFFFFFFFFull;
        }
        __device__ __forceinline__ void set_row(int r, uint64_t v) {
            const int o = 7 * r, d = o >> 2, sh = (o & 3) * 8;
            const uint64_t m = 0x00FFFFFFFFFFFFFFull, M = m << sh, V = (v & m) << sh;
            wr(d, (rd(d) & ~(uint32_t)M) | (uint32_t)V);
            wr(d + 1, (rd(d + 1) & ~(uint32_t)(M >> 32)) | (uint32_t)(V >> 32));
            if (sh > 8) wr(d + 2, (rd(d + 2) & ~(uint32_t)(m >> (64 - sh))) | (uint32_t)((v & m) >> (64 - sh)));
        }
    };
    __device__ static __forceinline__ int B(uint64_t row, int c) { return (int)(int8_t)(uint8_t)(row >> (8 * c)); }
    __device__ static __forceinline__ int UB(uint64_t row, int c) { return (int)(uint8_t)(row >> (8 * c)); }
    __device__ static __forceinline__ void SETB(uint64_t& row, int c, int v) {
        row = (row & ~(0xFFull << (8 * c))) | ((uint64_t)(uint8_t)v << (8 * c));
    }
    __device__ static __forceinline__ int rsum5(uint64_t r) { return B(r, 0) + B(r, 1) + B(r, 2) + B(r, 3) + B(r, 4); }
    __device__ static __forceinline__ int rsum7(uint64_t r) { return rsum5(r) + B(r, 5) + B(r, 6); }

    // v mod n for v < 2^22 (exact in f32, quotient estimate off by at most one)
    __device__ static __forceinline__ uint32_t small_mod(uint32_t v, uint32_t n, float rn) {
        const uint32_t q = (uint32_t)((float)v * rn);
        int r = (int)(v - q * n);
        r = r < 0 ? r + (int)n : r;
        r = r >= (int)n ? r - (int)n : r;
        return (uint32_t)r;
    }
    // (4594591 * b) mod n with Python floor-mod semantics (:316-323), n <= 40: Horner over 16-bit limbs
    __device__ static __forceinline__ int draw_mod(long long b, int n) {
        const float rn = 1.0f / (float)n;
        const bool neg = b < 0;
        const uint64_t ub = neg ? (uint64_t)(-b) : (uint64_t)b;
        uint32_t mb = 0;
#pragma unroll
        for (int k = 3; k >= 0; k--) mb = small_mod(mb * 65536u + (uint32_t)((ub >> (16 * k)) & 0xFFFFu), (uint32_t)n, rn);
        uint32_t ma = small_mod(70u, (uint32_t)n, rn);                              // 4594591 = 70 * 65536 + 7071
        ma = small_mod(ma * 65536u + 7071u, (uint32_t)n, rn);
        uint32_t r = small_mod(ma * mb, (uint32_t)n, rn);
        if (neg && r) r = (uint32_t)n - r;
        return (int)r;
    }

    // _get_deck_card :306-336 on the register view: picks (colour, index), updates the deck rows; false = deck empty
    __device__ static __forceinline__ bool draw_card(SW& s, int tier, long long seed, Rng& rng, int* color_o, int* idx_o) {
        uint64_t cnt = s.row(25 + 2 * tier), bits = s.row(26 + 2 * tier);
        const int total = rsum5(cnt);
        if (total == 0) return false;
        int color = 0, card_index = 0;
        if (seed == 0) {                                                   // true random :311-315
            const double u = rng.u01();
            double acc = 0.0;
            int k = 0;
            for (; k < 5; k++) { acc += (double)B(cnt, k) / (double)total; if (acc > u) break; }
            if (k >= 5) { for (k = 4; k > 0 && B(cnt, k) == 0; k--) {} }
            color = k;
            const uint32_t b = (uint32_t)UB(bits, color);
            const int nb = __popc(b);
            const double u2 = rng.u01();
            acc = 0.0;
            int idx = -1, last = 0;
            for (int i = 0; i < 8; i++) {
                const int set = (b >> (7 - i)) & 1;
                if (set) last = i;
                acc += (double)set / (double)nb;
                if (acc > u2) { idx = i; break; }
            }
            card_index = idx < 0 ? last : idx;
        } else {                                                           // seeded universe draw :316-323
            int n = 0;
            long long seedv = 0, pw = 1;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const uint32_t b = (uint32_t)UB(bits, c);
                n += __popc(b);
                seedv += (long long)b * pw;
                pw *= 32;
            }
            int rem = draw_mod(seed + seedv, n);
            // rem-th candidate in colour-major, MSB-first order
            for (int c = 0; c < 5; c++) {
                const uint32_t b = (uint32_t)UB(bits, c);
                const int pc = __popc(b);
                if (rem < pc) {
                    color = c;
                    for (int i = 0; i < 8; i++)
                        if ((b >> (7 - i)) & 1) { if (rem == 0) { card_index = i; break; } rem--; }
                    break;
                }
                rem -= pc;
            }
        }
        SETB(bits, color, (int)((uint32_t)UB(bits, color) & ~(0x80u >> card_index)));   // int8 wrap :327
        SETB(cnt, color, B(cnt, color) - 1);
        s.set_row(25 + 2 * tier, cnt);
        s.set_row(26 + 2 * tier, bits);
        *color_o = color; *idx_o = card_index;
        return true;
    }

    // _buy_card :344-357 + _give_nobles_if_earned :465-470 on the register view
    __device__ static __forceinline__ void buy_card(SW& s, uint64_t& bank, uint64_t& gems, uint64_t c0, uint64_t c1, int player) {
        uint64_t cards = s.row(R_PCARDS + player);
        int miss = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const int8_t d = (int8_t)((int8_t)(B(c0, c) - B(gems, c)) - B(cards, c));
            miss += d > 0 ? d : 0;
        }
#pragma unroll
        for (int c = 0; c < 5; c++) {
            int8_t need = (int8_t)(B(c0, c) - B(cards, c));
            need = need < 0 ? (int8_t)0 : need;
            const int8_t g = (int8_t)B(gems, c);
            const int8_t paid = need < g ? need : g;
            SETB(gems, c, g - paid);
            SETB(bank, c, B(bank, c) + paid);
        }
        SETB(gems, GOLD, B(gems, GOLD) - miss);
        SETB(bank, GOLD, B(bank, GOLD) + miss);
#pragma unroll
        for (int c = 0; c < COLS; c++) SETB(cards, c, B(cards, c) + B(c1, c));
        s.set_row(R_PCARDS + player, cards);
        for (int i = 0; i < NN; i++) {
            const uint64_t noble = s.row(R_NOBLES + i);
            if (rsum5(noble) <= 0) continue;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 5; c++) ok = ok && B(cards, c) >= B(noble, c);
            if (ok) {
                s.set_row(R_PNOB + NN * player + i, noble);
                s.set_row(R_NOBLES + i, 0ull);
            }
        }
    }

    // monotone "age" of a state for the clean-up (a state older than the root can never be reached again): the move counter
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return get_round(st); }

    // true when the env step of `move` may read random_seed: buying / reserving a visible or deck card draws a replacement
    // (_get_deck_card :306-336); buying a reserved card and the gem moves never do
    __device__ static __forceinline__ bool move_uses_seed(int move) { return move < 27; }
    static constexpr int SEED_ACTIONS = 27;     // == the bound above: only these entries get one child slot per universe (forest.hip.h RecGeom)

    // ALL lanes call with wave-uniform arguments; returns the next player; the LDS state is updated and synchronised.
    __device__ static int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        move = uni_i32(move); player = uni_i32(player);
        seed = (long long)(((uint64_t)uni_u32((uint32_t)((uint64_t)seed >> 32)) << 32) | uni_u32((uint32_t)seed));
        const int l = lane_id();
        SW s;
        s.load(st);
        uint64_t bank = s.row(0), gems = s.row(R_GEMS + player);
        const int res = R_RES + 6 * player;
        int fill_row = -1, fill_tier = 0, fill_color = 0, fill_idx = 0;   // 2-row slot that receives a card (or zeros)
        bool fill_got = false;
        int copy_src = -1, copy_dst = 0;                                    // 14-byte move board slot -> reserve slot
        int shift_i = -1;                                                   // bought reserve slot: close the gap
        if (move < 12) {                                                   // _buy :370-373
            buy_card(s, bank, gems, s.row(1 + 2 * move), s.row(2 + 2 * move), player);
            fill_tier = move >> 2;
            fill_got = draw_card(s, fill_tier, seed, rng, &fill_color, &fill_idx);
            fill_row = 1 + 2 * move;
        } else if (move < 27) {                                            // _reserve :382-400
            const int i = move - 12;
            int slot = 2;
            for (int k = 2; k >= 0; k--) if (rsum5(s.row(res + 2 * k)) == 0) slot = k;
            if (i < 12) {
                copy_src = 1 + 2 * i; copy_dst = res + 2 * slot;
                fill_tier = i >> 2;
                fill_got = draw_card(s, fill_tier, seed, rng, &fill_color, &fill_idx);
                fill_row = 1 + 2 * i;
            } else {
                fill_tier = i - 12;
                fill_got = draw_card(s, fill_tier, seed, rng, &fill_color, &fill_idx);
                if (fill_got) fill_row = res + 2 * slot;                    // an empty deck leaves the slot as it is
            }
            if (B(bank, GOLD) > 0 && rsum7(gems) <= 9) { SETB(gems, GOLD, B(gems, GOLD) + 1); SETB(bank, GOLD, B(bank, GOLD) - 1); }
        } else if (move < 30) {                                            // _buy_reserve :414-420
            const int i = move - 27;
            buy_card(s, bank, gems, s.row(res + 2 * i), s.row(res + 2 * i + 1), player);
            shift_i = i;
        } else if (move < 80) {                                            // _get_gems :436-444 / _give_gems :455-463
            constexpr uint32_t g3[5] = SPL_GEMS3_COL_INIT, g2[5] = SPL_GEMS2_COL_INIT;
            const bool take = move < 60;
            const int i = take ? move - 30 : move - 60, nd = take ? 25 : 15;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const int k = i < nd ? (int)(((take ? g3[c] : g2[c]) >> (i & 31)) & 1u) : (c == i - nd ? 2 : 0);
                SETB(bank, c, take ? B(bank, c) - k : B(bank, c) + k);
                SETB(gems, c, take ? B(gems, c) + k : B(gems, c) - k);
            }
        }
        SETB(bank, PTS, B(bank, PTS) + 1);                                 // move counter, int8 wrap :203
        s.set_row(0, bank);
        s.set_row(R_GEMS + player, gems);
        s.store(st);
        wave_sync();
        // ---- 14-byte card moves, lane-parallel in LDS ----
        if (copy_src >= 0 && l < 14) st[copy_dst * COLS + l] = st[copy_src * COLS + l];
        if (fill_row >= 0 && l < 14)
            st[fill_row * COLS + l] = fill_got ? SPL_CARDS[fill_tier][fill_color][fill_idx][l] : (int8_t)0;
        if (shift_i >= 0 && l < 6 * COLS) {                                // rows res+2i.. <- rows +2, last slot <- 0
            const int lo = 2 * shift_i * COLS;
            const int8_t v = (l >= lo && l < 4 * COLS) ? st[res * COLS + l + 2 * COLS] : (int8_t)0;
            if (l >= lo) st[res * COLS + l] = v;
        }
        wave_sync();
        return (player + 1) % NP;
    }

    __device__ static __forceinline__ int get_round(const int8_t* st) { return (uint8_t)st[PTS]; }      // :303-304

    __device__ static int get_score(const int8_t* st, int p) {                                          // :151-154
        int s = row(st, R_PCARDS + p)[PTS];
#pragma unroll
        for (int i = 0; i < NN; i++) s += row(st, R_PNOB + NN * p + i)[PTS];
        return s;
    }

    // Board.check_end_game :221-240.  Uniform across the wave (every lane evaluates the same LDS bytes).
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out /*[P]*/, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        int round = get_round(st);
#pragma unroll
        for (int p = 0; p < NP; p++) out[p] = 0.f;
        if (round % NP != 0) return false;
        float sc[NP];
        float mx = -1e30f;
#pragma unroll
        for (int p = 0; p < NP; p++) { sc[p] = (float)get_score(st, p); mx = sc[p] > mx ? sc[p] : mx; }
        if (!(mx >= 15.f || round >= MAX_MOVES)) return false;
        int cnt = 0;
#pragma unroll
        for (int p = 0; p < NP; p++) cnt += sc[p] == mx;
        bool several = cnt > 1;
        if (several) {
            mx = -1e30f;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                int nb = sum5(row(st, R_PCARDS + p));
                sc[p] = (float)((double)sc[p] - (double)nb / 100.);
                mx = sc[p] > mx ? sc[p] : mx;
            }
            cnt = 0;
#pragma unroll
            for (int p = 0; p < NP; p++) cnt += sc[p] == mx;
            several = cnt > 1;
        }
#pragma unroll
        for (int p = 0; p < NP; p++) out[p] = (sc[p] == mx) ? (several ? 0.01f : 1.f) : -1.f;
        return true;
    }

    // Board.swap_players :244-253 -- wave-cooperative, byte-parallel: dst[r] = src[rolled(r)]
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        for (int i = lane_id() + R_GEMS * COLS; i < S; i += 64) tmp[i] = st[i];      // only the per-player rows move
        wave_sync();
        for (int i = lane_id() + R_GEMS * COLS; i < S; i += 64) {
            int r = i / COLS, c = i - r * COLS, src;
            if (r < R_PNOB) src = R_GEMS + (r - R_GEMS + k) % NP;
            else if (r < R_PCARDS) src = R_PNOB + (r - R_PNOB + NN * k) % (NP * NN);
            else if (r < R_RES) src = R_PCARDS + (r - R_PCARDS + k) % NP;
            else src = R_RES + (r - R_RES + 6 * k) % (6 * NP);
            st[i] = tmp[src * COLS + c];
        }
        wave_sync();
    }

    // Board.get_symmetries :255-301 as an index map: candidate form c of the state / of the action vector.
    // Order: identity, 3 tiers x 3 permutations of the 4 visible cards, then per player the permutations of the filled
    // reserve slots (np_cards_symmetries / np_reserve_symmetries, SplendorLogic.py); reserve forms exist only when the
    // table entry is not -1.  sym_state_src returns the source BYTE index, sym_action_src the source action index.
    static constexpr int NSYM_CAND = 10 + 2 * NP;
    __device__ static __forceinline__ int n_reserved(const int8_t* st, int pl) {
        int nb = 3;
        for (int c = 2; c >= 0; c--) if (sum5(row(st, R_RES + 6 * pl + 2 * c)) == 0) nb = c;
        return nb;
    }
    __device__ static __forceinline__ bool sym_exists(const int8_t* st, int c) {
        if (c < 10) return true;
        const int pl = (c - 10) >> 1, s = (c - 10) & 1;
        return SPL_RESERVE_SYM[n_reserved(st, pl)][s][0] >= 0;
    }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int c, int i) {
        const int r = i / COLS, col = i - r * COLS;
        int src = r;
        if (c >= 1 && c < 10) {
            const int t = (c - 1) / 3, s = (c - 1) % 3, o = r - (1 + 8 * t);
            if (o >= 0 && o < 8) src = 1 + 8 * t + 2 * SPL_CARD_SYM[s][o >> 1] + (o & 1);
        } else if (c >= 10) {
            const int pl = (c - 10) >> 1, s = (c - 10) & 1, o = r - (R_RES + 6 * pl);
            if (o >= 0 && o < 6) src = R_RES + 6 * pl + 2 * SPL_RESERVE_SYM[n_reserved(st, pl)][s][o >> 1] + (o & 1);
        }
        return st[src * COLS + col];
    }
    __device__ static __forceinline__ int sym_action_src(const int8_t* st, int c, int a) {
        if (c >= 1 && c < 10) {
            const int t = (c - 1) / 3, s = (c - 1) % 3;
            if (a >= 4 * t && a < 4 * t + 4) return 4 * t + SPL_CARD_SYM[s][a - 4 * t];
            if (a >= 12 + 4 * t && a < 16 + 4 * t) return 12 + 4 * t + SPL_CARD_SYM[s][a - 12 - 4 * t];
        } else if (c >= 10 && ((c - 10) >> 1) == 0) {
            if (a >= 27 && a < 30) return 27 + SPL_RESERVE_SYM[n_reserved(st, 0)][(c - 10) & 1][a - 27];
        }
        return a;
    }

    // Board.init_game :156-175 -- lane 0 only; `st` must be zeroed by the caller.
    __device__ static void init_board(int8_t* st, Rng& rng) {
        int8_t* bank = row(st, 0);
        const int gems_in_play = NP == 2 ? 4 : (NP == 3 ? 5 : 7);
        for (int c = 0; c < 5; c++) bank[c] = (int8_t)gems_in_play;
        bank[GOLD] = 5;
        for (int t = 0; t < 3; t++) {
            int len = SPL_DECK_LEN[t];
            for (int c = 0; c < 5; c++) {
                row(st, 25 + 2 * t)[c] = (int8_t)len;
                row(st, 26 + 2 * t)[c] = (int8_t)(uint8_t)(0xFF00u >> len);
            }
        }
        for (int t = 0; t < 3; t++)
            for (int i = 0; i < 4; i++) fill_new_card(st, t, i, 0, rng);
        int perm[10];
        for (int i = 0; i < 10; i++) perm[i] = i;
        for (int i = 0; i < NN; i++) {
            int j = i + (int)(rng.u01() * (10 - i));
            j = j > 9 ? 9 : j;
            int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
            for (int c = 0; c < COLS; c++) row(st, 31 + i)[c] = SPL_NOBLES[perm[i]][c];
        }
    }
};

}  // namespace azg
