// game_splendor.cuh -- Splendor (2-4 players) env step for one wavefront, state staged in LDS.
//
// Semantics follow splendor/SplendorLogicNumba.py `Board` (line numbers cited); the byte layout of the state is the
// reference's int8[(32+10n+n*n)][7] (copy_state :207-219).  valid_mask() is lane-parallel and branch-free (one action per lane, two
// ballots for the 81 actions); make_move() is branchy integer work on ~400 LDS bytes and runs on lane 0.
#pragma once
#include "azg_common.cuh"
#include "splendor_tables.h"

namespace azg {

template <int NP>
struct SplendorDev {
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

    __device__ static void buy_card(int8_t* st, const int8_t* c0, const int8_t* c1, int player) {       // :344-357
        int8_t* bank = row(st, 0);
        int8_t* gems = row(st, R_GEMS + player);
        int8_t* cards = row(st, R_PCARDS + player);
        int miss = missing(c0, gems, cards);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            int8_t need = (int8_t)(c0[c] - cards[c]);
            need = need < 0 ? (int8_t)0 : need;
            int8_t paid = need < gems[c] ? need : gems[c];
            gems[c] -= paid;
            bank[c] += paid;
        }
        gems[GOLD] = (int8_t)(gems[GOLD] - miss);
        bank[GOLD] = (int8_t)(bank[GOLD] + miss);
#pragma unroll
        for (int c = 0; c < COLS; c++) cards[c] += c1[c];
        for (int i = 0; i < NN; i++) {                                     // _give_nobles_if_earned :465-470
            int8_t* noble = row(st, R_NOBLES + i);
            if (sum5(noble) <= 0) continue;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 5; c++) ok = ok && cards[c] >= noble[c];
            if (ok) {
                int8_t* dst = row(st, R_PNOB + NN * player + i);
#pragma unroll
                for (int c = 0; c < COLS; c++) { dst[c] = noble[c]; noble[c] = 0; }
            }
        }
    }

    // Board.make_move :190-205 -- lane 0 only
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        int8_t* bank = row(st, 0);
        int8_t* gems = row(st, R_GEMS + player);
        if (move < 12) {                                                   // _buy :370-373
            int8_t c[14];
            const int8_t* src = row(st, 1 + 2 * move);
#pragma unroll
            for (int i = 0; i < 14; i++) c[i] = src[i];
            buy_card(st, c, c + 7, player);
            fill_new_card(st, move >> 2, move & 3, seed, rng);
        } else if (move < 27) {                                            // _reserve :382-400
            int i = move - 12;
            int8_t* res = row(st, R_RES + 6 * player);
            int slot = 2;
            for (int s = 2; s >= 0; s--) if (sum5(res + 2 * s * COLS) == 0) slot = s;
            int8_t* dst = res + 2 * slot * COLS;
            if (i < 12) {
                const int8_t* src = row(st, 1 + 2 * i);
#pragma unroll
                for (int k = 0; k < 14; k++) dst[k] = src[k];
                fill_new_card(st, i >> 2, i & 3, seed, rng);
            } else {
                int8_t card[14];
                if (get_deck_card(st, i - 12, seed, rng, card)) {
#pragma unroll
                    for (int k = 0; k < 14; k++) dst[k] = card[k];
                }
            }
            if (bank[GOLD] > 0 && sum7(gems) <= 9) { gems[GOLD] += 1; bank[GOLD] -= 1; }
        } else if (move < 30) {                                            // _buy_reserve :414-420
            int i = move - 27;
            int8_t* res = row(st, R_RES + 6 * player);
            int8_t c[14];
#pragma unroll
            for (int k = 0; k < 14; k++) c[k] = res[2 * i * COLS + k];
            buy_card(st, c, c + 7, player);
            for (int k = 2 * i * COLS; k < 4 * COLS; k++) res[k] = res[k + 2 * COLS];   // shift towards slot 0
            for (int k = 4 * COLS; k < 6 * COLS; k++) res[k] = 0;
        } else if (move < 60) {                                            // _get_gems :436-444
            int i = move - 30;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                int8_t k = i < 25 ? SPL_GEMS3[i < 25 ? i : 0][c] : (int8_t)(c == i - 25 ? 2 : 0);
                bank[c] -= k; gems[c] += k;
            }
        } else if (move < 80) {                                            // _give_gems :455-463
            int i = move - 60;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                int8_t k = i < 15 ? SPL_GEMS2[i < 15 ? i : 0][c] : (int8_t)(c == i - 15 ? 2 : 0);
                bank[c] += k; gems[c] -= k;
            }
        }
        bank[PTS] = (int8_t)(bank[PTS] + 1);                               // move counter, int8 wrap :203
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
        for (int i = lane_id(); i < S; i += 64) tmp[i] = st[i];
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
