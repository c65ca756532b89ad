// game_abalone.hip.h -- Abalone env step on the device plugin interface: abalone/AbaloneLogicNumba.py
// (Board :166-440) as shipped: INITIAL_LAYOUT = 1 (Belgian Daisy), ENABLE_DYNAMIC_KOMI = False (:5-6).
//
// State int8 [9][9][4] on an axial hex grid (playable cells: 4 <= r + q <= 12): plane 0 marbles of player 0, plane 1 of player 1,
// plane 2 the board mask, plane 3 misc with misc[0][0..2] = the two scores and the move counter; byte = (r*9 + q)*4 + plane.
// 3402 actions = anchor cell x 42 planes: one marble in direction d (6), two / three marbles along `axis` moved in direction d
// (3 x 6 each), the anchor being the marble with the smallest (r, q) (:66-88).  Deterministic: make_move ignores random_seed, so
// a resolved edge is memoised in every universe slot.
//
// valid_moves is restated as a predicate of ONE action id (the reference enumerates cells, group sizes and directions with
// early exits, :271-356), evaluated one action per lane in 54 ballots; make_move / init run on lane 0 over the LDS state.
#pragma once
#include "azg_common.hip.h"

namespace azg {

struct AbaloneDev {
    static constexpr int P = 2;
    static constexpr int ROWS = 81, COLS = 4;
    static constexpr int S = 324;
    static constexpr int SP = RoundUp16<S>::value;
    static constexpr int A = 3402;
    static constexpr int AW = (A + 63) / 64;
    static constexpr bool STOCHASTIC = false;
    static constexpr bool RANDOM_SYM = false;   // get_symmetries draws no randomness

    __device__ static __forceinline__ int dr(int d) { return d == 1 || d == 2 ? 1 : (d == 4 || d == 5 ? -1 : 0); }
    __device__ static __forceinline__ int dq(int d) { return d == 0 || d == 5 ? 1 : (d == 2 || d == 3 ? -1 : 0); }      // DIRECTIONS :58-65
    __device__ static __forceinline__ int cell(const int8_t* st, int r, int q, int z) { return st[((r * 9 + q) << 2) + z]; }
    __device__ static __forceinline__ int8_t& cellw(int8_t* st, int r, int q, int z) { return st[((r * 9 + q) << 2) + z]; }
    __device__ static __forceinline__ bool on_board(const int8_t* st, int r, int q) {                                      // :90-94
        return r >= 0 && r < 9 && q >= 0 && q < 9 && cell(st, r, q, 2) == 1;
    }
    struct Act { int r, q, size, axis, d; };
    __device__ static __forceinline__ Act decode(int a) {                                                                  // :76-88
        const int plane = a % 42;
        Act x;
        x.q = (a / 42) % 9; x.r = a / 378; x.d = plane % 6;
        x.size = plane < 6 ? 1 : (plane < 24 ? 2 : 3);
        x.axis = plane < 6 ? 0 : (plane < 24 ? (plane - 6) / 6 : (plane - 24) / 6);
        return x;
    }
    __device__ static __forceinline__ int encode(int r, int q, int size, int axis, int d) {                                // :67-74
        return r * 378 + q * 42 + (size == 1 ? d : (size == 2 ? 6 + axis * 6 + d : 24 + axis * 6 + d));
    }

    // Board.valid_moves restricted to one action (:271-356)
    __device__ static bool valid_action(const int8_t* st, int a, int player) {
        const Act x = decode(a);
        const int opp = 1 - player, r = x.r, q = x.q, d = x.d, ax = x.axis;
        if (cell(st, r, q, player) == 0) return false;
        if (x.size == 1) {
            const int nr = r + dr(d), nq = q + dq(d);
            return on_board(st, nr, nq) && cell(st, nr, nq, player) == 0 && cell(st, nr, nq, opp) == 0;
        }
        const int r1 = r + dr(ax), q1 = q + dq(ax);
        if (!on_board(st, r1, q1) || cell(st, r1, q1, player) == 0) return false;
        if (x.size == 3) {
            const int r2 = r1 + dr(ax), q2 = q1 + dq(ax);
            if (!(on_board(st, r2, q2) && cell(st, r2, q2, player) == 1)) return false;
        }
        const bool inl = d == ax || d == (ax + 3) % 6;
        if (!inl) {                                                                                                        // broadside
            for (int i = 0; i < x.size; i++) {
                const int tr = r + i * dr(ax) + dr(d), tq = q + i * dq(ax) + dq(d);
                if (!on_board(st, tr, tq) || cell(st, tr, tq, player) == 1 || cell(st, tr, tq, opp) == 1) return false;
            }
            return true;
        }
        const int fr = d == ax ? r + (x.size - 1) * dr(ax) : r, fq = d == ax ? q + (x.size - 1) * dq(ax) : q;
        const int tr = fr + dr(d), tq = fq + dq(d);
        if (!on_board(st, tr, tq)) return false;
        if (cell(st, tr, tq, player) == 1) return false;
        if (cell(st, tr, tq, opp) == 0) return true;
        int opp_count = 0, cr = tr, cq = tq;                                                                               // sumito
        for (int it = 0; it < 4; it++) {
            if (!on_board(st, cr, cq)) return opp_count > 0;
            if (cell(st, cr, cq, opp) == 1) {
                opp_count++;
                if (opp_count >= x.size) return false;
                cr += dr(d); cq += dq(d);
            } else return cell(st, cr, cq, player) != 1;
        }
        return false;
    }
    // An action starts at a cell that holds one of the player's own marbles (:271-356 skip every other cell first): at most 14 of the 81
    // cells.  The wave compacts those cells (two ballots over the board), then evaluates 64 (own cell, plane) pairs per pass -- <= 10
    // passes instead of the 54 over all 3402 action ids -- and every valid lane sets its action's bit in the mask.
    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
        for (int k = l; k < AW; k += 64) mask_lds[k] = 0ull;
        const uint64_t b0 = __ballot(st[(l << 2) + player] != 0);                          // cells 0..63 (cell c = r * 9 + q, 4 bytes each)
        const uint64_t b1 = __ballot(l < 81 - 64 && st[((l + 64) << 2) + player] != 0);    // cells 64..80
        const int n_own = __popcll(b0) + __popcll(b1);
        int mycell = 0;                                                                     // lane i: the i-th own cell
        {
            uint64_t w0 = b0, w1 = b1;
            for (int i = 0; i < n_own; i++) {
                int c;
                if (w0) { c = __builtin_ctzll(w0); w0 &= w0 - 1; } else { c = 64 + __builtin_ctzll(w1); w1 &= w1 - 1; }
                if (l == i) mycell = c;
            }
        }
        wave_sync();
        const int n_pairs = n_own * 42;
#pragma unroll 1
        for (int base = 0; base < n_pairs; base += 64) {
            const int x = base + l, ci = x / 42, plane = x - ci * 42;
            const int c = __shfl(mycell, ci < 64 ? ci : 0, 64);
            const int a = c * 42 + plane;
            if (x < n_pairs && valid_action(st, a, player))
                __hip_atomic_fetch_or((unsigned long long*)&mask_lds[a >> 6], 1ull << (a & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_sync();
    }

    __device__ static __forceinline__ bool move_uses_seed(int) { return false; }
    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        return lane0_make_move<AbaloneDev>(st, move, player, seed, rng);
    }
    // Board.make_move :358-396 -- lane 0 only
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        (void)seed; (void)rng;
        const Act x = decode(move);
        const int opp = 1 - player, r = x.r, q = x.q, d = x.d, ax = x.axis;
        const bool inl = d == ax || d == (ax + 3) % 6;
        if (x.size == 1 || !inl) {
            for (int i = 0; i < x.size; i++) {
                const int cr = x.size > 1 ? r + i * dr(ax) : r, cq = x.size > 1 ? q + i * dq(ax) : q;
                cellw(st, cr, cq, player) = 0;
                cellw(st, cr + dr(d), cq + dq(d), player) = 1;
            }
        } else {
            const int fr = d == ax ? r + (x.size - 1) * dr(ax) : r, fq = d == ax ? q + (x.size - 1) * dq(ax) : q;
            const int br = d == ax ? r : r + (x.size - 1) * dr(ax), bq = d == ax ? q : q + (x.size - 1) * dq(ax);
            const int tr = fr + dr(d), tq = fq + dq(d);
            if (on_board(st, tr, tq) && cell(st, tr, tq, opp) == 1) {
                int cr = tr, cq = tq;
                while (on_board(st, cr, cq) && cell(st, cr, cq, opp) == 1) { cr += dr(d); cq += dq(d); }
                cellw(st, tr, tq, opp) = 0;
                if (on_board(st, cr, cq)) cellw(st, cr, cq, opp) = 1;
                else cellw(st, 0, player, 3) = (int8_t)(cell(st, 0, player, 3) + 1);
            }
            cellw(st, br, bq, player) = 0;
            cellw(st, tr, tq, player) = 1;
        }
        cellw(st, 0, 2, 3) = (int8_t)(cell(st, 0, 2, 3) + 1);
        return 1 - player;
    }

    __device__ static __forceinline__ int get_round(const int8_t* st) { return cell(st, 0, 2, 3); }                       // :262-263
    __device__ static __forceinline__ int get_score(const int8_t* st, int p) { return cell(st, 0, p == 0 ? 0 : 1, 3); }   // :265-266
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return (int)(uint8_t)cell(st, 0, 2, 3); }            // every move adds 1

    // Board.check_end_game :398-413 (uniform)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        const int s0 = cell(st, 0, 0, 3), s1 = cell(st, 0, 1, 3);
        out[0] = out[1] = 0.f;
        if (s0 >= 6) { out[0] = 1.f; out[1] = -1.f; return true; }
        if (s1 >= 6) { out[0] = -1.f; out[1] = 1.f; return true; }
        if (cell(st, 0, 2, 3) >= 127) {
            if (s0 > s1) { out[0] = 1.f; out[1] = -1.f; }
            else if (s1 > s0) { out[0] = -1.f; out[1] = 1.f; }
            else { out[0] = out[1] = 0.001f; }
            return true;
        }
        return false;
    }

    // Board.swap_players :415-426 -- wave-cooperative: the two marble planes and the two scores trade places
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        (void)tmp;
        if (k % 2 != 1) return;
        for (int c = lane_id(); c < 81; c += 64) { const int8_t t = st[4 * c]; st[4 * c] = st[4 * c + 1]; st[4 * c + 1] = t; }
        wave_sync();
        if (lane_id() == 0) { const int8_t t = st[3]; st[3] = st[7]; st[7] = t; }
        wave_sync();
    }

    // init_game :175-222 (layout 1) -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        (void)rng;
        for (int r = 0; r < 9; r++)
            for (int q = 0; q < 9; q++)
                if (r + q >= 4 && r + q <= 12) cellw(st, r, q, 2) = 1;
        // rows {0, 1, 2, 6, 7, 8}: [q_lo, q_hi) of the opponent's and of player 0's marbles
        const int rows[6] = {0, 1, 2, 6, 7, 8};
        const int olo[6] = {4, 3, 3, 4, 3, 3}, ohi[6] = {6, 6, 5, 6, 6, 5};
        const int mlo[6] = {7, 6, 6, 1, 0, 0}, mhi[6] = {9, 9, 8, 3, 3, 2};
        for (int i = 0; i < 6; i++) {
            for (int q = olo[i]; q < ohi[i]; q++) cellw(st, rows[i], q, 1) = 1;
            for (int q = mlo[i]; q < mhi[i]; q++) cellw(st, rows[i], q, 0) = 1;
        }
    }

    // ---- get_symmetries :428-460: 6 rotations x 2 flips (always 12 forms, identity first) ----
    static constexpr int NSYM_CAND = 12;
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int) { return true; }
    // the INVERSE of form c = (rot, flip) on a cell: forward = rotate^rot after flip, so inverse = flip after rotate^(6 - rot)
    __device__ static __forceinline__ void inv_cell(int c, int& r, int& q) {
        const int rot = c >> 1, flip = c & 1;
        for (int k = 0; k < (6 - rot) % 6; k++) { const int a = q + r - 4, b = 8 - r; r = a; q = b; }
        if (flip) q = 12 - r - q;
    }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int c, int i) {
        const int z = i & 3;
        if (z == 3) return st[i];                                     // the misc layer is copied untransformed
        int r = (i >> 2) / 9, q = (i >> 2) % 9;
        if (!(r + q >= 4 && r + q <= 12)) return 0;                   // only board cells are written
        inv_cell(c, r, q);
        return (r >= 0 && r < 9 && q >= 0 && q < 9) ? st[((r * 9 + q) << 2) + z] : (int8_t)0;
    }
    // source action of output action a under form c (ACTION_SYMMETRIES[rot][flip][src] == a, :99-146); -1 = none on the grid
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int c, int a) {
        const int rot = c >> 1, flip = c & 1;
        const Act x = decode(a);
        int mr[3], mq[3];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            mr[i] = x.r + (i < x.size ? i : 0) * dr(x.axis); mq[i] = x.q + (i < x.size ? i : 0) * dq(x.axis);
            inv_cell(c, mr[i], mq[i]);
            ok = ok && mr[i] >= 0 && mr[i] < 9 && mq[i] >= 0 && mq[i] < 9;
        }
        if (!ok) return -1;
        int mi = 0;
        for (int i = 1; i < x.size; i++)
            if (mr[i] < mr[mi] || (mr[i] == mr[mi] && mq[i] < mq[mi])) mi = i;
        int axis = 0;
        if (x.size > 1) {
            const int oi = mi == 0 ? 1 : 0, ddr = mr[oi] - mr[mi], ddq = mq[oi] - mq[mi];
            axis = (ddr == 0 && ddq > 0) ? 0 : ((ddr > 0 && ddq == 0) ? 1 : ((ddr > 0 && ddq < 0) ? 2 : 0));
        }
        int d = (x.d - rot + 6) % 6;
        if (flip) d = d < 4 ? 3 - d : 9 - d;                          // [3, 2, 1, 0, 5, 4] is its own inverse
        return encode(mr[mi], mq[mi], x.size, axis, d);
    }
};

}  // namespace azg
