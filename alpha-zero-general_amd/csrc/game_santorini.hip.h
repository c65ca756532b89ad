// game_santorini.hip.h -- Santorini 5x5 (NB_GODS = 1: no gods, 11: basic gods) env step for one wavefront.
//
// Semantics follow santorini/SantoriniLogicNumba.py `Board` + SantoriniConstants.py (lines cited).  State bytes are the
// reference's int8[5][5][3] interleaved (workers, levels, gods_power).  The reference enumerates valid moves with nested
// worker x move x build loops per god (:125-432); here every lane evaluates the SAME rules as a predicate of ONE
// action id (worker, power, move_dir, build_dir), 64 actions per pass, and the passes' ballots are the mask words.
#pragma once
#include "azg_common.hip.h"

namespace azg {

template <int NB>
struct SantoriniDev {
    static constexpr bool STOCHASTIC = false;   // the env step is a function of (state, action, random_seed): edges are memoised
    static constexpr bool RANDOM_SYM = false;   // get_symmetries draws no randomness
    static constexpr bool ENDED_FILLS_MASK = true;   // game_ended(st, p, ., mask) leaves valid_mask(st, p) in `mask` when the game goes on
    static constexpr int P = 2;
    static constexpr int ROWS = 25, COLS = 3;
    static constexpr int S = 75;
    static constexpr int SP = 80;
    static constexpr int A = NB * 2 * 81;                    // action_size :17-19
    static constexpr int AW = (A + 63) / 64;
    enum { NO_GOD = 0, APOLLO, MINOTAUR, ATLAS, HEPHAESTUS, ARTEMIS, DEMETER, HERMES, PAN, ATHENA, PROMETHEUS };
    static constexpr int NO_MOVE = 4, NO_BUILD = 4, MAX_ITER_FOR_HERMES = 5;

    __device__ static __forceinline__ int W(const int8_t* st, int pos) { return st[pos * 3]; }
    __device__ static __forceinline__ int LV(const int8_t* st, int pos) { return st[pos * 3 + 1]; }
    __device__ static __forceinline__ int GP(const int8_t* st, int i) { return st[i * 3 + 2]; }

    struct Pos { int y, x; };
    __device__ static __forceinline__ Pos dir(Pos p, int d) { return Pos{p.y + d / 3 - 1, p.x + d % 3 - 1}; }   // :56-70
    __device__ static __forceinline__ bool in_grid(Pos p) { return p.y >= 0 && p.y < 5 && p.x >= 0 && p.x < 5; }
    __device__ static __forceinline__ int idx(Pos p) { return p.y * 5 + p.x; }

    __device__ static Pos worker_pos(const int8_t* st, int id) {                                   // :667-673
        int f = 0;
#pragma unroll
        for (int i = 24; i >= 0; i--) f = (W(st, i) == id) ? i : f;
        return Pos{f / 5, f % 5};
    }
    // The same for all four workers at once, wave-parallel: lane i < 25 reads cell i, one ballot per worker id, lowest set bit = the
    // first cell holding it (0 when there is none, as the reference's scan).  wp[0..1] = player 0's workers 1, 2; wp[2..3] = player 1's.
    struct Workers { int cell[4]; };
    __device__ static __forceinline__ Workers find_workers(const int8_t* st) {
        const int l = lane_id();
        const int w = l < 25 ? W(st, l) : 0;
        Workers k;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int id = (i & 1) + 1, sgn = i < 2 ? 1 : -1;
            const uint64_t m = __ballot(w == id * sgn);
            k.cell[i] = m ? first_lane(m) : 0;
        }
        return k;
    }
    __device__ static __forceinline__ Pos worker_of(const Workers& k, int player, int worker) {
        // (a select chain, not k.cell[2 * player + worker]: a dynamically indexed private array lives in scratch memory, and every
        // byte of scratch a wave touches is written back to HBM once 4096 waves' worth no longer fits the L2)
        const int a = player ? k.cell[2] : k.cell[0], b = player ? k.cell[3] : k.cell[1];
        const int c = worker ? b : a;
        return Pos{c / 5, c - 5 * (c / 5)};
    }

    __device__ static bool able_to_move(const int8_t* st, Pos old, Pos np, int player, bool no_climb, bool swap,
                                        bool push) {                                               // :675-716
        if (old.y == np.y && old.x == np.x) return true;
        if (!in_grid(np)) return false;
        const int w = W(st, idx(np));
        if (w != 0) {
            const bool is_opp = player == 0 ? (w == -1 || w == -2) : (w == 1 || w == 2);
            if ((swap || push) && is_opp) {
                if (push) {
                    Pos pp{2 * np.y - old.y, 2 * np.x - old.x};                                    // :52-53
                    if (!in_grid(pp)) return false;
                    if (W(st, idx(pp)) != 0) return false;
                    if (LV(st, idx(pp)) > 3) return false;
                }
            } else return false;
        }
        const int nl = LV(st, idx(np));
        if (nl > 3) return false;
        return !(nl > LV(st, idx(old)) + (no_climb ? 0 : 1));
    }

    __device__ static bool able_to_build(const int8_t* st, Pos p, int ignore, bool two_levels, bool dome) {   // :718-729
        if (!in_grid(p)) return false;
        const int w = W(st, idx(p));
        if (!(w == 0 || w == ignore)) return false;
        return !(LV(st, idx(p)) >= (two_levels ? 2 : (dome ? 3 : 4)));
    }

    __device__ static __forceinline__ int owned_god(const int8_t* st, int player) {
        int god = -1;
#pragma unroll
        for (int k = PROMETHEUS; k >= 0; k--) god = (GP(st, k + NB * player) > 0) ? k : god;      // first of the elif chain
        return god;
    }

    // Board.valid_moves (:125-432) as a predicate of one action
    __device__ static bool valid_action(const int8_t* st, int a, int player, int god, bool opp_athena, const Workers& wk) {
        const int worker = a / (NB * 81);
        int rem = a - worker * NB * 81;
        const int power = rem / 81; rem -= power * 81;
        const int md = rem / 9, bd = rem - md * 9;
        if (god < 0) return false;
        if (power != NO_GOD && power != god) return false;
        const int wid = (worker + 1) * (player == 0 ? 1 : -1);
        const Pos old = worker_of(wk, player, worker);
        const Pos np = dir(old, md);
        const Pos bp = dir(np, bd);
        if (NB == 1) {          // no gods: power 0 is the only one (the switch below would keep ten gods' rules and registers alive)
            if (md == NO_MOVE || bd == NO_BUILD) return false;
            return able_to_move(st, old, np, player, opp_athena, false, false) && able_to_build(st, bp, wid, false, false);
        }
        switch (god) {
        case NO_GOD: case PAN: case ATHENA: {
            if (power != NO_GOD || md == NO_MOVE || bd == NO_BUILD) return false;
            const bool nc = god == ATHENA ? false : opp_athena;                                    // :383
            return able_to_move(st, old, np, player, nc, false, false) && able_to_build(st, bp, wid, false, false);
        }
        case APOLLO: case MINOTAUR: {                                                               // :154-198
            if (md == NO_MOVE || bd == NO_BUILD) return false;
            bool use_power = false;
            if (!able_to_move(st, old, np, player, opp_athena, false, false)) {
                if (!able_to_move(st, old, np, player, opp_athena, god == APOLLO, god == MINOTAUR)) return false;
                use_power = true;
            }
            if (power != (use_power ? god : (int)NO_GOD)) return false;
            return able_to_build(st, bp, wid, false, false);
        }
        case ATLAS: case HEPHAESTUS: {                                                              // :200-239
            if (md == NO_MOVE || bd == NO_BUILD) return false;
            if (!able_to_move(st, old, np, player, opp_athena, false, false)) return false;
            if (power == NO_GOD) return able_to_build(st, bp, wid, false, false);
            return able_to_build(st, bp, wid, god == HEPHAESTUS, god == ATLAS);
        }
        case ARTEMIS: {                                                                             // :242-281
            const int avoid = GP(st, ARTEMIS + NB * player) % 64 - 1;
            if (md == NO_MOVE) return false;
            if (avoid < 0) {
                if (!able_to_move(st, old, np, player, opp_athena, false, false)) return false;
                if (bd == NO_BUILD) return power == ARTEMIS;
                return power == NO_GOD && able_to_build(st, bp, wid, false, false);
            }
            if (worker != avoid / 9 || power != NO_GOD || md == avoid % 9 || bd == NO_BUILD) return false;
            return able_to_move(st, old, np, player, opp_athena, false, false) && able_to_build(st, bp, wid, false, false);
        }
        case DEMETER: {                                                                             // :284-318
            const int avoid = GP(st, DEMETER + NB * player) % 64 - 1;
            if (avoid < 0) {
                if (power != DEMETER || md == NO_MOVE || bd == NO_BUILD) return false;
                return able_to_move(st, old, np, player, opp_athena, false, false) &&
                       able_to_build(st, bp, wid, false, false);
            }
            if (worker != avoid / 9 || power != NO_GOD || md != NO_MOVE) return false;
            if (bd == NO_BUILD) return true;                                                        // cancel 2nd turn
            if (bd == avoid % 9) return false;
            return able_to_build(st, dir(old, bd), wid, false, false);
        }
        case HERMES: {                                                                              // :321-351
            const int nb_prev = GP(st, HERMES + NB * player) % 64;
            if (power == NO_GOD && md == NO_MOVE)
                return bd != NO_BUILD && able_to_build(st, dir(old, bd), wid, false, false);
            if (power == HERMES)
                return md != NO_MOVE && bd == NO_BUILD && nb_prev < MAX_ITER_FOR_HERMES &&
                       able_to_move(st, old, np, player, opp_athena, false, false) &&
                       LV(st, idx(np)) == LV(st, idx(old));
            // classic turn
            return md != NO_MOVE && bd != NO_BUILD && nb_prev == 0 &&
                   able_to_move(st, old, np, player, opp_athena, false, false) &&
                   able_to_build(st, bp, wid, false, false);
        }
        case PROMETHEUS: {                                                                          // :392-428
            const int v = GP(st, PROMETHEUS + NB * player) % 64 - 1;
            const int prev = v < 0 ? -1 : v / 9;
            if (bd == NO_BUILD) return false;
            if (prev < 0) {
                const bool use_power = md == NO_MOVE;
                if (power != (use_power ? (int)PROMETHEUS : (int)NO_GOD)) return false;
                return able_to_move(st, old, np, player, opp_athena, false, false) &&
                       able_to_build(st, bp, wid, false, false);
            }
            if (worker != prev || power != NO_GOD || md == NO_MOVE) return false;
            return able_to_move(st, old, np, player, true, false, false) && able_to_build(st, bp, wid, false, false);
        }
        }
        return false;
    }

    // the valid-move scan for `player`: fills mask_lds (may be null) and tells whether any action is valid
    __device__ static bool scan_valid(const int8_t* st, int player, int god, bool opp_athena, const Workers& wk, uint64_t* mask_lds) {
        const int l = lane_id();
        bool any = false;
        if constexpr (NB == 1) {
            for (int k = 0; k < AW; k++) {
                const int a = k * 64 + l;
                const uint64_t m = __ballot(a < A && valid_action(st, a < A ? a : 0, player, god, opp_athena, wk));
                if (mask_lds && l == 0) mask_lds[k] = m;
                any = any || m != 0ull;
            }
        } else {
            // with gods: an action is (worker, power, move, build) and only the powers "none" and the player's own god can be valid, so
            // 2 workers x 2 powers x 81 = 324 of the 1782 actions are evaluated (8 passes of the wave instead of 28) and their bits are
            // placed into the mask words by lane 0 (a block of 81 does not start on a word boundary)
            if (mask_lds) {
                if (l < AW) mask_lds[l] = 0ull;
                wave_sync();
            }
            if (god >= 0) {
                for (int worker = 0; worker < 2; worker++)
                    for (int pi = 0; pi < (god == NO_GOD ? 1 : 2); pi++) {
                        const int base = (worker * NB + (pi == 0 ? (int)NO_GOD : god)) * 81;
                        for (int p = 0; p < 2; p++) {
                            const int i = 64 * p + l;
                            const uint64_t m = __ballot(i < 81 && valid_action(st, base + (i < 81 ? i : 0), player, god, opp_athena, wk));
                            any = any || m != 0ull;
                            if (mask_lds && l == 0 && m) {
                                const int first = base + 64 * p, w = first >> 6, o = first & 63;
                                mask_lds[w] |= m << o;
                                if (o && (m >> (64 - o))) mask_lds[w + 1] |= m >> (64 - o);
                            }
                        }
                    }
            }
            if (mask_lds) wave_sync();
        }
        return any;
    }
    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int god = owned_god(st, player);
        const bool opp_athena = GP(st, ATHENA + NB * ((player + 1) % 2)) > 64;                      // :133
        const Workers wk = find_workers(st);
        scan_valid(st, player, god, opp_athena, wk, mask_lds);
    }
    // Board.get_symmetries :578-653: identity, rot90 x1..3 (np.rot90: out[i][j] = in[j][4-i]), fliplr, flipud, swap of
    // the own workers, swap of the opponent's workers.  The Artemis / Demeter memo (65 + 9*worker + direction) follows
    // the direction permutation (:589-595) and, with Athena's, the worker swap (:630-636); the policy follows the
    // (move, build) direction permutation (:597-609) or swaps its two worker halves (:641-642).
    static constexpr int NSYM_CAND = 8;
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int) { return true; }
    __device__ static __forceinline__ int dir_core(int kind, int d) {        // SantoriniConstants.py:60,68,77
        const int dr = d / 3, dc = d - 3 * dr;
        if (kind == 0) return 3 * (2 - dc) + dr;         // rotation   [6,3,0,7,4,1,8,5,2]
        if (kind == 1) return 3 * dr + (2 - dc);         // flip LR    [2,1,0,5,4,3,8,7,6]
        return 3 * (2 - dr) + dc;                        // flip UD    [6,7,8,3,4,5,0,1,2]
    }
    __device__ static __forceinline__ int dir_core_inv(int kind, int d) {
        if (kind != 0) return dir_core(kind, d);         // involutions
        return dir_core(0, dir_core(0, dir_core(0, d))); // rotation^-1 = rotation^3
    }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int c, int i) {
        const int pos = i / 3, plane = i - 3 * pos;
        if (plane < 2) {
            int si = pos / 5, sj = pos - 5 * si;
            if (c >= 1 && c <= 3) { for (int k = 0; k < c; k++) { const int t = si; si = sj; sj = 4 - t; } }
            else if (c == 4) sj = 4 - sj;
            else if (c == 5) si = 4 - si;
            int8_t v = st[(si * 5 + sj) * 3 + plane];
            if (plane == 0 && c == 6 && v > 0) v = (int8_t)(3 - v);
            if (plane == 0 && c == 7 && v < 0) v = (int8_t)(-3 - v);
            return v;
        }
        int v = st[i];                                   // gods_power.flat[pos]
        if (pos >= 2 * NB || v < 65) return (int8_t)v;
        const int god = pos % NB;
        if (c >= 1 && c <= 5 && (god == ARTEMIS || god == DEMETER)) {
            const int k = v - 65, worker = k / 9;
            int dir = k - 9 * worker;
            if (c <= 3) { for (int r = 0; r < c; r++) dir = dir_core(0, dir); }
            else dir = dir_core(c == 4 ? 1 : 2, dir);
            return (int8_t)(65 + 9 * worker + dir);
        }
        if ((c == 6 && pos < NB) || (c == 7 && pos >= NB)) {
            if (god == ARTEMIS || god == DEMETER || god == ATHENA) return (int8_t)((v - 65 + 9) % 18 + 65);
        }
        return (int8_t)v;
    }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int c, int a) {
        if (c == 0 || c == 7) return a;
        if (c == 6) return (a + A / 2) % A;
        const int worker = a / (NB * 81);
        int rem = a - worker * (NB * 81);
        const int power = rem / 81;
        rem -= power * 81;
        int md = rem / 9, bd = rem - 9 * md;
        if (c <= 3) { for (int r = 0; r < c; r++) { md = dir_core_inv(0, md); bd = dir_core_inv(0, bd); } }
        else { md = dir_core_inv(c == 4 ? 1 : 2, md); bd = dir_core_inv(c == 4 ? 1 : 2, bd); }
        return NB * 81 * worker + 81 * power + 9 * md + bd;
    }

    // monotone "age" of a state for the clean-up: the (saturating) move counter
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return get_round(st); }

    // no chance events in Santorini: random_seed is never read by make_move (:434-550)
    __device__ static __forceinline__ bool move_uses_seed(int) { return false; }

    // Board.make_move :434-550 for the whole wave: the moving worker is located by one ballot over the 25 cells (every lane then holds
    // the same positions in scalar registers), the handful of byte updates of the move are lane 0's stores
    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        (void)seed; (void)rng;
        const int worker = move / (NB * 81);
        const Workers wk = find_workers(st);
        const Pos old = worker_of(wk, player, worker);
        int np_ = 0;
        if (lane_id() == 0) np_ = make_move_at(st, move, player, old);
        np_ = __builtin_amdgcn_readfirstlane(np_);
        wave_sync();
        return np_;
    }

    // Board.make_move :434-550 -- host-style entry (one thread): locates the worker by the reference's scan
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        (void)seed; (void)rng;
        const int worker = move / (NB * 81);
        const int wid = (worker + 1) * (player == 0 ? 1 : -1);
        return make_move_at(st, move, player, worker_pos(st, wid));
    }
    __device__ static int make_move_at(int8_t* st, int move, int player, const Pos old) {
        const int worker = move / (NB * 81);
        int rem = move - worker * NB * 81;
        const int power = rem / 81; rem -= power * 81;
        const int md = rem / 9, bd = rem - md * 9;
        const int wid = (worker + 1) * (player == 0 ? 1 : -1);
        const Pos np = dir(old, md);
        const int io = idx(old) * 3, in = idx(np) * 3;
        bool opp_next = true;
        switch (power) {
        case NO_GOD: {
            const int old_level = st[io + 1];
            st[io] = 0; st[in] = (int8_t)wid;
            if (bd != NO_BUILD) st[idx(dir(np, bd)) * 3 + 1] += 1;
            if (GP(st, PAN + NB * player) > 0) {
                if (st[in + 1] <= old_level - 2) st[(PAN + NB * player) * 3 + 2] = 65;
            } else if (GP(st, ATHENA + NB * player) > 0) {
                st[(ATHENA + NB * player) * 3 + 2] = (int8_t)(64 + (st[in + 1] > old_level ? 1 : 0));
            } else {
                for (int i = player * NB; i < (player + 1) * NB; i++)
                    if (st[i * 3 + 2] > 64) st[i * 3 + 2] = 64;
            }
            break; }
        case APOLLO: {
            const int8_t a = st[io], b = st[in];
            st[io] = b; st[in] = a;
            st[idx(dir(np, bd)) * 3 + 1] += 1;
            break; }
        case MINOTAUR: {
            const Pos pp{2 * np.y - old.y, 2 * np.x - old.x};
            const int8_t a = st[io], b = st[in];
            st[io] = 0; st[in] = a; st[idx(pp) * 3] = b;
            st[idx(dir(np, bd)) * 3 + 1] += 1;
            break; }
        case ATLAS:
            st[io] = 0; st[in] = (int8_t)wid;
            st[idx(dir(np, bd)) * 3 + 1] = 4;
            break;
        case HEPHAESTUS:
            st[io] = 0; st[in] = (int8_t)wid;
            st[idx(dir(np, bd)) * 3 + 1] += 2;
            break;
        case ARTEMIS:
            st[io] = 0; st[in] = (int8_t)wid;
            st[(ARTEMIS + NB * player) * 3 + 2] = (int8_t)(64 + (worker * 9 + (8 - md) + 1));
            opp_next = false;
            break;
        case DEMETER:
            st[io] = 0; st[in] = (int8_t)wid;
            st[idx(dir(np, bd)) * 3 + 1] += 1;
            st[(DEMETER + NB * player) * 3 + 2] = (int8_t)(64 + (worker * 9 + bd + 1));
            opp_next = false;
            break;
        case HERMES:
            st[io] = 0; st[in] = (int8_t)wid;
            st[(HERMES + NB * player) * 3 + 2] += 1;
            opp_next = false;
            break;
        case PROMETHEUS:
            st[idx(dir(old, bd)) * 3 + 1] += 1;
            st[(PROMETHEUS + NB * player) * 3 + 2] = (int8_t)(64 + (worker * 9 + 1));
            opp_next = false;
            break;
        default: break;
        }
        if (st[2 * NB * 3 + 2] < 127) st[2 * NB * 3 + 2] += 1;                                      // :544-545
        return opp_next ? 1 - player : player;
    }

    __device__ static __forceinline__ int get_round(const int8_t* st) { return GP(st, 2 * NB); }   // :655-656

    __device__ static int get_score(const int8_t* st, int player) {                                 // :84-97
        int hi = 0;
#pragma unroll
        for (int i = 0; i < 25; i++) {
            const int w = W(st, i), lv = LV(st, i);
            hi = ((player == 0 ? w > 0 : w < 0) && lv > hi) ? lv : hi;
        }
        return hi;
    }

    // Board.check_end_game(next_player) :552-565 -- wave-cooperative (needs the valid-move scan)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        out[0] = out[1] = 0.f;
        const int l = lane_id();
        // get_score(p) == 3 <=> one of p's workers stands on level 3 (levels under a worker are 0..3): one read per cell, two ballots
        const int w = l < 25 ? W(st, l) : 0, lv = l < 25 ? LV(st, l) : 0;
        const bool top0 = __ballot(w > 0 && lv == 3) != 0ull, top1 = __ballot(w < 0 && lv == 3) != 0ull;
        if (top0 || GP(st, PAN + NB * 0) > 64) { out[0] = 1.f; out[1] = -1.f; return true; }
        if (top1 || GP(st, PAN + NB * 1) > 64) { out[0] = -1.f; out[1] = 1.f; return true; }
        const int god = owned_god(st, next_player);
        const bool opp_athena = GP(st, ATHENA + NB * ((next_player + 1) % 2)) > 64;
        const Workers wk = find_workers(st);
        // "no valid move for next_player" needs the valid-move scan: it is run once and its mask kept in mask_scratch, so that
        // the caller that asks for the valid mask of the same (state, player) right afterwards (create_leaf: MCTS.py:131 then :142) finds
        // it there instead of evaluating every action a second time (ENDED_FILLS_MASK)
        const bool any = scan_valid(st, next_player, god, opp_athena, wk, mask_scratch);
        if (!any) {
            if (next_player == 0) { out[0] = -1.f; out[1] = 1.f; }
            else { out[0] = 1.f; out[1] = -1.f; }
            return true;
        }
        return false;
    }

    // Board.swap_players :567-576
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        if (k != 1) return;
        const int l = lane_id();
        if (l < 2 * NB) tmp[l] = st[l * 3 + 2];
        wave_sync();
        if (l < 25) st[l * 3] = (int8_t)(-st[l * 3]);
        if (l < 2 * NB) st[l * 3 + 2] = tmp[(l + NB) % (2 * NB)];
        wave_sync();
    }

    // init_game :99-120 with INIT_METHOD == 1 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        int cells[25];
        for (int i = 0; i < 25; i++) cells[i] = i;
        const int wl[4] = {1, -1, 2, -2};
        for (int i = 0; i < 4; i++) {
            int j = i + (int)(rng.u01() * (25 - i));
            j = j > 24 ? 24 : j;
            const int t = cells[i]; cells[i] = cells[j]; cells[j] = t;
            st[cells[i] * 3] = (int8_t)wl[i];
        }
        int g0 = NO_GOD, g1 = NO_GOD;
        if (NB > 1) {
            int gods[10];
            for (int i = 0; i < NB - 1; i++) gods[i] = i;
            for (int i = 0; i < 2; i++) {
                int j = i + (int)(rng.u01() * (NB - 1 - i));
                j = j > NB - 2 ? NB - 2 : j;
                const int t = gods[i]; gods[i] = gods[j]; gods[j] = t;
            }
            g0 = gods[0] + 1; g1 = gods[1] + 1;
        }
        st[(g0 + NB * 0) * 3 + 2] = 64;
        st[(g1 + NB * 1) * 3 + 2] = 64;
    }
};

}  // namespace azg
