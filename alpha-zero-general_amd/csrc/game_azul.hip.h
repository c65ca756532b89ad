// game_azul.hip.h -- Azul (2 players) env step for one wavefront, state staged in LDS.
// Semantics follow azul/AzulLogicNumba.py `Board` (lines cited); state bytes = the reference's int8[23][6]:
// scores(1) bag(1) discards(1) centre(1) factories(5) player_colours(2) player_row_numbers(2) walls(10).
// valid_mask() is lane-parallel over the 180 actions (3 passes); make_move() (incl. round scoring and the factory
// refill, a seeded or random draw of up to 20 tiles) is branchy integer work on 138 LDS bytes and runs on lane 0.
#pragma once
#include "azg_common.hip.h"

namespace azg {

struct AzulDev {
    static constexpr bool STOCHASTIC = false;   // the env step is a function of (state, action, random_seed): edges are memoised
    static constexpr bool RANDOM_SYM = false;   // get_symmetries draws no randomness
    static constexpr int P = 2;
    static constexpr int ROWS = 23, COLS = 6;
    // record heap sizing: a visited node has 21-26 valid moves (one 32-entry page = 736 B; a third of the nodes need two pages); the
    // clean-up runs at 70 % of the node capacity, so 1100 B per capacity node leaves the live records 45 % of slack
    static constexpr int REC_BYTES_HINT = 1100;
    static constexpr int S = 138;
    static constexpr int SP = 144;
    static constexpr int A = 180;                 // action_size :55-57
    static constexpr int AW = 3;
    enum { R_SCORES = 0, R_BAG = 1, R_DISC = 2, R_CENTRE = 3, R_FACT = 4, R_PCOL = 9, R_PROW = 11, R_WALL = 13 };

    __device__ static __forceinline__ const int8_t* row(const int8_t* st, int r) { return st + r * COLS; }
    __device__ static __forceinline__ int8_t* row(int8_t* st, int r) { return st + r * COLS; }

    // Board.valid_moves restricted to one action (:97-123)
    __device__ static bool valid_action(const int8_t* st, int a, int player) {
        const int f = a / 30, c = (a % 30) / 6, line = a % 6;
        const int8_t* src = f == 0 ? row(st, R_CENTRE) : row(st, R_FACT + f - 1);
        const bool avail = f == 0 ? (src[c] != 0) : (src[c] > 0);
        if (!avail) return false;
        if (line == 5) return true;
        const int8_t pc = row(st, R_PCOL + player)[line];
        const bool line_free = pc == -1;
        const bool wall_free = row(st, R_WALL + 5 * player + line)[(c + line) % 5] == 0;
        const bool correct = pc == c;
        const bool not_full = row(st, R_PROW + player)[line] < line + 1;
        return (line_free && wall_free) || (correct && not_full);
    }

    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
#pragma unroll
        for (int k = 0; k < AW; k++) {
            const int a = k * 64 + l;
            const uint64_t m = __ballot(a < A && valid_action(st, a < A ? a : 0, player));
            if (l == 0) mask_lds[k] = m;
        }
    }

    __device__ static long long floor_mod(long long x, long long m) {
        if (m == 0) return 0;
        long long r = x % m;
        if (r != 0 && ((r < 0) != (m < 0))) r += m;
        return r;
    }

    // select_tiles_from_bag :257-268
    __device__ static void select_tiles(int8_t* st, int num, long long seed, Rng& rng, int8_t* result6) {
        int8_t* bag = row(st, R_BAG);
        for (int c = 0; c < 6; c++) result6[c] = 0;
        for (int k = 0; k < num; k++) {
            long long total = 0;
            for (int c = 0; c < 6; c++) total += bag[c];
            int idx;
            if (seed == 0) {
                const double u = rng.u01();
                double acc = 0.0;
                idx = 0;
                if (total > 0) {
                    int c;
                    for (c = 0; c < 6; c++) { acc += (double)bag[c] / (double)total; if (acc > u) break; }
                    if (c >= 6) { for (c = 5; c > 0 && bag[c] <= 0; c--) {} }
                    idx = c;
                }
            } else {
                long long s = 0;
                for (int c = 0; c < 5; c++) s += (long long)bag[c] << c;
                const long long fake = floor_mod(4594591LL * (seed + s), total);
                long long cum = 0;
                idx = 5;
                for (int c = 0; c < 5; c++) { cum += bag[c]; if (cum > fake) { idx = c; break; } }
            }
            result6[idx] += 1;
            bag[idx] -= 1;
        }
    }

    // setup_new_round :237-255
    __device__ static int setup_new_round(int8_t* st, long long seed, Rng& rng) {
        int8_t* bag = row(st, R_BAG);
        int8_t* disc = row(st, R_DISC);
        for (int i = 0; i < 5; i++) {
            int8_t* fac = row(st, R_FACT + i);
            int sum = 0;
            for (int c = 0; c < 6; c++) sum += bag[c];
            int8_t sel[6];
            if (sum < 4) {
                for (int c = 0; c < 6; c++) { fac[c] = bag[c]; bag[c] = disc[c]; disc[c] = 0; }
                select_tiles(st, 4 - sum, seed, rng, sel);
                for (int c = 0; c < 6; c++) fac[c] += sel[c];
            } else {
                select_tiles(st, 4, seed, rng, sel);
                for (int c = 0; c < 6; c++) fac[c] = sel[c];
            }
        }
        int next;
        if (row(st, R_PCOL + 1)[5] == 1) { next = 1; row(st, R_PCOL + 1)[5] = 0; }
        else { next = 0; row(st, R_PCOL + 0)[5] = 0; }
        row(st, R_SCORES)[2] += 1;
        row(st, R_CENTRE)[5] = 1;
        return next;
    }

    __device__ static int count_consecutive(const int8_t* base, int stride, int index) {       // :214-225
        int count = 1, left = index - 1, right = index + 1;
        while (left >= 0 && base[left * stride] == 1) { count++; left--; }
        while (right < 5 && base[right * stride] == 1) { count++; right++; }
        return count;
    }

    __device__ static int score_change(int8_t* wall, int r, int c) {                           // :227-235
        wall[r * COLS + c] = 1;
        const bool row_adj = (c > 0 && wall[r * COLS + c - 1] == 1) || (c < 4 && wall[r * COLS + c + 1] == 1);
        const bool col_adj = (r > 0 && wall[(r - 1) * COLS + c] == 1) || (r < 4 && wall[(r + 1) * COLS + c] == 1);
        if (!row_adj && !col_adj) return 1;
        const int rs = row_adj ? count_consecutive(wall + r * COLS, 1, c) : 0;
        const int cs = col_adj ? count_consecutive(wall + c, COLS, r) : 0;
        return rs + cs;
    }

    __device__ static void score_round(int8_t* st) {                                           // :169-190
        int8_t* scores = row(st, R_SCORES);
        int8_t* disc = row(st, R_DISC);
        // pass 1: wall placement + scoring, in (player, row) order; colours are read before any reset
        for (int p = 0; p < 2; p++)
            for (int r = 0; r < 5; r++)
                if (row(st, R_PROW + p)[r] == r + 1) {
                    const int col = row(st, R_PCOL + p)[r];
                    const int c = ((col + r) % 5 + 5) % 5;
                    scores[p] = (int8_t)(scores[p] + score_change(row(st, R_WALL + 5 * p), r, c));
                    row(st, R_WALL + 5 * p + r)[c] = 1;
                }
        // pass 2: discards + reset of the completed lines
        for (int p = 0; p < 2; p++)
            for (int r = 0; r < 5; r++)
                if (row(st, R_PROW + p)[r] == r + 1) {
                    const int col = row(st, R_PCOL + p)[r];
                    disc[col] = (int8_t)(disc[col] + r);
                    row(st, R_PROW + p)[r] = 0;
                    row(st, R_PCOL + p)[r] = -1;
                }
        for (int p = 0; p < 2; p++) {
            int fl = row(st, R_PROW + p)[5];
            fl = fl > 7 ? 7 : (fl < 0 ? 0 : fl);
            const int pen = fl <= 2 ? fl : (fl <= 5 ? 2 * fl - 2 : 3 * fl - 7);     // {0,1,2,4,6,8,11,14}
            const int s = scores[p] - pen;
            scores[p] = (int8_t)(s > 0 ? s : 0);
            row(st, R_PROW + p)[5] = 0;
        }
    }

    __device__ static bool game_over(const int8_t* st) {                                       // :161-167
        bool over = false;
        for (int i = 0; i < 10; i++) {
            const int8_t* w = row(st, R_WALL + i);
            over = over || (w[0] == 1 && w[1] == 1 && w[2] == 1 && w[3] == 1 && w[4] == 1);
        }
        return over;
    }

    __device__ static void score_bonuses(int8_t* st) {                                         // :192-212
        int8_t* scores = row(st, R_SCORES);
        for (int p = 0; p < 2; p++) {
            const int8_t* w = row(st, R_WALL + 5 * p);
            int add = 0;
            for (int r = 0; r < 5; r++) {
                bool all = true;
                for (int c = 0; c < 5; c++) all = all && w[r * COLS + c] == 1;
                add += all ? 2 : 0;
            }
            for (int c = 0; c < 5; c++) {
                bool all = true;
                for (int r = 0; r < 5; r++) all = all && w[r * COLS + c] == 1;
                add += all ? 7 : 0;
            }
            for (int i = 0; i < 5; i++) {
                bool all = true;
                for (int j = 0; j < 5; j++) all = all && w[j * COLS + (j + i) % 5] == 1;
                add += all ? 10 : 0;
            }
            scores[p] = (int8_t)(scores[p] + add);
        }
    }
    // Board.get_symmetries :310-331: the 120 permutations of the 5 factories in itertools.permutations(range(5)) order
    // (lexicographic); form c moves factory perm_c[i] to slot i, and the 30 actions of each factory with it.
    static constexpr int NSYM_CAND = 120;
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int) { return true; }
    __device__ static __forceinline__ int perm_elem(int c, int i) {          // i-th element of the c-th permutation of 0..4
        int used = 0, rem = c, e = 0;
        const int fact[5] = {24, 6, 2, 1, 1};
        for (int k = 0; k <= i; k++) {
            int d = rem / fact[k];
            rem -= d * fact[k];
            e = 0;
            for (int v = 0; v < 5; v++) {                                     // d-th unused value
                if (used & (1 << v)) continue;
                if (d == 0) { e = v; break; }
                d--;
            }
            used |= 1 << e;
        }
        return e;
    }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int c, int i) {
        const int r = i / COLS, col = i - r * COLS;
        const int src = (r >= R_FACT && r < R_FACT + 5) ? R_FACT + perm_elem(c, r - R_FACT) : r;
        return st[src * COLS + col];
    }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int c, int a) {
        if (a < 30) return a;
        const int f = a / 30 - 1;
        return 30 * (perm_elem(c, f) + 1) + (a - 30 * (f + 1));
    }

    // monotone "age" of a state for the clean-up: Azul's round only advances every ~10-20 plies, so the tiles still on the
    // table refine it -- every move takes at least one tile off the factories / centre, a new round puts up to 20 back:
    // age = 21 * round + (20 - tiles on the table), saturating at 255 (saturated nodes are simply kept)
    __device__ static __forceinline__ int gc_age(const int8_t* st) {
        int tiles = 0;
#pragma unroll
        for (int r = R_CENTRE; r < R_FACT + 5; r++)
#pragma unroll
            for (int c = 0; c < 5; c++) tiles += row(st, r)[c];
        tiles = tiles > 20 ? 20 : tiles;
        const int age = 21 * get_round(st) + (20 - tiles);
        return age > 255 ? 255 : age;
    }

    // any move can end the round, whose refill draws tiles with random_seed (setup_new_round :237-255)
    __device__ static __forceinline__ bool move_uses_seed(int) { return true; }

    // Board.make_move :125-159.  The ply's tile moves are a dozen byte updates and the "table empty?" scan reads 35 consecutive bytes
    // (a few wide LDS reads once the compiler merges them): lane 0 runs the step.  (Round 4, measured and dropped: lane 0 moves the
    // tiles, all lanes read one table byte each and a ballot answers "empty?", lane 0 alone runs the rare end of round -- two more LDS
    // hand-overs than the serial form, -2.7 % env-steps/s at 4096 games.)
    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        return lane0_make_move<AzulDev>(st, move, player, seed, rng);
    }

    // the tiles of one move: factory / centre -> pattern line, floor, discard (:125-147)
    __device__ static void take_tiles(int8_t* st, int move, int player) {
        int8_t* fac = move < 30 ? row(st, R_CENTRE) : row(st, R_FACT + (move - 30) / 30);
        const int colour = (move % 30) / 6, line = move % 6;
        int8_t* pr = row(st, R_PROW + player);
        int8_t* pc = row(st, R_PCOL + player);
        const int num = fac[colour];
        int to_floor;
        if (line == 5) to_floor = num;
        else {
            const int on_line = pr[line];
            const int to_line = line + 1 - on_line < num ? line + 1 - on_line : num;
            to_floor = num - to_line;
            pr[line] = (int8_t)(pr[line] + to_line);
            pc[line] = (int8_t)colour;
        }
        pr[5] = (int8_t)(pr[5] + to_floor);
        row(st, R_DISC)[colour] = (int8_t)(row(st, R_DISC)[colour] + to_floor);
        fac[colour] = 0;
        if (move < 30) {
            if (fac[5] == 1) { pr[5] = (int8_t)(pr[5] + 1); pc[5] = 1; fac[5] = 0; }
        } else {
            int8_t* centre = row(st, R_CENTRE);
            for (int c = 0; c < 6; c++) { centre[c] = (int8_t)(centre[c] + fac[c]); fac[c] = 0; }
        }
    }
    // the table is empty: score the round, refill, bonuses at the end of the game (:149-158); returns the next player
    __device__ static __attribute__((noinline)) int end_round(int8_t* st, long long seed, Rng& rng) {
        score_round(st);
        const int next = setup_new_round(st, seed, rng);
        if (game_over(st)) score_bonuses(st);
        return next;
    }

    // Board.make_move :125-159 -- one thread (host-style entry)
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        take_tiles(st, move, player);
        bool empty = true;
        for (int i = 0; i < 5 * COLS; i++) empty = empty && row(st, R_FACT)[i] == 0;
        for (int c = 0; c < 5; c++) empty = empty && row(st, R_CENTRE)[c] == 0;
        if (empty) return end_round(st, seed, rng);
        return (player + 1) % 2;
    }

    __device__ static __forceinline__ int get_round(const int8_t* st) { return st[2]; }        // :333-334
    __device__ static __forceinline__ int get_score(const int8_t* st, int p) { return st[p]; } // :84-85

    // Board.check_end_game :283-301 (uniform)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        out[0] = out[1] = 0.f;
        if (!game_over(st)) return false;
        int rows[2] = {0, 0};
        for (int p = 0; p < 2; p++)
            for (int r = 0; r < 5; r++) {
                const int8_t* w = row(st, R_WALL + 5 * p + r);
                rows[p] += (w[0] == 1 && w[1] == 1 && w[2] == 1 && w[3] == 1 && w[4] == 1) ? 1 : 0;
            }
        const int s0 = st[0], s1 = st[1];
        if (s0 > s1 || (s0 == s1 && rows[0] > rows[1])) { out[0] = 1.f; out[1] = -1.f; }
        else if (s1 > s0 || (s0 == s1 && rows[1] > rows[0])) { out[0] = -1.f; out[1] = 1.f; }
        else { out[0] = 0.01f; out[1] = 0.01f; }
        return true;
    }

    // Board.swap_players :303-308 -- wave-cooperative byte permutation
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        (void)k;
        for (int i = lane_id(); i < S; i += 64) tmp[i] = st[i];
        wave_sync();
        for (int i = lane_id(); i < S; i += 64) {
            const int r = i / COLS, c = i - r * COLS;
            int src = i;
            if (r == 0 && c < 2) src = 1 - c;
            else if (r == R_PCOL || r == R_PROW) src = i + COLS;
            else if (r == R_PCOL + 1 || r == R_PROW + 1) src = i - COLS;
            else if (r >= R_WALL && r < R_WALL + 5) src = i + 5 * COLS;
            else if (r >= R_WALL + 5) src = i - 5 * COLS;
            st[i] = tmp[src];
        }
        wave_sync();
    }

    // init_game :87-93 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        for (int c = 0; c < 5; c++) row(st, R_BAG)[c] = 20;
        for (int p = 0; p < 2; p++)
            for (int c = 0; c < 5; c++) row(st, R_PCOL + p)[c] = -1;
        setup_new_round(st, 0, rng);
    }
};

}  // namespace azg
