/* ORACLE (test infrastructure, not product).  Azul, 2 players.
 * Scalar C restatement of azul/AzulLogicNumba.py `Board` (reference lines cited).  State = int8[23][6], byte-identical
 * to board.tobytes(): scores(1) bag(1) discards(1) centre(1) factories(5) player_colours(2) player_row_numbers(2) walls(10). */
#include <string.h>
#include "azg_oracle.h"

#define COLS 6
#define ROW(st, r) ((st) + (r) * COLS)
enum { R_SCORES = 0, R_BAG = 1, R_DISC = 2, R_CENTRE = 3, R_FACT = 4, R_PCOL = 9, R_PROW = 11, R_WALL = 13 };

void azul_valid_moves(const azo_game* g, const int8_t* st, int player, uint8_t* out) {          /* :97-123 */
    (void)g;
    const int8_t* pc = ROW(st, R_PCOL + player);
    const int8_t* pr = ROW(st, R_PROW + player);
    for (int f = 0; f < 6; f++) {
        const int8_t* src = f == 0 ? ROW(st, R_CENTRE) : ROW(st, R_FACT + f - 1);
        for (int c = 0; c < 5; c++) {
            int avail = f == 0 ? (src[c] != 0) : (src[c] > 0);
            for (int line = 0; line < 6; line++) {
                int line_free = line == 5 ? 1 : (pc[line] == -1);
                int wall_free = line == 5 ? 1 : (ROW(st, R_WALL + 5 * player + line)[(c + line) % 5] == 0);
                int correct = pc[line] == c;
                int not_full = pr[line] < line + 1;
                out[f * 30 + c * 6 + line] = (uint8_t)(avail && ((line_free && wall_free) || (correct && not_full)));
            }
        }
    }
}

static int64_t floor_mod(int64_t x, int64_t m) {
    if (m == 0) return 0;                 /* NumPy integer x % 0 -> 0 (RuntimeWarning) */
    int64_t r = x % m;
    if (r != 0 && ((r < 0) != (m < 0))) r += m;
    return r;
}

/* select_tiles_from_bag :257-268 */
static void select_tiles(int8_t* st, int num, int64_t seed, azo_rng* rng, int8_t* result6) {
    int8_t* bag = ROW(st, R_BAG);
    memset(result6, 0, 6);
    for (int k = 0; k < num; k++) {
        int64_t total = 0;
        for (int c = 0; c < 6; c++) total += bag[c];
        int idx;
        if (seed == 0) {
            double u = azo_rng_u01(rng), acc = 0.0;
            idx = 0;
            if (total > 0) {
                int c;
                for (c = 0; c < 6; c++) { acc += (double)bag[c] / (double)total; if (acc > u) break; }
                if (c >= 6) { for (c = 5; c > 0 && bag[c] <= 0; c--) {} }
                idx = c;
            }
        } else {
            int64_t s = 0;
            for (int c = 0; c < 5; c++) s += (int64_t)bag[c] << c;
            int64_t fake = floor_mod(4594591LL * (seed + s), total);
            int64_t cum = 0;
            idx = 5;
            for (int c = 0; c < 5; c++) { cum += bag[c]; if (cum > fake) { idx = c; break; } }
        }
        result6[idx] += 1;
        bag[idx] -= 1;
    }
}

/* setup_new_round :237-255 */
static int setup_new_round(int8_t* st, int64_t seed, azo_rng* rng) {
    int8_t* bag = ROW(st, R_BAG);
    int8_t* disc = ROW(st, R_DISC);
    for (int i = 0; i < 5; i++) {
        int8_t* fac = ROW(st, R_FACT + i);
        int sum = 0;
        for (int c = 0; c < 6; c++) sum += bag[c];
        int8_t sel[6];
        if (sum < 4) {
            int to_add = 4 - sum;
            memcpy(fac, bag, 6);
            memcpy(bag, disc, 6);
            memset(disc, 0, 6);
            select_tiles(st, to_add, seed, rng, sel);
            for (int c = 0; c < 6; c++) fac[c] += sel[c];
        } else {
            select_tiles(st, 4, seed, rng, sel);
            memcpy(fac, sel, 6);
        }
    }
    int next;
    if (ROW(st, R_PCOL + 1)[5] == 1) { next = 1; ROW(st, R_PCOL + 1)[5] = 0; }
    else { next = 0; ROW(st, R_PCOL + 0)[5] = 0; }
    ROW(st, R_SCORES)[2] += 1;
    ROW(st, R_CENTRE)[5] = 1;
    return next;
}

static int count_consecutive(const int8_t* base, int stride, int len, int index) {              /* :214-225 */
    int count = 1, left = index - 1, right = index + 1;
    while (left >= 0 && base[left * stride] == 1) { count++; left--; }
    while (right < len && base[right * stride] == 1) { count++; right++; }
    return count;
}

static int score_change(int8_t* wall /* 5 rows of COLS */, int r, int c) {                      /* :227-235 */
    wall[r * COLS + c] = 1;
    int row_adj = (c > 0 && wall[r * COLS + c - 1] == 1) || (c < 4 && wall[r * COLS + c + 1] == 1);
    int col_adj = (r > 0 && wall[(r - 1) * COLS + c] == 1) || (r < 4 && wall[(r + 1) * COLS + c] == 1);
    if (!row_adj && !col_adj) return 1;
    int rs = row_adj ? count_consecutive(wall + r * COLS, 1, 5, c) : 0;
    int cs = col_adj ? count_consecutive(wall + c, COLS, 5, r) : 0;
    return rs + cs;
}

static void score_round(int8_t* st) {                                                           /* :169-190 */
    static const int8_t penalty[8] = {0, 1, 2, 4, 6, 8, 11, 14};
    int8_t* scores = ROW(st, R_SCORES);
    int8_t* disc = ROW(st, R_DISC);
    int pl[10], rw[10], col[10], n = 0;
    for (int p = 0; p < 2; p++)
        for (int r = 0; r < 5; r++)
            if (ROW(st, R_PROW + p)[r] == r + 1) { pl[n] = p; rw[n] = r; col[n] = ROW(st, R_PCOL + p)[r]; n++; }
    for (int i = 0; i < n; i++) {
        int c = ((col[i] + rw[i]) % 5 + 5) % 5;
        scores[pl[i]] = (int8_t)(scores[pl[i]] + score_change(ROW(st, R_WALL + 5 * pl[i]), rw[i], c));
        ROW(st, R_WALL + 5 * pl[i] + rw[i])[c] = 1;
    }
    for (int i = 0; i < n; i++) disc[col[i]] = (int8_t)(disc[col[i]] + rw[i]);
    for (int i = 0; i < n; i++) { ROW(st, R_PROW + pl[i])[rw[i]] = 0; ROW(st, R_PCOL + pl[i])[rw[i]] = -1; }
    for (int p = 0; p < 2; p++) {
        int fl = ROW(st, R_PROW + p)[5];
        if (fl > 7) fl = 7;
        int s = scores[p] - penalty[fl < 0 ? 0 : fl];
        scores[p] = (int8_t)(s > 0 ? s : 0);
        ROW(st, R_PROW + p)[5] = 0;
    }
}

static int game_over(const int8_t* st) {                                                        /* :161-167 */
    for (int i = 0; i < 10; i++) {
        const int8_t* w = ROW(st, R_WALL + i);
        if (w[0] == 1 && w[1] == 1 && w[2] == 1 && w[3] == 1 && w[4] == 1) return 1;
    }
    return 0;
}

static void score_bonuses(int8_t* st) {                                                         /* :192-212 */
    int8_t* scores = ROW(st, R_SCORES);
    for (int p = 0; p < 2; p++) {
        const int8_t* w = ROW(st, R_WALL + 5 * p);
        for (int r = 0; r < 5; r++) {
            int all = 1;
            for (int c = 0; c < 5; c++) all &= w[r * COLS + c] == 1;
            if (all) scores[p] = (int8_t)(scores[p] + 2);
        }
        for (int c = 0; c < 5; c++) {
            int all = 1;
            for (int r = 0; r < 5; r++) all &= w[r * COLS + c] == 1;
            if (all) scores[p] = (int8_t)(scores[p] + 7);
        }
        int diags = 0;
        for (int i = 0; i < 5; i++) {
            int all = 1;
            for (int j = 0; j < 5; j++) all &= w[j * COLS + (j + i) % 5] == 1;
            diags += all;
        }
        scores[p] = (int8_t)(scores[p] + diags * 10);
    }
}

int azul_make_move(const azo_game* g, int8_t* st, int move, int player, int64_t seed, azo_rng* rng) {   /* :125-159 */
    (void)g;
    int8_t* fac = move < 30 ? ROW(st, R_CENTRE) : ROW(st, R_FACT + (move - 30) / 30);
    int colour = (move % 30) / 6, line = move % 6;
    int8_t* pr = ROW(st, R_PROW + player);
    int8_t* pc = ROW(st, R_PCOL + player);
    int num = fac[colour], to_floor;
    if (line == 5) to_floor = num;
    else {
        int on_line = pr[line];
        int to_line = line + 1 - on_line < num ? line + 1 - on_line : num;
        to_floor = num - to_line;
        pr[line] = (int8_t)(pr[line] + to_line);
        pc[line] = (int8_t)colour;
    }
    pr[5] = (int8_t)(pr[5] + to_floor);
    ROW(st, R_DISC)[colour] = (int8_t)(ROW(st, R_DISC)[colour] + to_floor);
    fac[colour] = 0;
    if (move < 30) {
        if (fac[5] == 1) { pr[5] = (int8_t)(pr[5] + 1); pc[5] = 1; fac[5] = 0; }
    } else {
        int8_t* centre = ROW(st, R_CENTRE);
        for (int c = 0; c < 6; c++) { centre[c] = (int8_t)(centre[c] + fac[c]); fac[c] = 0; }
    }
    int empty = 1;
    for (int i = 0; i < 5 * COLS; i++) empty &= ROW(st, R_FACT)[i] == 0;
    for (int c = 0; c < 5; c++) empty &= ROW(st, R_CENTRE)[c] == 0;
    if (empty) {
        score_round(st);
        int next = setup_new_round(st, seed, rng);
        if (game_over(st)) score_bonuses(st);
        return next;
    }
    return (player + 1) % 2;
}

void azul_game_ended(const azo_game* g, const int8_t* st, int next_player, float* out) {        /* :283-301 */
    (void)g; (void)next_player;
    out[0] = out[1] = 0.f;
    if (!game_over(st)) return;
    int rows[2] = {0, 0};
    for (int p = 0; p < 2; p++)
        for (int r = 0; r < 5; r++) {
            const int8_t* w = ROW(st, R_WALL + 5 * p + r);
            rows[p] += (w[0] == 1 && w[1] == 1 && w[2] == 1 && w[3] == 1 && w[4] == 1);
        }
    int s0 = ROW(st, R_SCORES)[0], s1 = ROW(st, R_SCORES)[1];
    if (s0 > s1 || (s0 == s1 && rows[0] > rows[1])) { out[0] = 1.f; out[1] = -1.f; }
    else if (s1 > s0 || (s0 == s1 && rows[1] > rows[0])) { out[0] = -1.f; out[1] = 1.f; }
    else { out[0] = 0.01f; out[1] = 0.01f; }
}

void azul_swap_players(const azo_game* g, int8_t* st, int k) {                                  /* :303-308 (any k) */
    (void)g; (void)k;
    int8_t t = ROW(st, R_SCORES)[0]; ROW(st, R_SCORES)[0] = ROW(st, R_SCORES)[1]; ROW(st, R_SCORES)[1] = t;
    int8_t tmp[5 * COLS];
    memcpy(tmp, ROW(st, R_PCOL), COLS); memcpy(ROW(st, R_PCOL), ROW(st, R_PCOL + 1), COLS); memcpy(ROW(st, R_PCOL + 1), tmp, COLS);
    memcpy(tmp, ROW(st, R_PROW), COLS); memcpy(ROW(st, R_PROW), ROW(st, R_PROW + 1), COLS); memcpy(ROW(st, R_PROW + 1), tmp, COLS);
    memcpy(tmp, ROW(st, R_WALL), 5 * COLS); memcpy(ROW(st, R_WALL), ROW(st, R_WALL + 5), 5 * COLS); memcpy(ROW(st, R_WALL + 5), tmp, 5 * COLS);
}

int azul_get_round(const azo_game* g, const int8_t* st) { (void)g; return ROW(st, R_SCORES)[2]; }    /* :333-334 */
int azul_get_score(const azo_game* g, const int8_t* st, int p) { (void)g; return ROW(st, R_SCORES)[p]; }  /* :84-85 */

void azul_init_board(const azo_game* g, int8_t* st, azo_rng* rng) {                             /* init_game :87-93 */
    memset(st, 0, (size_t)g->S);
    for (int c = 0; c < 5; c++) ROW(st, R_BAG)[c] = 20;
    for (int p = 0; p < 2; p++) for (int c = 0; c < 5; c++) ROW(st, R_PCOL + p)[c] = -1;
    setup_new_round(st, 0, rng);
}

/* RNG-free start of SURVEY.md Appendix C.1: setup_new_round(31416) */
void azul_known_start(const azo_game* g, int8_t* st) {
    memset(st, 0, (size_t)g->S);
    for (int c = 0; c < 5; c++) ROW(st, R_BAG)[c] = 20;
    for (int p = 0; p < 2; p++) for (int c = 0; c < 5; c++) ROW(st, R_PCOL + p)[c] = -1;
    setup_new_round(st, 31416, NULL);
}

/* get_symmetries :310-331 -- the 120 factory permutations in itertools.permutations(range(5)) order */
static int next_perm(int* a, int n) {
    int i = n - 2;
    while (i >= 0 && a[i] >= a[i + 1]) i--;
    if (i < 0) return 0;
    int j = n - 1;
    while (a[j] <= a[i]) j--;
    int t = a[i]; a[i] = a[j]; a[j] = t;
    for (int l = i + 1, r = n - 1; l < r; l++, r--) { t = a[l]; a[l] = a[r]; a[r] = t; }
    return 1;
}

int azul_symmetries(const azo_game* g, const int8_t* st, const float* pi, const uint8_t* va, int8_t* os, float* op,
                    uint8_t* ov, int max_sym) {
    const int S = g->S, A = 180;
    int perm[5] = {0, 1, 2, 3, 4}, k = 0;
    do {
        if (k >= max_sym) return k;
        int8_t* o = os + (size_t)k * S;
        memcpy(o, st, (size_t)S);
        memcpy(op + (size_t)k * A, pi, sizeof(float) * A);
        memcpy(ov + (size_t)k * A, va, A);
        for (int i = 0; i < 5; i++) {
            memcpy(ROW(o, R_FACT + i), ROW(st, R_FACT + perm[i]), COLS);
            memcpy(op + (size_t)k * A + 30 * (i + 1), pi + 30 * (perm[i] + 1), sizeof(float) * 30);
            memcpy(ov + (size_t)k * A + 30 * (i + 1), va + 30 * (perm[i] + 1), 30);
        }
        k++;
    } while (next_perm(perm, 5));
    return k;
}
