/* ORACLE (test infrastructure).  Akropolis env step: a scalar restatement of akropolis/AkropolisLogicNumba.py (Board :270-611, grid
 * helpers :95-131, tables :184-230) with the shipped constants (akropolis/AkropolisConstants.py: N_PLAYERS = 2, CITY_SIZE = 13,
 * CONSTR_SITE_SIZE = 4, N_STACKS = 11).
 *
 * State = int8 [13][13][8] (:7-32), byte index (r * 13 + q) * 8 + z, odd-r offset hex grid:
 *   z = p      tile description of player p's city (0 empty, 1 quarry, 2..6 district B Y R P G, 7..11 plaza B Y R P G)
 *   z = 2 + p  height, z = 4 + p  tile id (61 = the start tile)
 *   z = 6      per-player scalars at (row, col): (p, c) plazas, (2 + p, c) districts, (4 + p, 0) total score code, (4 + p, 1) stones
 *   z = 7      globals: (i, j) construction site tile i = three descriptions + tile id, (4, 0..7) bitfield of the tiles still in the
 *              stacks (MSB first), (5, 0) round, (5, 1) stacks left
 * Action = slot * 1014 + cell * 6 + orientation (:53-61); pattern (cell, o) covers cell + DIR[o], cell, cell + DIR[o + 1] (:198-216).
 *
 * Randomness (_draw_tiles_constr_site :503-518): with random_seed != 0 (MCTS simulations) the refill is the function
 * (2014 (random_seed + round) + 42) mod 61 of the state; with random_seed == 0 (real moves, init) it is np.random.choice(available),
 * defined by the RNG contract as available[floor(u * len)] (tools/refshim/harness.py CounterRandom.choice feeds the reference the
 * same).  get_symmetries (:472-501) rotates about cell (0, 0) -- the corner of the grid, so cells and patterns fall off the board; the
 * scatter with its wrap-around index -1 and last-writer-wins order is reproduced as written. */
#include <string.h>
#include "azg_oracle.h"
#include "akropolis_tables.h"

enum { EMPTY = 0, QUARRY = 1, DISTRICT_BLUE = 2, DISTRICT_YELLOW = 3, DISTRICT_RED = 4, DISTRICT_PURPLE = 5, DISTRICT_GREEN = 6,
       PLAZA_BLUE = 7 };
enum { BLUE, YELLOW, RED, PURPLE, GREEN };
#define CS 13
#define AREA 169
#define NPAT 1014
#define NSITE 4
#define AT(s, r, q, z) ((s)[(((r) * CS + (q)) << 3) + (z)])
#define FLAT(s, idx, z) ((s)[((idx) << 3) + (z)])
#define PLAZAS(s, p, c) AT(s, p, c, 6)
#define DISTRICTS(s, p, c) AT(s, 2 + (p), c, 6)
#define TOTAL(s, p) AT(s, 4 + (p), 0, 6)
#define STONES(s, p) AT(s, 4 + (p), 1, 6)
#define SITE(s, i, j) AT(s, i, j, 7)
#define BITPACK(s, j) AT(s, NSITE, j, 7)
#define ROUND_(s) AT(s, NSITE + 1, 0, 7)
#define STACKS(s) AT(s, NSITE + 1, 1, 7)
static const int PLAZA_STARS[5] = {1, 2, 2, 2, 3};
/* (dq, dr): SW SE E NE NW W (AkropolisConstants.py:77-80) */
static const int DIR_EVEN[6][2] = {{-1, 1}, {0, 1}, {1, 0}, {0, -1}, {-1, -1}, {-1, 0}};
static const int DIR_ODD[6][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, 0}};

static int neighbor(int idx, int d) {                   /* the cell in direction d, or -1 off the board */
    const int r = idx / CS, q = idx % CS;
    const int nq = q + ((r & 1) ? DIR_ODD[d][0] : DIR_EVEN[d][0]), nr = r + ((r & 1) ? DIR_ODD[d][1] : DIR_EVEN[d][1]);
    return (nq >= 0 && nq < CS && nr >= 0 && nr < CS) ? nr * CS + nq : -1;
}
/* PATTERNS[p] :198-216: all three cells on the board, else (-1, -1, -1) */
static int pattern_cells(int p, int* c) {
    const int s = p / 6, o = p % 6;
    c[0] = neighbor(s, o); c[1] = s; c[2] = neighbor(s, (o + 1) % 6);
    if (c[0] < 0 || c[2] < 0) { c[0] = c[1] = c[2] = -1; return 0; }
    return 1;
}
static int type_of(int d) { return d == 0 ? 0 : (d == 1 ? 1 : (d <= 6 ? 2 : 3)); }          /* DESCR_TO_TYPE_COLOR */
static int color_of(int d) { return d <= 1 ? 0 : (d <= 6 ? d - 2 : d - 7); }

static int score_of(const int8_t* s, int p) {                                              /* get_score :421-424 */
    int t = 0;
    for (int c = 0; c < 5; c++) t += (int)DISTRICTS(s, p, c) * PLAZAS(s, p, c) * PLAZA_STARS[c];
    return t + STONES(s, p);
}

static void draw_tiles(int8_t* s, int64_t seed, int initial, azo_rng* rng) {                /* :503-518 */
    for (int i = initial ? 0 : 1; i < NSITE; i++) {
        int avail[64], n = 0;
        for (int t = 0; t < 64; t++)
            if (((uint8_t)BITPACK(s, t >> 3) >> (7 - (t & 7))) & 1) avail[n++] = t;
        int tile;
        if (initial || seed == 0) {
            int k = (int)(azo_rng_u01(rng) * (double)n);
            tile = avail[k >= n ? n - 1 : k];
        } else {
            int64_t v = (2014 * (seed + (int64_t)ROUND_(s)) + 42) % 61;
            if (v < 0) v += 61;                                                            /* Python's % for random_seed = -1 */
            tile = avail[v % n];
        }
        for (int j = 0; j < 3; j++) SITE(s, i, j) = (int8_t)((AKRO_TILES[tile] >> (4 * j)) & 15);
        SITE(s, i, 3) = (int8_t)tile;
        BITPACK(s, tile >> 3) = (int8_t)((uint8_t)BITPACK(s, tile >> 3) & ~(128u >> (tile & 7)));
    }
}

static void update_districts(int8_t* s, int p) {                                           /* :520-611 */
    int district[5] = {0, 0, 0, 0, 0};
    uint8_t outer[AREA], seen[AREA];
    int stack[AREA], n = 0;
    memset(outer, 0, sizeof(outer)); memset(seen, 0, sizeof(seen));
    for (int i = 0; i < AREA; i++) {
        const int d = FLAT(s, i, p), h = FLAT(s, i, 2 + p);
        if (d == DISTRICT_GREEN) district[GREEN] += h;
        else if (d == DISTRICT_YELLOW) {
            int isolated = 1;
            for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); if (nb >= 0 && FLAT(s, nb, p) == DISTRICT_YELLOW) isolated = 0; }
            if (isolated) district[YELLOW] += h;
        } else if (d == DISTRICT_PURPLE) {
            int ok = 1;
            for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); if (nb < 0 || FLAT(s, nb, 2 + p) == 0) ok = 0; }
            if (ok) district[PURPLE] += h;
        } else if (d == EMPTY) {
            for (int k = 0; k < 6; k++) if (neighbor(i, k) < 0) { outer[i] = 1; break; }
            if (outer[i]) stack[n++] = i;
        }
    }
    for (int k0 = 0; k0 < n; k0++)                                                         /* flood fill from the border */
        for (int k = 0; k < 6; k++) {
            const int nb = neighbor(stack[k0], k);
            if (nb < 0 || outer[nb] || FLAT(s, nb, p) != EMPTY) continue;
            outer[nb] = 1; stack[n++] = nb;
        }
    for (int i = 0; i < AREA; i++)
        if (FLAT(s, i, p) == DISTRICT_RED)
            for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); if (nb < 0 || outer[nb]) { district[RED] += FLAT(s, i, 2 + p); break; } }
    int best = 0;
    for (int st0 = 0; st0 < AREA; st0++) {                                                 /* heaviest chain of houses */
        if (FLAT(s, st0, p) != DISTRICT_BLUE || seen[st0]) continue;
        int chain = 0, top = 0;
        stack[top++] = st0; seen[st0] = 1;
        while (top) {
            const int cur = stack[--top];
            chain += FLAT(s, cur, 2 + p);
            for (int k = 0; k < 6; k++) {
                const int nb = neighbor(cur, k);
                if (nb < 0 || seen[nb] || FLAT(s, nb, p) != DISTRICT_BLUE) continue;
                seen[nb] = 1; stack[top++] = nb;
            }
        }
        if (chain > best) best = chain;
    }
    district[BLUE] = best;
    for (int c = 0; c < 5; c++) DISTRICTS(s, p, c) = (int8_t)district[c];
}

static int pattern_valid(const int8_t* s, int pat, int player) {                           /* valid_moves :358-398 */
    int c[3];
    if (!pattern_cells(pat, c)) return 0;
    const int ha = FLAT(s, c[0], 2 + player);
    if (ha != FLAT(s, c[1], 2 + player) || ha != FLAT(s, c[2], 2 + player)) return 0;
    if (ha == 0) {
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 6; k++) {
                const int nb = neighbor(c[j], k);
                if (nb >= 0 && FLAT(s, nb, 2 + player) > 0) return 1;                      /* (the triple itself has height 0) */
            }
        return 0;
    }
    const int ta = FLAT(s, c[0], 4 + player);
    return !(ta == FLAT(s, c[1], 4 + player) && ta == FLAT(s, c[2], 4 + player));
}

void akropolis_valid_moves(const azo_game* g, const int8_t* s, int player, uint8_t* out) {  /* :354-413 */
    memset(out, 0, (size_t)g->A);
    int slots = STONES(s, player) + 1;
    if (slots > NSITE) slots = NSITE;
    uint8_t pv[NPAT];
    for (int p = 0; p < NPAT; p++) pv[p] = (uint8_t)pattern_valid(s, p, player);
    for (int i = 0; i < slots; i++)
        if (SITE(s, i, 0) != EMPTY) memcpy(out + i * NPAT, pv, NPAT);
}

int akropolis_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t seed, azo_rng* rng) {   /* :314-352 */
    (void)g;
    const int slot = move / NPAT, pat = move % NPAT;
    int8_t tile[4];
    int c[3];
    for (int j = 0; j < 4; j++) tile[j] = SITE(s, slot, j);
    for (int i = slot; i < NSITE - 1; i++)
        for (int j = 0; j < 4; j++) SITE(s, i, j) = SITE(s, i + 1, j);
    for (int j = 0; j < 4; j++) SITE(s, NSITE - 1, j) = EMPTY;
    pattern_cells(pat, c);
    for (int j = 0; j < 3; j++) {
        const int under = FLAT(s, c[j], player);
        if (type_of(under) == 3) PLAZAS(s, player, color_of(under)) = (int8_t)(PLAZAS(s, player, color_of(under)) - 1);
        if (type_of(under) == 1) STONES(s, player) = (int8_t)(STONES(s, player) + 1);
        FLAT(s, c[j], player) = tile[j];
        FLAT(s, c[j], 2 + player) = (int8_t)(FLAT(s, c[j], 2 + player) + 1);
        FLAT(s, c[j], 4 + player) = tile[3];
        if (type_of(tile[j]) == 3) PLAZAS(s, player, color_of(tile[j])) = (int8_t)(PLAZAS(s, player, color_of(tile[j])) + 1);
    }
    STONES(s, player) = (int8_t)(STONES(s, player) - slot);
    update_districts(s, player);
    TOTAL(s, player) = (int8_t)(score_of(s, player) / 2 - 128);                            /* encode_score_to_int8 :239-248 */
    ROUND_(s) = (int8_t)(ROUND_(s) + 1);
    if (SITE(s, 1, 0) == EMPTY && STACKS(s) > 0) {
        draw_tiles(s, seed, 0, rng);
        STACKS(s) = (int8_t)(STACKS(s) - 1);
    }
    return (player + 1) % 2;
}

void akropolis_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) { /* :426-437 */
    (void)g; (void)next_player;
    out[0] = out[1] = 0.f;
    if (!(STACKS(s) <= 0 && SITE(s, 1, 0) == EMPTY)) return;
    long proxy[2];
    for (int p = 0; p < 2; p++) proxy[p] = (long)score_of(s, p) * 1000 + STONES(s, p);
    const long m = proxy[0] > proxy[1] ? proxy[0] : proxy[1];
    const int single = (proxy[0] == m) + (proxy[1] == m) == 1;
    for (int p = 0; p < 2; p++) out[p] = proxy[p] == m ? (single ? 1.f : 0.001f) : -1.f;
}

void akropolis_swap_players(const azo_game* g, int8_t* s, int k) {                          /* :439-470 */
    (void)g;
    if (k % 2 == 0) return;
    for (int i = 0; i < AREA; i++)
        for (int z = 0; z < 6; z += 2) { const int8_t t = FLAT(s, i, z); FLAT(s, i, z) = FLAT(s, i, z + 1); FLAT(s, i, z + 1) = t; }
    for (int c = 0; c < 5; c++) {
        int8_t t = PLAZAS(s, 0, c); PLAZAS(s, 0, c) = PLAZAS(s, 1, c); PLAZAS(s, 1, c) = t;
        t = DISTRICTS(s, 0, c); DISTRICTS(s, 0, c) = DISTRICTS(s, 1, c); DISTRICTS(s, 1, c) = t;
    }
    int8_t t = TOTAL(s, 0); TOTAL(s, 0) = TOTAL(s, 1); TOTAL(s, 1) = t;
    t = STONES(s, 0); STONES(s, 0) = STONES(s, 1); STONES(s, 1) = t;
}

int akropolis_get_round(const azo_game* g, const int8_t* s) { (void)g; return ROUND_(s); }
int akropolis_get_score(const azo_game* g, const int8_t* s, int p) { (void)g; return score_of(s, p); }

void akropolis_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                     /* :275-295 */
    memset(s, 0, (size_t)g->S);
    STONES(s, 0) = 1; STONES(s, 1) = 2;
    for (int t = 0; t < 61; t++)
        if ((AKRO_TILES[t] >> 12) <= 2) BITPACK(s, t >> 3) = (int8_t)((uint8_t)BITPACK(s, t >> 3) | (128u >> (t & 7)));
    STACKS(s) = 11;
    for (int p = 0; p < 2; p++) TOTAL(s, p) = (int8_t)(STONES(s, p) / 2 - 128);
    const int centre = (CS / 2) * CS + CS / 2;
    for (int p = 0; p < 2; p++) {
        FLAT(s, centre, p) = PLAZA_BLUE; FLAT(s, centre, 2 + p) = 1; FLAT(s, centre, 4 + p) = 61;
        PLAZAS(s, p, BLUE) = 1;
        for (int d = 0; d < 6; d += 2) {                                                  /* NEIGHBORS[centre, ::2] */
            const int nb = neighbor(centre, d);
            FLAT(s, nb, p) = QUARRY; FLAT(s, nb, 2 + p) = 1; FLAT(s, nb, 4 + p) = 61;
        }
    }
    draw_tiles(s, 0, 1, rng);
}

/* ---- get_symmetries :472-501 ---- */
static int rotate_cell(int idx, int k) {                                                    /* :95-114: k x 60 degrees about cell (0, 0) */
    if (idx < 0) return -1;
    const int r = idx / CS, q = idx - r * CS;
    int x = q - ((r - (r & 1)) / 2), z = r, y = -x - z;
    for (int i = 0; i < k; i++) { const int nx = -z, ny = -x, nz = -y; x = nx; y = ny; z = nz; }
    const int r2 = z, q2 = x + ((r2 - (r2 & 1)) / 2);
    return (r2 >= 0 && r2 < CS && q2 >= 0 && q2 < CS) ? r2 * CS + q2 : -1;
}
static int rotate_pattern(int pat, int k) {                                                 /* :116-129: first pattern with the rotated cells */
    int c[3], t[3], rc[3];
    pattern_cells(pat, c);
    for (int j = 0; j < 3; j++) rc[j] = rotate_cell(c[j], k);
    if (rc[0] < 0 && rc[1] < 0 && rc[2] < 0) return 0;                                     /* pattern 0 is (-1, -1, -1) */
    if (rc[1] < 0) return -1;
    for (int o = 0; o < 6; o++) {
        if (!pattern_cells(rc[1] * 6 + o, t)) continue;
        if (t[0] == rc[0] && t[2] == rc[2]) return rc[1] * 6 + o;
    }
    return -1;
}

int akropolis_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                         uint8_t* ov, int max_sym) {
    const int S = g->S, A = g->A;
    int k = 0;
    for (int rot = 0; rot < 6 && k < max_sym; rot++, k++) {
        int8_t* st = os + (size_t)k * S;
        float* p = op + (size_t)k * A;
        uint8_t* v = ov + (size_t)k * A;
        memset(st, 0, (size_t)S); memset(p, 0, sizeof(float) * (size_t)A); memset(v, 0, (size_t)A);
        for (int i = 0; i < AREA; i++) {
            const int nb = rotate_cell(i, rot);
            if (nb >= 0) memcpy(st + 8 * nb, s + 8 * i, 8);
        }
        for (int i = 0; i < AREA; i++) { st[8 * i + 6] = s[8 * i + 6]; st[8 * i + 7] = s[8 * i + 7]; }
        for (int a = 0; a < A; a++)
            if (valids[a]) {
                int ni = (a / NPAT) * NPAT + rotate_pattern(a % NPAT, rot);
                if (ni < 0) ni += A;                                                       /* new_p[-1]: Python's wrap-around */
                p[ni] = pi[a]; v[ni] = valids[a];
            }
    }
    return k;
}
