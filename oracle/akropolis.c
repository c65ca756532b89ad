/* ORACLE (test infrastructure).  Akropolis env step: a scalar restatement of akropolis/AkropolisLogicNumba.py (Board :270-611, grid
 * helpers :95-131, tables :184-230) for N = N_PLAYERS = 2 (the shipped constant), 3 and 4 (akropolis/AkropolisConstants.py:
 * CITY_SIZE = 13, CONSTR_SITE_SIZE = N + 2, N_STACKS = 11; the tile set grows with N, TILES_DATA[:, 3] <= N).
 *
 * State = int8 [13][13][3 N + 2] (:7-32), byte index (r * 13 + q) * (3 N + 2) + z, odd-r offset hex grid:
 *   z = p          tile description of player p's city (0 empty, 1 quarry, 2..6 district B Y R P G, 7..11 plaza B Y R P G)
 *   z = N + p      height, z = 2 N + p  tile id (61 = the start tile)
 *   z = 3 N        per-player scalars at (row, col): (p, c) plazas, (N + p, c) districts, (2 N + p, 0) total score code,
 *                  (2 N + p, 1) stones
 *   z = 3 N + 1    globals: (i, j) construction site tile i = three descriptions + tile id, (N + 2, 0..7) bitfield of the tiles still
 *                  in the stacks (MSB first), (N + 3, 0) round, (N + 3, 1) stacks left
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
/* the bound game's player count (N_PLAYERS): set by ak_bind() at every entry point; cell stride ST = 3 NP + 2 bytes */
static __thread int NP = 2, ST = 8, NSITE = 4;
static void ak_bind(const azo_game* g) { NP = g->P; ST = 3 * NP + 2; NSITE = NP + 2; }
#define AT(s, r, q, z) ((s)[((r) * CS + (q)) * ST + (z)])
#define FLAT(s, idx, z) ((s)[(idx) * ST + (z)])
#define DESCR(s, idx, p) FLAT(s, idx, p)
#define HEIGHT(s, idx, p) FLAT(s, idx, NP + (p))
#define TILEID(s, idx, p) FLAT(s, idx, 2 * NP + (p))
#define PLAZAS(s, p, c) AT(s, p, c, 3 * NP)
#define DISTRICTS(s, p, c) AT(s, NP + (p), c, 3 * NP)
#define TOTAL(s, p) AT(s, 2 * NP + (p), 0, 3 * NP)
#define STONES(s, p) AT(s, 2 * NP + (p), 1, 3 * NP)
#define SITE(s, i, j) AT(s, i, j, 3 * NP + 1)
#define BITPACK(s, j) AT(s, NSITE, j, 3 * NP + 1)
#define ROUND_(s) AT(s, NSITE + 1, 0, 3 * NP + 1)
#define STACKS(s) AT(s, NSITE + 1, 1, 3 * NP + 1)
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
        const int d = FLAT(s, i, p), h = HEIGHT(s, i, p);
        if (d == DISTRICT_GREEN) district[GREEN] += h;
        else if (d == DISTRICT_YELLOW) {
            int isolated = 1;
            for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); if (nb >= 0 && FLAT(s, nb, p) == DISTRICT_YELLOW) isolated = 0; }
            if (isolated) district[YELLOW] += h;
        } else if (d == DISTRICT_PURPLE) {
            int ok = 1;
            for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); if (nb < 0 || HEIGHT(s, nb, p) == 0) ok = 0; }
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
            for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); if (nb < 0 || outer[nb]) { district[RED] += HEIGHT(s, i, p); break; } }
    int best = 0;
    for (int st0 = 0; st0 < AREA; st0++) {                                                 /* heaviest chain of houses */
        if (FLAT(s, st0, p) != DISTRICT_BLUE || seen[st0]) continue;
        int chain = 0, top = 0;
        stack[top++] = st0; seen[st0] = 1;
        while (top) {
            const int cur = stack[--top];
            chain += HEIGHT(s, cur, p);
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
    const int ha = HEIGHT(s, c[0], player);
    if (ha != HEIGHT(s, c[1], player) || ha != HEIGHT(s, c[2], player)) return 0;
    if (ha == 0) {
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 6; k++) {
                const int nb = neighbor(c[j], k);
                if (nb >= 0 && HEIGHT(s, nb, player) > 0) return 1;                      /* (the triple itself has height 0) */
            }
        return 0;
    }
    const int ta = TILEID(s, c[0], player);
    return !(ta == TILEID(s, c[1], player) && ta == TILEID(s, c[2], player));
}

void akropolis_valid_moves(const azo_game* g, const int8_t* s, int player, uint8_t* out) {  /* :354-413 */
    ak_bind(g);
    memset(out, 0, (size_t)g->A);
    int slots = STONES(s, player) + 1;
    if (slots > NSITE) slots = NSITE;
    uint8_t pv[NPAT];
    for (int p = 0; p < NPAT; p++) pv[p] = (uint8_t)pattern_valid(s, p, player);
    for (int i = 0; i < slots; i++)
        if (SITE(s, i, 0) != EMPTY) memcpy(out + i * NPAT, pv, NPAT);
}

int akropolis_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t seed, azo_rng* rng) {   /* :314-352 */
    ak_bind(g);
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
        HEIGHT(s, c[j], player) = (int8_t)(HEIGHT(s, c[j], player) + 1);
        TILEID(s, c[j], player) = tile[3];
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
    return (player + 1) % NP;
}

void akropolis_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) { /* :426-437 */
    (void)next_player;
    ak_bind(g);
    for (int p = 0; p < NP; p++) out[p] = 0.f;
    if (!(STACKS(s) <= 0 && SITE(s, 1, 0) == EMPTY)) return;
    long proxy[4], m = -1;
    int nmax = 0;
    for (int p = 0; p < NP; p++) { proxy[p] = (long)score_of(s, p) * 1000 + STONES(s, p); if (proxy[p] > m) m = proxy[p]; }
    for (int p = 0; p < NP; p++) nmax += proxy[p] == m;
    for (int p = 0; p < NP; p++) out[p] = proxy[p] == m ? (nmax == 1 ? 1.f : 0.001f) : -1.f;
}

void akropolis_swap_players(const azo_game* g, int8_t* s, int k) {                          /* :439-470 */
    ak_bind(g);
    k = ((k % NP) + NP) % NP;
    if (k == 0) return;
    int8_t t[12];
    for (int i = 0; i < AREA; i++) {                                                       /* new[p] = old[(p + k) % NP] in every plane */
        for (int z = 0; z < 3 * NP; z++) t[z] = FLAT(s, i, z);
        for (int b = 0; b < 3; b++)
            for (int p = 0; p < NP; p++) FLAT(s, i, b * NP + p) = t[b * NP + (p + k) % NP];
    }
    for (int c = 0; c < 5; c++) {
        for (int p = 0; p < NP; p++) { t[p] = PLAZAS(s, p, c); t[4 + p] = DISTRICTS(s, p, c); }
        for (int p = 0; p < NP; p++) { PLAZAS(s, p, c) = t[(p + k) % NP]; DISTRICTS(s, p, c) = t[4 + (p + k) % NP]; }
    }
    for (int p = 0; p < NP; p++) { t[p] = TOTAL(s, p); t[4 + p] = STONES(s, p); }
    for (int p = 0; p < NP; p++) { TOTAL(s, p) = t[(p + k) % NP]; STONES(s, p) = t[4 + (p + k) % NP]; }
}

int akropolis_get_round(const azo_game* g, const int8_t* s) { ak_bind(g); return ROUND_(s); }
int akropolis_get_score(const azo_game* g, const int8_t* s, int p) { ak_bind(g); return score_of(s, p); }

void akropolis_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                     /* :275-295 */
    ak_bind(g);
    memset(s, 0, (size_t)g->S);
    for (int p = 0; p < NP; p++) STONES(s, p) = (int8_t)(p + 1);
    for (int t = 0; t < 61; t++)
        if ((int)(AKRO_TILES[t] >> 12) <= NP) BITPACK(s, t >> 3) = (int8_t)((uint8_t)BITPACK(s, t >> 3) | (128u >> (t & 7)));
    STACKS(s) = 11;
    for (int p = 0; p < NP; p++) TOTAL(s, p) = (int8_t)(STONES(s, p) / 2 - 128);
    const int centre = (CS / 2) * CS + CS / 2;
    for (int p = 0; p < NP; p++) {
        DESCR(s, centre, p) = PLAZA_BLUE; HEIGHT(s, centre, p) = 1; TILEID(s, centre, p) = 61;
        PLAZAS(s, p, BLUE) = 1;
        for (int d = 0; d < 6; d += 2) {                                                  /* NEIGHBORS[centre, ::2] */
            const int nb = neighbor(centre, d);
            DESCR(s, nb, p) = QUARRY; HEIGHT(s, nb, p) = 1; TILEID(s, nb, p) = 61;
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
    ak_bind(g);
    const int S = g->S, A = g->A;
    int k = 0;
    for (int rot = 0; rot < 6 && k < max_sym; rot++, k++) {
        int8_t* st = os + (size_t)k * S;
        float* p = op + (size_t)k * A;
        uint8_t* v = ov + (size_t)k * A;
        memset(st, 0, (size_t)S); memset(p, 0, sizeof(float) * (size_t)A); memset(v, 0, (size_t)A);
        for (int i = 0; i < AREA; i++) {
            const int nb = rotate_cell(i, rot);
            if (nb >= 0) memcpy(st + ST * nb, s + ST * i, (size_t)ST);
        }
        for (int i = 0; i < AREA; i++) { FLAT(st, i, 3 * NP) = FLAT(s, i, 3 * NP); FLAT(st, i, 3 * NP + 1) = FLAT(s, i, 3 * NP + 1); }
        for (int a = 0; a < A; a++)
            if (valids[a]) {
                int ni = (a / NPAT) * NPAT + rotate_pattern(a % NPAT, rot);
                if (ni < 0) ni += A;                                                       /* new_p[-1]: Python's wrap-around */
                p[ni] = pi[a]; v[ni] = valids[a];
            }
    }
    return k;
}
