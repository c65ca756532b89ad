// game_akropolis.hip.h -- Akropolis env step on the device plugin interface: akropolis/AkropolisLogicNumba.py
// (Board :270-611, grid helpers :95-131, tables :184-230) for N = N_PLAYERS = 2 (the shipped constant), 3 and 4
// (AkropolisConstants.py: CITY_SIZE = 13, CONSTR_SITE_SIZE = N + 2, N_STACKS = 11; the tile set grows with N).
//
// State int8 [13][13][3 N + 2] (:7-32), byte (r * 13 + q) * (3 N + 2) + z on an odd-r offset hex grid:
//   z = p          tile description of player p's city (0 empty, 1 quarry, 2..6 district B Y R P G, 7..11 plaza B Y R P G)
//   z = N + p      height, z = 2 N + p  tile id (61 = the start tile)
//   z = 3 N        per-player scalars at (row, col): (p, c) plazas, (N + p, c) districts, (2 N + p, 0) total score code,
//                  (2 N + p, 1) stones
//   z = 3 N + 1    globals: (i, j) construction-site tile i = three descriptions + tile id, (N + 2, 0..7) bitfield of the tiles still
//                  in the stacks (MSB first), (N + 3, 0) round, (N + 3, 1) stacks left
// Action = slot * 1014 + cell * 6 + orientation (:53-61); pattern (cell, o) covers cell + DIR[o], cell, cell + DIR[o + 1].
//
// The env step is a function of (state, action, random_seed) for random_seed != 0 -- the refill is (2014 (seed + round) + 42) mod 61
// (:510-512) -- so edges are memoised like Splendor's; real moves and init (random_seed == 0) draw np.random.choice(available) =
// available[floor(u len)] from the tree's counter stream.  get_symmetries (:472-501) rotates about cell (0, 0), the corner of the
// grid: cells and patterns fall off the board, the scatter wraps index -1 to the last action and later writers win -- lane 0 builds
// the forms exactly as written (BUILT symmetric forms, k_env_symmetries_built).
//
// Valid moves: 1014 pattern predicates in 16 ballots, expanded to the N + 2 slots in 64..96 words; init and the tile placement of make_move
// on lane 0, the district scoring (per-cell rules + flood fill over 169 cells) on all lanes; swap_players is a byte map applied by all lanes.
#pragma once
#include "azg_common.hip.h"
#include "akropolis_tables.h"

namespace azg {

template <int NPL>
struct AkropolisDev {
    static constexpr int P = NPL;
    static constexpr int ST = 3 * NPL + 2;         // bytes per cell
    static constexpr int ROWS = 169, COLS = ST;
    static constexpr int S = 169 * ST;
    static constexpr int SP = RoundUp16<S>::value;
    static constexpr int A = (NPL + 2) * 1014;
    static constexpr int AW = (A + 63) / 64;
    static constexpr bool STOCHASTIC = false;
    static constexpr bool RANDOM_SYM = true;      // symmetric forms are built by lane 0 (k_env_symmetries_built); no draw is consumed
    static constexpr bool SYM_DEDUP = false;
    static constexpr int REC_NV_HINT = 1200;      // a few hundred placements x the affordable tiles per node (record heap sizing)
    enum { EMPTY = 0, QUARRY = 1, DISTRICT_BLUE = 2, DISTRICT_YELLOW = 3, DISTRICT_RED = 4, DISTRICT_PURPLE = 5, DISTRICT_GREEN = 6,
           PLAZA_BLUE = 7 };
    enum { BLUE, YELLOW, RED, PURPLE, GREEN };
    enum { CS = 13, AREA = 169, NPAT = 1014, NSITE = NPL + 2, ZS = 3 * NPL, ZG = 3 * NPL + 1 };
    enum { O_ROUND = ((NSITE + 1) * CS) * ST + ZG, O_STACKS = ((NSITE + 1) * CS + 1) * ST + ZG };

    __device__ static __forceinline__ int at(int r, int q, int z) { return (r * CS + q) * ST + z; }
    __device__ static __forceinline__ int o_descr(int idx, int p) { return idx * ST + p; }
    __device__ static __forceinline__ int o_height(int idx, int p) { return idx * ST + NPL + p; }
    __device__ static __forceinline__ int o_tileid(int idx, int p) { return idx * ST + 2 * NPL + p; }
    __device__ static __forceinline__ int o_plazas(int p, int c) { return at(p, c, ZS); }
    __device__ static __forceinline__ int o_districts(int p, int c) { return at(NPL + p, c, ZS); }
    __device__ static __forceinline__ int o_total(int p) { return at(2 * NPL + p, 0, ZS); }
    __device__ static __forceinline__ int o_stones(int p) { return at(2 * NPL + p, 1, ZS); }
    __device__ static __forceinline__ int o_site(int i, int j) { return at(i, j, ZG); }
    __device__ static __forceinline__ int o_bitpack(int j) { return at(NSITE, j, ZG); }
    __device__ static __forceinline__ int stars(int c) { return c == 0 ? 1 : (c == 4 ? 3 : 2); }                 // PLAZA_STARS
    __device__ static __forceinline__ int type_of(int d) { return d == 0 ? 0 : (d == 1 ? 1 : (d <= 6 ? 2 : 3)); }   // DESCR_TO_TYPE_COLOR
    __device__ static __forceinline__ int color_of(int d) { return d <= 1 ? 0 : (d <= 6 ? d - 2 : d - 7); }

    // the cell in direction d (SW SE E NE NW W, AkropolisConstants.py:77-80), or -1 off the board
    __device__ static __forceinline__ int neighbor(int idx, int d) {
        const int r = idx / CS, q = idx - r * CS;
        // (dq, dr) even rows: (-1,1) (0,1) (1,0) (0,-1) (-1,-1) (-1,0); odd rows: (0,1) (1,1) (1,0) (1,-1) (0,-1) (-1,0)
        const int dr = d < 2 ? 1 : ((d == 2 || d == 5) ? 0 : -1);
        const int dq_even = (d == 0 || d >= 4) ? -1 : (d == 2 ? 1 : 0);
        const int dq = (d == 2 || d == 5) ? dq_even : dq_even + (r & 1);
        const int nq = q + dq, nr = r + dr;
        return (nq >= 0 && nq < CS && nr >= 0 && nr < CS) ? nr * CS + nq : -1;
    }
    // PATTERNS[p] :198-216
    __device__ static __forceinline__ bool pattern_cells(int p, int* c) {
        const int s = p / 6, o = p - 6 * s;
        c[0] = neighbor(s, o); c[1] = s; c[2] = neighbor(s, o == 5 ? 0 : o + 1);
        if (c[0] < 0 || c[2] < 0) { c[0] = c[1] = c[2] = -1; return false; }
        return true;
    }

    __device__ static int get_score(const int8_t* st, int p) {                 // :421-424
        int t = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) t += (int)st[o_districts(p, c)] * st[o_plazas(p, c)] * stars(c);
        return t + st[o_stones(p)];
    }
    __device__ static __forceinline__ int get_round(const int8_t* st) { return st[O_ROUND]; }
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return (int)(uint8_t)st[O_ROUND]; }
    __device__ static __forceinline__ bool move_uses_seed(int) { return true; }     // a move refills when one tile is left: state-dependent

    __device__ static void draw_tiles(int8_t* st, long long seed, bool initial, Rng& rng) {       // :503-518
        for (int i = initial ? 0 : 1; i < NSITE; i++) {
            uint64_t bits = 0;
            for (int j = 0; j < 8; j++) bits |= (uint64_t)(uint8_t)st[o_bitpack(j)] << (8 * (7 - j));     // tile t = bit 63 - t
            const int n = __popcll((unsigned long long)bits);
            int k;
            if (initial || seed == 0) {
                k = (int)(rng.u01() * (double)n);
                k = k >= n ? n - 1 : k;
            } else {
                long long v = (2014ll * (seed + (long long)st[O_ROUND]) + 42ll) % 61ll;
                if (v < 0) v += 61;
                k = (int)(v % n);
            }
            int tile = 0;
            for (int t = 0; t < 64; t++)
                if ((bits >> (63 - t)) & 1ull) { if (k == 0) { tile = t; break; } k--; }
            const uint32_t w = AKRO_TILES[tile];
            for (int j = 0; j < 3; j++) st[o_site(i, j)] = (int8_t)((w >> (4 * j)) & 15u);
            st[o_site(i, 3)] = (int8_t)tile;
            st[o_bitpack(tile >> 3)] = (int8_t)(((uint8_t)st[o_bitpack(tile >> 3)]) & ~(128u >> (tile & 7)));
        }
    }

    struct Bits169 {
        uint64_t w[3];
        __device__ void clear() { w[0] = w[1] = w[2] = 0; }
        __device__ bool get(int i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
        __device__ void set(int i) { w[i >> 6] |= 1ull << (i & 63); }
    };
    // update_districts :520-611, all lanes: a lane owns cells l, l + 64, l + 128.  The per-cell rules (green, isolated yellow, enclosed
    // purple, red next to the outside) are independent; the "outside" = the empty cells connected to the border is a flood fill done as
    // frontier sweeps over three 64-bit ballot words until nothing changes; the heaviest chain of houses (blue) is a connected-component
    // search over the blue cells only, left to lane 0 (a handful of cells).  Integer sums: any order gives the reference's numbers.
    __device__ static void update_districts(int8_t* st, int p) {
        const int l = lane_id();
        int green = 0, yellow = 0, purple = 0;
        uint64_t outer[3], empty[3], blue[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int i = 64 * c + l;
            const bool in = i < AREA;
            const int d = in ? st[o_descr(i, p)] : -1, h = in ? st[o_height(i, p)] : 0;
            bool border = false, yel_nb = false, open_nb = false;
            if (in) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int nb = neighbor(i, k);
                    border = border || nb < 0;
                    if (nb >= 0) {
                        yel_nb = yel_nb || st[o_descr(nb, p)] == DISTRICT_YELLOW;
                        open_nb = open_nb || st[o_height(nb, p)] == 0;
                    }
                }
            }
            if (d == DISTRICT_GREEN) green += h;
            if (d == DISTRICT_YELLOW && !yel_nb) yellow += h;
            if (d == DISTRICT_PURPLE && !border && !open_nb) purple += h;
            empty[c] = __ballot(d == EMPTY);
            outer[c] = __ballot(d == EMPTY && border);
            blue[c] = __ballot(d == DISTRICT_BLUE);
        }
        auto get = [](const uint64_t (&w)[3], int i) { return (w[i >> 6] >> (i & 63)) & 1ull; };
        // flood fill of the empty cells from the border
        for (;;) {
            uint64_t grow[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int i = 64 * c + l;
                bool g = false;
                if (i < AREA && ((empty[c] & ~outer[c]) >> l) & 1ull) {
#pragma unroll
                    for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); g = g || (nb >= 0 && get(outer, nb)); }
                }
                grow[c] = __ballot(g);
            }
            if (!(grow[0] | grow[1] | grow[2])) break;
            outer[0] |= grow[0]; outer[1] |= grow[1]; outer[2] |= grow[2];
        }
        int red = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int i = 64 * c + l;
            if (i < AREA && st[o_descr(i, p)] == DISTRICT_RED) {
                bool edge = false;
#pragma unroll
                for (int k = 0; k < 6; k++) { const int nb = neighbor(i, k); edge = edge || nb < 0 || get(outer, nb); }
                if (edge) red += st[o_height(i, p)];
            }
        }
        green = wave_sum_i32(green); yellow = wave_sum_i32(yellow); purple = wave_sum_i32(purple); red = wave_sum_i32(red);
        if (l == 0) {
            int best = 0;                                                          // heaviest chain of houses
            uint64_t todo[3] = {blue[0], blue[1], blue[2]};
            uint8_t stack[AREA];
            for (int c = 0; c < 3; c++)
                while (todo[c]) {
                    const int s0 = 64 * c + (__ffsll((unsigned long long)todo[c]) - 1);
                    int chain = 0, top = 0;
                    stack[top++] = (uint8_t)s0; todo[c] &= todo[c] - 1;
                    while (top) {
                        const int cur = stack[--top];
                        chain += st[o_height(cur, p)];
                        for (int k = 0; k < 6; k++) {
                            const int nb = neighbor(cur, k);
                            if (nb < 0 || !((todo[nb >> 6] >> (nb & 63)) & 1ull)) continue;
                            todo[nb >> 6] &= ~(1ull << (nb & 63)); stack[top++] = (uint8_t)nb;
                        }
                    }
                    best = chain > best ? chain : best;
                }
            st[o_districts(p, BLUE)] = (int8_t)best;
            st[o_districts(p, YELLOW)] = (int8_t)yellow;
            st[o_districts(p, RED)] = (int8_t)red;
            st[o_districts(p, PURPLE)] = (int8_t)purple;
            st[o_districts(p, GREEN)] = (int8_t)green;
        }
        wave_sync();
    }

    // valid_moves :358-398 for one pattern
    __device__ static bool pattern_valid(const int8_t* st, int pat, int player) {
        int c[3];
        if (!pattern_cells(pat, c)) return false;
        const int ha = st[o_height(c[0], player)];
        if (ha != st[o_height(c[1], player)] || ha != st[o_height(c[2], player)]) return false;
        if (ha == 0) {
            bool connected = false;
#pragma unroll 1
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 6; k++) {
                    const int nb = neighbor(c[j], k);
                    connected = connected || (nb >= 0 && st[o_height(nb, player)] > 0);      // (the triple itself has height 0)
                }
            return connected;
        }
        const int ta = st[o_tileid(c[0], player)];
        return !(ta == st[o_tileid(c[1], player)] && ta == st[o_tileid(c[2], player)]);
    }
    // Board.valid_moves :354-413: the pattern predicates go to mask_lds[0..15] first, then every action looks its pattern up; words are
    // finalised from the top so that a word is overwritten only after its last reader
    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
#pragma unroll 1
        for (int k = 0; k < 16; k++) {
            const int p = k * 64 + l;
            const uint64_t m = __ballot(p < NPAT && pattern_valid(st, p < NPAT ? p : 0, player));
            if (l == 0) mask_lds[k] = m;
        }
        wave_sync();
        int slots = st[o_stones(player)] + 1;
        slots = slots > NSITE ? NSITE : slots;
#pragma unroll 1
        for (int k = AW - 1; k >= 0; k--) {
            const int a = k * 64 + l;
            const int slot = a / NPAT, pat = a - slot * NPAT;
            bool v = a < A && slot < slots && st[o_site(slot < NSITE ? slot : 0, 0)] != EMPTY;
            v = v && ((mask_lds[pat >> 6] >> (pat & 63)) & 1ull);
            const uint64_t m = __ballot(v);
            wave_sync();
            if (l == 0) mask_lds[k] = m;
            wave_sync();
        }
    }

    // Board.make_move :314-352: lane 0 places the tile, all lanes rescore the player's districts, lane 0 finishes (score, round, refill)
    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        if (lane_id() == 0) place_tile(st, move, player);
        wave_sync();
        update_districts(st, player);
        int np = 0;
        if (lane_id() == 0) np = finish_move(st, player, seed, rng);
        np = __builtin_amdgcn_readfirstlane(np);
        rng.counter = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rng.counter >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rng.counter);
        wave_sync();
        return np;
    }
    __device__ static void place_tile(int8_t* st, int move, int player) {
        const int slot = move / NPAT, pat = move - slot * NPAT;
        int8_t tile[4];
        int c[3];
        for (int j = 0; j < 4; j++) tile[j] = st[o_site(slot, j)];
        for (int i = slot; i < NSITE - 1; i++)
            for (int j = 0; j < 4; j++) st[o_site(i, j)] = st[o_site(i + 1, j)];
        for (int j = 0; j < 4; j++) st[o_site(NSITE - 1, j)] = EMPTY;
        pattern_cells(pat, c);
        for (int j = 0; j < 3; j++) {
            const int under = st[o_descr(c[j], player)];
            if (type_of(under) == 3) st[o_plazas(player, color_of(under))] = (int8_t)(st[o_plazas(player, color_of(under))] - 1);
            if (type_of(under) == 1) st[o_stones(player)] = (int8_t)(st[o_stones(player)] + 1);
            st[o_descr(c[j], player)] = tile[j];
            st[o_height(c[j], player)] = (int8_t)(st[o_height(c[j], player)] + 1);
            st[o_tileid(c[j], player)] = tile[3];
            if (type_of(tile[j]) == 3) st[o_plazas(player, color_of(tile[j]))] = (int8_t)(st[o_plazas(player, color_of(tile[j]))] + 1);
        }
        st[o_stones(player)] = (int8_t)(st[o_stones(player)] - slot);
    }
    __device__ static int finish_move(int8_t* st, int player, long long seed, Rng& rng) {
        st[o_total(player)] = (int8_t)(get_score(st, player) / 2 - 128);        // encode_score_to_int8 :239-248
        st[O_ROUND] = (int8_t)(st[O_ROUND] + 1);
        if (st[o_site(1, 0)] == EMPTY && st[O_STACKS] > 0) {
            draw_tiles(st, seed, false, rng);
            st[O_STACKS] = (int8_t)(st[O_STACKS] - 1);
        }
        return player + 1 == NPL ? 0 : player + 1;
    }

    // Board.check_end_game :426-437 (uniform)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        if (!(st[O_STACKS] <= 0 && st[o_site(1, 0)] == EMPTY)) {
#pragma unroll
            for (int p = 0; p < NPL; p++) out[p] = 0.f;
            return false;
        }
        int proxy[NPL], m = -1, nmax = 0;
#pragma unroll
        for (int p = 0; p < NPL; p++) { proxy[p] = get_score(st, p) * 1000 + st[o_stones(p)]; m = proxy[p] > m ? proxy[p] : m; }
#pragma unroll
        for (int p = 0; p < NPL; p++) nmax += proxy[p] == m;
#pragma unroll
        for (int p = 0; p < NPL; p++) out[p] = proxy[p] == m ? (nmax == 1 ? 1.f : 0.001f) : -1.f;
        return true;
    }

    // Board.swap_players :439-470: new[p] = old[(p + k) % N] in the per-player planes and scalars
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        k = ((k % NPL) + NPL) % NPL;
        if (k == 0) return;
        for (int i = lane_id(); i < S; i += 64) tmp[i] = st[i];
        wave_sync();
        for (int i = lane_id(); i < S; i += 64) {
            const int cell = i / ST, z = i - cell * ST;
            int src = i;
            if (z < ZS) {
                const int b = z / NPL, p = z - b * NPL;
                src = cell * ST + b * NPL + (p + k) % NPL;
            } else if (z == ZS) {
                const int r = cell / CS, q = cell - r * CS;
                if (r < 3 * NPL && (r < 2 * NPL ? q < 5 : q < 2)) {
                    const int b = r / NPL, p = r - b * NPL;
                    src = at(b * NPL + (p + k) % NPL, q, ZS);
                }
            }
            st[i] = tmp[src];
        }
        wave_sync();
    }

    // init_game :275-295 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        for (int p = 0; p < NPL; p++) st[o_stones(p)] = (int8_t)(p + 1);
        for (int t = 0; t < 61; t++)
            if ((int)(AKRO_TILES[t] >> 12) <= NPL) st[o_bitpack(t >> 3)] = (int8_t)(((uint8_t)st[o_bitpack(t >> 3)]) | (128u >> (t & 7)));
        st[O_STACKS] = 11;
        for (int p = 0; p < NPL; p++) st[o_total(p)] = (int8_t)(st[o_stones(p)] / 2 - 128);
        const int centre = (CS / 2) * CS + CS / 2;
        for (int p = 0; p < NPL; p++) {
            st[o_descr(centre, p)] = PLAZA_BLUE; st[o_height(centre, p)] = 1; st[o_tileid(centre, p)] = 61;
            st[o_plazas(p, BLUE)] = 1;
            for (int d = 0; d < 6; d += 2) {                                   // NEIGHBORS[centre, ::2]
                const int nb = neighbor(centre, d);
                st[o_descr(nb, p)] = QUARRY; st[o_height(nb, p)] = 1; st[o_tileid(nb, p)] = 61;
            }
        }
        draw_tiles(st, 0, true, rng);
    }

    // ---- get_symmetries :472-501: six rotations about cell (0, 0), built by lane 0 as written ----
    static constexpr int NSYM_CAND = 6;
    __device__ static int rotate_cell(int idx, int k) {                        // :95-114
        if (idx < 0) return -1;
        const int r = idx / CS, q = idx - r * CS;
        int x = q - ((r - (r & 1)) / 2), z = r, y = -x - z;
        for (int i = 0; i < k; i++) { const int nx = -z, ny = -x, nz = -y; x = nx; y = ny; z = nz; }
        const int r2 = z, q2 = x + ((r2 - (r2 & 1)) / 2);
        return (r2 >= 0 && r2 < CS && q2 >= 0 && q2 < CS) ? r2 * CS + q2 : -1;
    }
    __device__ static int rotate_pattern(int pat, int k) {                     // :116-129: the FIRST pattern with the rotated cells
        int c[3], t[3], rc[3];
        pattern_cells(pat, c);
        for (int j = 0; j < 3; j++) rc[j] = rotate_cell(c[j], k);
        if (rc[0] < 0 && rc[1] < 0 && rc[2] < 0) return 0;                      // pattern 0 is (-1, -1, -1)
        if (rc[1] < 0) return -1;
        for (int o = 0; o < 6; o++) {
            if (!pattern_cells(rc[1] * 6 + o, t)) continue;
            if (t[0] == rc[0] && t[2] == rc[2]) return rc[1] * 6 + o;
        }
        return -1;
    }
    __device__ static bool sym_build(const int8_t* st, int c, int8_t* cand, int16_t* act_src, Rng& rng, const uint8_t* valids) {
        (void)rng;
        for (int i = 0; i < S; i++) cand[i] = (i % ST) >= ZS ? st[i] : 0;        // the scalar layers z = 3 N, 3 N + 1 are kept (:486)
        for (int i = 0; i < AREA; i++) {
            const int nb = rotate_cell(i, c);
            if (nb >= 0)
                for (int z = 0; z < ZS; z++) cand[nb * ST + z] = st[i * ST + z];
        }
        for (int a = 0; a < A; a++) act_src[a] = -1;
        for (int a = 0; a < A; a++)
            if (valids[a]) {
                const int slot = a / NPAT;
                int ni = slot * NPAT + rotate_pattern(a - slot * NPAT, c);
                if (ni < 0) ni += A;                                            // new_p[-1]: Python's wrap-around
                act_src[ni] = (int16_t)a;                                       // ascending scan: the last writer wins
            }
        return true;
    }
    // (the deterministic per-byte interface is unused: k_env_symmetries takes the built path)
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int c) { return c == 0; }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int, int i) { return st[i]; }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int, int a) { return a; }
};

}  // namespace azg
