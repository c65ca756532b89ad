/* ORACLE (test infrastructure, not product).  Santorini 5x5, NB_GODS = 1 (no gods) or 11 (basic gods), INIT_METHOD 0/1.
 * Scalar C restatement of santorini/SantoriniLogicNumba.py `Board` and SantoriniConstants.py (lines cited).
 * State = int8[5][5][3] interleaved (workers, levels, gods_power), byte-identical to board.tobytes(). */
#include <string.h>
#include "azg_oracle.h"

enum { NO_GOD = 0, APOLLO, MINOTAUR, ATLAS, HEPHAESTUS, ARTEMIS, DEMETER, HERMES, PAN, ATHENA, PROMETHEUS };
#define NO_MOVE 4
#define NO_BUILD 4
#define MAX_ITER_FOR_HERMES 5

#define W(st, pos) ((st)[(pos) * 3 + 0])
#define LV(st, pos) ((st)[(pos) * 3 + 1])
#define GP(st, i) ((st)[(i) * 3 + 2])      /* gods_power.flat[i] */

typedef struct { int y, x; } pos_t;

static pos_t apply_dir(pos_t p, int d) { pos_t r = { p.y + d / 3 - 1, p.x + d % 3 - 1 }; return r; }   /* :56-70 */
static int in_grid(pos_t p) { return p.y >= 0 && p.y < 5 && p.x >= 0 && p.x < 5; }
static int idx(pos_t p) { return p.y * 5 + p.x; }

static int encode_action(int nb, int worker, int power, int md, int bd) {                /* Constants.py:31-34 */
    return nb * 81 * worker + 81 * power + 9 * md + bd;
}

static pos_t worker_pos(const int8_t* st, int id) {                                      /* :667-673 */
    for (int i = 0; i < 25; i++) if (W(st, i) == id) { pos_t p = { i / 5, i % 5 }; return p; }
    pos_t bad = { -1, -1 };
    return bad;
}

static int able_to_push(const int8_t* st, pos_t np) {                                    /* :702-716 */
    if (!in_grid(np)) return 0;
    if (W(st, idx(np)) != 0) return 0;
    if (LV(st, idx(np)) > 3) return 0;
    return 1;
}

static int able_to_move(const int8_t* st, pos_t old, pos_t np, int player, int no_climb, int swap, int push) { /* :675-699 */
    if (old.y == np.y && old.x == np.x) return 1;
    if (!in_grid(np)) return 0;
    int w = W(st, idx(np));
    if (w != 0) {
        int is_opp = player == 0 ? (w == -1 || w == -2) : (w == 1 || w == 2);
        if ((swap || push) && is_opp) {
            if (push) {
                pos_t pp = { 2 * np.y - old.y, 2 * np.x - old.x };                         /* _position_if_pushed :52-53 */
                if (!able_to_push(st, pp)) return 0;
            }
        } else return 0;
    }
    int nl = LV(st, idx(np));
    if (nl > 3) return 0;
    int ol = LV(st, idx(old));
    if (nl > ol + (no_climb ? 0 : 1)) return 0;
    return 1;
}

static int able_to_build(const int8_t* st, pos_t p, int ignore, int two_levels, int dome) {  /* :718-729 */
    if (!in_grid(p)) return 0;
    int w = W(st, idx(p));
    if (!(w == 0 || w == ignore)) return 0;
    if (LV(st, idx(p)) >= (two_levels ? 2 : (dome ? 3 : 4))) return 0;
    return 1;
}

void santorini_valid_moves(const azo_game* g, const int8_t* st, int player, uint8_t* out) {   /* :125-432 */
    const int NB = g->variant;
    memset(out, 0, (size_t)g->A);
    int opponent = (player + 1) % 2;
    int opp_athena = GP(st, ATHENA + NB * opponent) > 64;
    int sgn = player == 0 ? 1 : -1;
#define ENC(w, p, m, b) out[encode_action(NB, (w), (p), (m), (b))] = 1
    /* ordered elif chain exactly as the reference (:135, :154, :177, ...) */
    int god = -1;
    for (int k = 0; k <= PROMETHEUS; k++) if (GP(st, k + NB * player) > 0) { god = k; break; }
    if (god < 0) return; /* 'Should not happen vm' :430 */

    if (god == NO_GOD || god == PAN || god == ATHENA || god == APOLLO || god == MINOTAUR || god == ATLAS ||
        god == HEPHAESTUS) {
        int no_climb = (god == ATHENA) ? 0 : opp_athena;        /* :383 Athena's own moves ignore it */
        for (int worker = 0; worker < 2; worker++) {
            int wid = (worker + 1) * sgn;
            pos_t old = worker_pos(st, wid);
            for (int md = 0; md < 9; md++) {
                if (md == NO_MOVE) continue;
                pos_t np = apply_dir(old, md);
                int use_power = 0;
                if (!able_to_move(st, old, np, player, no_climb, 0, 0)) {
                    if (god == APOLLO && able_to_move(st, old, np, player, no_climb, 1, 0)) use_power = 1;      /* :163-166 */
                    else if (god == MINOTAUR && able_to_move(st, old, np, player, no_climb, 0, 1)) use_power = 1; /* :186-189 */
                    else continue;
                }
                for (int bd = 0; bd < 9; bd++) {
                    if (bd == NO_BUILD) continue;
                    pos_t bp = apply_dir(np, bd);
                    if (god == ATLAS) {                                                     /* :200-218 */
                        if (able_to_build(st, bp, wid, 0, 0)) ENC(worker, NO_GOD, md, bd);
                        if (able_to_build(st, bp, wid, 0, 1)) ENC(worker, ATLAS, md, bd);
                    } else if (god == HEPHAESTUS) {                                         /* :221-239 */
                        if (able_to_build(st, bp, wid, 0, 0)) ENC(worker, NO_GOD, md, bd);
                        if (able_to_build(st, bp, wid, 1, 0)) ENC(worker, HEPHAESTUS, md, bd);
                    } else {
                        if (!able_to_build(st, bp, wid, 0, 0)) continue;
                        ENC(worker, use_power ? god : NO_GOD, md, bd);
                    }
                }
            }
        }
    } else if (god == ARTEMIS) {                                                            /* :242-281 */
        int avoid = GP(st, ARTEMIS + NB * player) % 64 - 1;
        if (avoid < 0) {
            for (int worker = 0; worker < 2; worker++) {
                int wid = (worker + 1) * sgn;
                pos_t old = worker_pos(st, wid);
                for (int md = 0; md < 9; md++) {
                    if (md == NO_MOVE) continue;
                    pos_t np = apply_dir(old, md);
                    if (!able_to_move(st, old, np, player, opp_athena, 0, 0)) continue;
                    for (int bd = 0; bd < 9; bd++) {
                        if (bd == NO_BUILD) ENC(worker, ARTEMIS, md, bd);
                        else {
                            pos_t bp = apply_dir(np, bd);
                            if (!able_to_build(st, bp, wid, 0, 0)) continue;
                            ENC(worker, NO_GOD, md, bd);
                        }
                    }
                }
            }
        } else {
            int worker = avoid / 9, wid = (worker + 1) * sgn;
            pos_t old = worker_pos(st, wid);
            for (int md = 0; md < 9; md++) {
                if (md == NO_MOVE || md == avoid % 9) continue;
                pos_t np = apply_dir(old, md);
                if (!able_to_move(st, old, np, player, opp_athena, 0, 0)) continue;
                for (int bd = 0; bd < 9; bd++) {
                    if (bd == NO_BUILD) continue;
                    pos_t bp = apply_dir(np, bd);
                    if (!able_to_build(st, bp, wid, 0, 0)) continue;
                    ENC(worker, NO_GOD, md, bd);
                }
            }
        }
    } else if (god == DEMETER) {                                                            /* :284-318 */
        int avoid = GP(st, DEMETER + NB * player) % 64 - 1;
        if (avoid < 0) {
            for (int worker = 0; worker < 2; worker++) {
                int wid = (worker + 1) * sgn;
                pos_t old = worker_pos(st, wid);
                for (int md = 0; md < 9; md++) {
                    if (md == NO_MOVE) continue;
                    pos_t np = apply_dir(old, md);
                    if (!able_to_move(st, old, np, player, opp_athena, 0, 0)) continue;
                    for (int bd = 0; bd < 9; bd++) {
                        if (bd == NO_BUILD) continue;
                        pos_t bp = apply_dir(np, bd);
                        if (!able_to_build(st, bp, wid, 0, 0)) continue;
                        ENC(worker, DEMETER, md, bd);
                    }
                }
            }
        } else {
            int worker = avoid / 9, wid = (worker + 1) * sgn;
            pos_t old = worker_pos(st, wid);
            for (int bd = 0; bd < 9; bd++) {
                if (bd == NO_BUILD) ENC(worker, NO_GOD, NO_MOVE, NO_BUILD);
                else {
                    pos_t bp = apply_dir(old, bd);
                    if (bd == avoid % 9) continue;
                    if (!able_to_build(st, bp, wid, 0, 0)) continue;
                    ENC(worker, NO_GOD, NO_MOVE, bd);
                }
            }
        }
    } else if (god == HERMES) {                                                             /* :321-351 */
        int nb_prev = GP(st, HERMES + NB * player) % 64;
        for (int worker = 0; worker < 2; worker++) {
            int wid = (worker + 1) * sgn;
            pos_t old = worker_pos(st, wid);
            int cur_level = LV(st, idx(old));
            for (int bd = 0; bd < 9; bd++)
                if (bd != NO_BUILD) {
                    pos_t bp = apply_dir(old, bd);
                    if (able_to_build(st, bp, wid, 0, 0)) ENC(worker, NO_GOD, NO_MOVE, bd);
                }
            if (nb_prev < MAX_ITER_FOR_HERMES)
                for (int md = 0; md < 9; md++)
                    if (md != NO_MOVE) {
                        pos_t np = apply_dir(old, md);
                        if (able_to_move(st, old, np, player, opp_athena, 0, 0))
                            if (LV(st, idx(np)) == cur_level) ENC(worker, HERMES, md, NO_BUILD);
                    }
            if (nb_prev == 0)
                for (int md = 0; md < 9; md++)
                    if (md != NO_MOVE) {
                        pos_t np = apply_dir(old, md);
                        if (able_to_move(st, old, np, player, opp_athena, 0, 0))
                            for (int bd = 0; bd < 9; bd++)
                                if (bd != NO_BUILD) {
                                    pos_t bp = apply_dir(np, bd);
                                    if (able_to_build(st, bp, wid, 0, 0)) ENC(worker, NO_GOD, md, bd);
                                }
                    }
        }
    } else if (god == PROMETHEUS) {                                                         /* :392-428 */
        int v = GP(st, PROMETHEUS + NB * player) % 64 - 1;
        /* Python floor division: (-1)//9 == -1 */
        int prev = v < 0 ? -1 : v / 9;
        if (prev < 0) {
            for (int worker = 0; worker < 2; worker++) {
                int wid = (worker + 1) * sgn;
                pos_t old = worker_pos(st, wid);
                for (int md = 0; md < 9; md++) {
                    int use_power = (md == NO_MOVE);
                    pos_t np = apply_dir(old, md);
                    if (!able_to_move(st, old, np, player, opp_athena, 0, 0)) continue;
                    for (int bd = 0; bd < 9; bd++) {
                        if (bd == NO_BUILD) continue;
                        pos_t bp = apply_dir(np, bd);
                        if (!able_to_build(st, bp, wid, 0, 0)) continue;
                        ENC(worker, use_power ? PROMETHEUS : NO_GOD, md, bd);
                    }
                }
            }
        } else {
            int worker = prev, wid = (worker + 1) * sgn;
            pos_t old = worker_pos(st, wid);
            for (int md = 0; md < 9; md++) {
                if (md == NO_MOVE) continue;
                pos_t np = apply_dir(old, md);
                if (!able_to_move(st, old, np, player, 1, 0, 0)) continue;
                for (int bd = 0; bd < 9; bd++) {
                    if (bd == NO_BUILD) continue;
                    pos_t bp = apply_dir(np, bd);
                    if (!able_to_build(st, bp, wid, 0, 0)) continue;
                    ENC(worker, NO_GOD, md, bd);
                }
            }
        }
    }
#undef ENC
}

int santorini_make_move(const azo_game* g, int8_t* st, int move, int player, int64_t seed, azo_rng* rng) {  /* :434-550 */
    (void)seed; (void)rng;
    const int NB = g->variant;
    int opponent_next = 1;
    int worker = move / (NB * 81), rem = move % (NB * 81);                                 /* _decode_action */
    int power = rem / 81; rem %= 81;
    int md = rem / 9, bd = rem % 9;
    int wid = (worker + 1) * (player == 0 ? 1 : -1);
    pos_t old = worker_pos(st, wid);
    pos_t np = apply_dir(old, md);
    switch (power) {
    case NO_GOD: {
        int old_level = LV(st, idx(old));
        W(st, idx(old)) = 0; W(st, idx(np)) = (int8_t)wid;
        if (bd != NO_BUILD) { pos_t bp = apply_dir(np, bd); LV(st, idx(bp)) += 1; }
        if (GP(st, PAN + NB * player) > 0) {
            int nl = LV(st, idx(np));
            if (nl <= old_level - 2) GP(st, PAN + NB * player) = 64 + 1;
        } else if (GP(st, ATHENA + NB * player) > 0) {
            int nl = LV(st, idx(np));
            GP(st, ATHENA + NB * player) = (int8_t)(64 + (nl > old_level ? 1 : 0));
        } else {
            for (int i = player * NB; i < (player + 1) * NB; i++)
                if (GP(st, i) > 64) GP(st, i) = 64;
        }
        break; }
    case APOLLO: {
        int8_t a = W(st, idx(old)), b = W(st, idx(np));
        W(st, idx(old)) = b; W(st, idx(np)) = a;
        pos_t bp = apply_dir(np, bd); LV(st, idx(bp)) += 1;
        break; }
    case MINOTAUR: {
        pos_t pp = { 2 * np.y - old.y, 2 * np.x - old.x };
        int8_t a = W(st, idx(old)), b = W(st, idx(np));
        W(st, idx(old)) = 0; W(st, idx(np)) = a; W(st, idx(pp)) = b;
        pos_t bp = apply_dir(np, bd); LV(st, idx(bp)) += 1;
        break; }
    case ATLAS: {
        W(st, idx(old)) = 0; W(st, idx(np)) = (int8_t)wid;
        pos_t bp = apply_dir(np, bd); LV(st, idx(bp)) = 4;
        break; }
    case HEPHAESTUS: {
        W(st, idx(old)) = 0; W(st, idx(np)) = (int8_t)wid;
        pos_t bp = apply_dir(np, bd); LV(st, idx(bp)) += 2;
        break; }
    case ARTEMIS:
        W(st, idx(old)) = 0; W(st, idx(np)) = (int8_t)wid;
        GP(st, ARTEMIS + NB * player) = (int8_t)(64 + (worker * 9 + (8 - md) + 1));
        opponent_next = 0;
        break;
    case DEMETER: {
        W(st, idx(old)) = 0; W(st, idx(np)) = (int8_t)wid;
        pos_t bp = apply_dir(np, bd); LV(st, idx(bp)) += 1;
        GP(st, DEMETER + NB * player) = (int8_t)(64 + (worker * 9 + bd + 1));
        opponent_next = 0;
        break; }
    case HERMES:
        W(st, idx(old)) = 0; W(st, idx(np)) = (int8_t)wid;
        GP(st, HERMES + NB * player) += 1;
        opponent_next = 0;
        break;
    case PROMETHEUS: {
        pos_t bp = apply_dir(old, bd); LV(st, idx(bp)) += 1;
        opponent_next = 0;
        GP(st, PROMETHEUS + NB * player) = (int8_t)(64 + (worker * 9 + 1));
        break; }
    default: break;
    }
    if (GP(st, 2 * NB) < 127) GP(st, 2 * NB) += 1;                                          /* :544-545 */
    return opponent_next ? 1 - player : player;
}

int santorini_get_score(const azo_game* g, const int8_t* st, int player) {                  /* :84-97 */
    (void)g;
    int hi = 0;
    for (int i = 0; i < 25; i++) {
        int w = W(st, i), l = LV(st, i);
        if ((player == 0 ? w > 0 : w < 0) && l > hi) hi = l;
    }
    return hi;
}

void santorini_game_ended(const azo_game* g, const int8_t* st, int next_player, float* out) {  /* :552-565 */
    const int NB = g->variant;
    out[0] = out[1] = 0.f;
    if (santorini_get_score(g, st, 0) == 3 || GP(st, PAN + NB * 0) > 64) { out[0] = 1.f; out[1] = -1.f; return; }
    if (santorini_get_score(g, st, 1) == 3 || GP(st, PAN + NB * 1) > 64) { out[0] = -1.f; out[1] = 1.f; return; }
    uint8_t vm[2 * 11 * 81];
    santorini_valid_moves(g, st, next_player, vm);
    int s = 0;
    for (int i = 0; i < g->A; i++) s += vm[i];
    if (s == 0) {
        if (next_player == 0) { out[0] = -1.f; out[1] = 1.f; }
        else { out[0] = 1.f; out[1] = -1.f; }
    }
}

void santorini_swap_players(const azo_game* g, int8_t* st, int k) {                         /* :567-576 */
    const int NB = g->variant;
    if (k != 1) return;
    for (int i = 0; i < 25; i++) W(st, i) = (int8_t)(-W(st, i));
    int8_t cp[22];
    for (int i = 0; i < 2 * NB; i++) cp[i] = GP(st, i);
    for (int i = 0; i < 2 * NB; i++) GP(st, i) = cp[(i + NB) % (2 * NB)];
}

int santorini_get_round(const azo_game* g, const int8_t* st) { return GP(st, 2 * g->variant); }   /* :655-656 */

/* init_game :99-120 with INIT_METHOD == 1 (random distinct cells for [1,-1,2,-2], random distinct gods); our RNG
   stream (partial Fisher-Yates, j = i + floor(u*(n-i))) replaces np.random.choice(replace=False). */
void santorini_init_board(const azo_game* g, int8_t* st, azo_rng* rng) {
    const int NB = g->variant;
    memset(st, 0, 75);
    int cells[25];
    for (int i = 0; i < 25; i++) cells[i] = i;
    static const int8_t wl[4] = { 1, -1, 2, -2 };
    for (int i = 0; i < 4; i++) {
        int j = i + (int)(azo_rng_u01(rng) * (25 - i));
        if (j > 24) j = 24;
        int t = cells[i]; cells[i] = cells[j]; cells[j] = t;
        W(st, cells[i]) = wl[i];
    }
    int g0 = NO_GOD, g1 = NO_GOD;
    if (NB > 1) {
        int gods[10];
        for (int i = 0; i < NB - 1; i++) gods[i] = i;
        for (int i = 0; i < 2; i++) {
            int j = i + (int)(azo_rng_u01(rng) * (NB - 1 - i));
            if (j > NB - 2) j = NB - 2;
            int t = gods[i]; gods[i] = gods[j]; gods[j] = t;
        }
        g0 = gods[0] + 1; g1 = gods[1] + 1;
    }
    GP(st, g0 + NB * 0) = 64;
    GP(st, g1 + NB * 1) = 64;
}

/* ---- get_symmetries :578-653 ---------------------------------------------------------------------------------- */
static const int ROT_CORE[9] = { 6, 3, 0, 7, 4, 1, 8, 5, 2 };       /* Constants.py:60 */
static const int FLR_CORE[9] = { 2, 1, 0, 5, 4, 3, 8, 7, 6 };       /* :68 */
static const int FUD_CORE[9] = { 6, 7, 8, 3, 4, 5, 0, 1, 2 };       /* :77 */

static void perm_policy(int NB, const int* core, const float* pi, const uint8_t* va, float* opi, uint8_t* ova) {
    int A = NB * 162;
    memcpy(opi, pi, sizeof(float) * (size_t)A);
    memcpy(ova, va, (size_t)A);
    for (int i = 0; i < A; i++) {
        int worker = i / (NB * 81), rem = i % (NB * 81);
        int power = rem / 81; rem %= 81;
        int md = rem / 9, bd = rem % 9;
        int ni = encode_action(NB, worker, power, core[md], core[bd]);
        opi[ni] = pi[i]; ova[ni] = va[i];
    }
}

static void perm_gods(int NB, const int* core, int8_t* st) {          /* _apply_permutation_gods :589-595 */
    for (int i = 0; i < 2 * NB; i++) {
        if (i % NB == ARTEMIS || i % NB == DEMETER) {
            int v = GP(st, i);
            if (v < 65) continue;
            int k = v - 65, worker = k / 9, dir = k % 9;
            GP(st, i) = (int8_t)(65 + 9 * worker + core[dir]);
        }
    }
}

static void map_grid(int8_t* st, const int8_t* src, int kind) {
    /* kind 0: np.rot90 (counter-clockwise): out[i][j] = in[j][4-i]; 1: fliplr; 2: flipud  -- planes 0,1 only */
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
            int si = kind == 0 ? j : (kind == 1 ? i : 4 - i);
            int sj = kind == 0 ? 4 - i : (kind == 1 ? 4 - j : j);
            W(st, i * 5 + j) = W(src, si * 5 + sj);
            LV(st, i * 5 + j) = LV(src, si * 5 + sj);
        }
}

static void swap_workers_gods(int NB, int8_t* st, int player) {        /* :630-636 */
    for (int i = NB * player; i < NB * (player + 1); i++)
        if (i % NB == ARTEMIS || i % NB == DEMETER || i % NB == ATHENA) {
            int v = GP(st, i);
            if (v < 65) continue;
            GP(st, i) = (int8_t)((v - 65 + 9) % 18 + 65);
        }
}

int santorini_symmetries(const azo_game* g, const int8_t* st, const float* pi, const uint8_t* va, int8_t* os,
                         float* op, uint8_t* ov, int max_sym) {
    const int NB = g->variant, A = g->A, S = 75;
    if (max_sym < 8) return -1;
    int k = 0;
    memcpy(os, st, S); memcpy(op, pi, sizeof(float) * (size_t)A); memcpy(ov, va, (size_t)A); k++;
    int8_t cur[75], nxt[75];
    memcpy(cur, st, S);
    const float* ppi = pi; const uint8_t* pva = va;
    for (int r = 0; r < 3; r++) {
        memcpy(nxt, cur, S);
        map_grid(nxt, cur, 0);
        perm_gods(NB, ROT_CORE, nxt);
        perm_policy(NB, ROT_CORE, ppi, pva, op + (size_t)k * A, ov + (size_t)k * A);
        memcpy(os + (size_t)k * S, nxt, S);
        memcpy(cur, nxt, S);
        ppi = op + (size_t)k * A; pva = ov + (size_t)k * A;
        k++;
    }
    for (int f = 1; f <= 2; f++) {
        memcpy(nxt, st, S);
        map_grid(nxt, st, f);
        perm_gods(NB, f == 1 ? FLR_CORE : FUD_CORE, nxt);
        perm_policy(NB, f == 1 ? FLR_CORE : FUD_CORE, pi, va, op + (size_t)k * A, ov + (size_t)k * A);
        memcpy(os + (size_t)k * S, nxt, S);
        k++;
    }
    /* own-worker swap :638-643 */
    memcpy(nxt, st, S);
    { pos_t w1 = worker_pos(st, 1), w2 = worker_pos(st, 2); W(nxt, idx(w1)) = 2; W(nxt, idx(w2)) = 1; }
    swap_workers_gods(NB, nxt, 0);
    memcpy(os + (size_t)k * S, nxt, S);
    memcpy(op + (size_t)k * A, pi + A / 2, sizeof(float) * (size_t)(A / 2));
    memcpy(op + (size_t)k * A + A / 2, pi, sizeof(float) * (size_t)(A / 2));
    memcpy(ov + (size_t)k * A, va + A / 2, (size_t)(A / 2));
    memcpy(ov + (size_t)k * A + A / 2, va, (size_t)(A / 2));
    k++;
    /* opponent-worker swap :646-651 */
    memcpy(nxt, st, S);
    { pos_t w1 = worker_pos(st, -1), w2 = worker_pos(st, -2); W(nxt, idx(w1)) = -2; W(nxt, idx(w2)) = -1; }
    swap_workers_gods(NB, nxt, 1);
    memcpy(os + (size_t)k * S, nxt, S);
    memcpy(op + (size_t)k * A, pi, sizeof(float) * (size_t)A);
    memcpy(ov + (size_t)k * A, va, (size_t)A);
    k++;
    return k;
}

/* RNG-free start state of SURVEY.md Appendix C.1: the INIT_METHOD == 0 layout (:105-106) with gods g0 / g1 */
void santorini_known_start(const azo_game* g, int8_t* st, int g0, int g1) {
    const int NB = g->variant;
    memset(st, 0, 75);
    W(st, 2 * 5 + 1) = 1; W(st, 2 * 5 + 3) = 2; W(st, 1 * 5 + 2) = -1; W(st, 3 * 5 + 2) = -2;
    GP(st, g0 + NB * 0) = 64;
    GP(st, g1 + NB * 1) = 64;
}
