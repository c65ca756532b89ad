/* ORACLE (test infrastructure).  Minivilles (Machi Koro) env step: a scalar restatement of
 * minivilles/MinivillesLogicNumba.py (Board, :66-372) for 2-4 players.
 *
 * State = int8 [18 + 20 n][2], column 0 = current state, column 1 = the state before the last "real" move (history used by the
 * radio-tower re-roll, :270-274); rows (copy_state :164-175): 0 round, 1 last dice, 2 player_state, 3..17 market[15],
 * 18.. money[n], 18+n.. cards[n][15], 18+16n.. monuments[n][4].
 *
 * Randomness: make_move ignores random_seed -- the dice (:232-242) and the three "random" choices of the purple cards (:49-52)
 * always draw from the global NumPy / Numba RNG, inside MCTS simulations too.  Here every draw is one uniform of the caller's
 * azo_rng: np.random.randint(1, 6) -> 1 + floor(5 u) (a die of this game shows 1..5: NumPy's upper bound is exclusive);
 * my_random_choice_and_normalize(mask) = searchsorted(cumsum(mask), u, 'right') with u in [0, 1) = the FIRST set index, but
 * the uniform is still consumed. */
#include <string.h>
#include "azg_oracle.h"

enum { CHAMPS, FERME, BOULANGERIE, CAFE, SUPERETTE, FORET, STADE, AFFAIRES, CHAINE, FROMAGERIE, MEUBLES, MINE, RESTAURANT, VERGER,
       MARCHE };
enum { GARE, CENTRECOM, RADIO, PARC };
static const int8_t CARDS_COST[15] = {1, 1, 1, 2, 2, 3, 6, 8, 7, 5, 3, 6, 3, 3, 2};      /* :391 */
static const int8_t MONU_COST[4] = {4, 10, 16, 22};                                      /* :392 */

#define AT(s, row, col) ((s)[2 * (row) + (col)])
#define R_MARKET 3
#define R_MONEY 18
#define R_CARDS(n) (18 + (n))
#define R_MONU(n) (18 + 16 * (n))

static int rnd_int(azo_rng* rng, int lo, int hi) { int v = lo + (int)(azo_rng_u01(rng) * (double)(hi - lo)); return v >= hi ? hi - 1 : v; }
static int first_set(azo_rng* rng, const int* mask, int n) {                            /* :49-52 */
    (void)azo_rng_u01(rng);
    for (int i = 0; i < n; i++) if (mask[i]) return i;
    return n;
}

static int mv_score(const azo_game* g, const int8_t* s, int p) {                         /* get_score :78-79 */
    int n = g->P, t = 0;
    for (int m = 0; m < 4; m++) t += AT(s, R_MONU(n) + 4 * p + m, 0) * MONU_COST[m];
    return t;
}
static int mv_wealth(const azo_game* g, const int8_t* s, int p) {                        /* get_wealth :81-84 */
    int w = mv_score(g, s, p) + AT(s, R_MONEY + p, 0);
    return w > 127 ? 127 : w;
}
static void add_money(int8_t* s, int p, int amount) {                                    /* _add_money :355-361 */
    int v = AT(s, R_MONEY + p, 0) + amount;
    if (v > 127) v = 127;
    if (v < 0) v = 0;
    AT(s, R_MONEY + p, 0) = (int8_t)v;
}
static int cards(const azo_game* g, const int8_t* s, int p, int c) { return AT(s, R_CARDS(g->P) + 15 * p + c, 0); }
static int monu(const azo_game* g, const int8_t* s, int p, int m) { return AT(s, R_MONU(g->P) + 4 * p + m, 0); }

static void roll_dice(const azo_game* g, const int8_t* s, int who, azo_rng* rng, int* dice, int* identical) {   /* :232-242 */
    int d = rnd_int(rng, 1, 6);
    *identical = 0;
    if (monu(g, s, who, GARE) > 0) {
        int d2 = rnd_int(rng, 1, 6);
        *identical = d == d2;
        d += d2;
    }
    *dice = d;
}

static void all_receive(const azo_game* g, int8_t* s, int c, int money) {
    for (int p = 0; p < g->P; p++) add_money(s, p, money * cards(g, s, p, c));
}
static void current_receive(const azo_game* g, int8_t* s, int who, int c, int money, int mall) {
    int bonus = (mall && monu(g, s, who, CENTRECOM) > 0) ? 1 : 0;
    add_money(s, who, (money + bonus) * cards(g, s, who, c));
}
static void current_give(const azo_game* g, int8_t* s, int who, int c, int money, int mall) {                  /* :259-267 */
    for (int pl = who + g->P - 1; pl > who; pl--) {
        int p = pl % g->P;
        int bonus = (mall && monu(g, s, p, CENTRECOM) > 0) ? 1 : 0;
        int amount = (money + bonus) * cards(g, s, p, c);
        int have = AT(s, R_MONEY + who, 0);
        if (have < amount) amount = have;
        /* the reference credits the card OWNER with -amount and the roller with +amount (:264-265): as written */
        add_money(s, p, -amount);
        add_money(s, who, amount);
    }
}

static void dice_effect(const azo_game* g, int8_t* s, int result, int who, azo_rng* rng) {                     /* :244-353 */
    const int n = g->P;
    switch (result) {
    case 1: all_receive(g, s, CHAMPS, 1); break;
    case 2: all_receive(g, s, FERME, 1); current_receive(g, s, who, BOULANGERIE, 1, 1); break;
    case 3: current_give(g, s, who, CAFE, 1, 1); current_receive(g, s, who, BOULANGERIE, 1, 1); break;
    case 4: current_receive(g, s, who, SUPERETTE, 3, 1); break;
    case 5: all_receive(g, s, FORET, 1); break;
    case 6:
        if (cards(g, s, who, STADE) > 0) {                                                                    /* _stadium :269-278 */
            for (int p = 0; p < n; p++) {
                if (p == who) continue;
                int amount = AT(s, R_MONEY + p, 0) < 2 ? AT(s, R_MONEY + p, 0) : 2;
                add_money(s, p, -amount);
                add_money(s, who, amount);
            }
        }
        if (cards(g, s, who, AFFAIRES) > 0) {                                                                 /* _business_center :280-303 */
            int w[AZO_MAX_PLAYERS], mask[15], mx = -128;
            for (int p = 0; p < n; p++) w[p] = (int8_t)mv_wealth(g, s, p);
            w[who] = 0;
            for (int p = 0; p < n; p++) if (w[p] > mx) mx = w[p];
            for (int p = 0; p < n; p++) mask[p] = w[p] == mx;
            int target = first_set(rng, mask, n);
            int cost[15], cmx = -128;
            for (int c = 0; c < 15; c++) cost[c] = (cards(g, s, target, c) < 1 ? cards(g, s, target, c) : 1) * CARDS_COST[c];
            cost[STADE] = cost[AFFAIRES] = cost[CHAINE] = 0;
            for (int c = 0; c < 15; c++) if (cost[c] > cmx) cmx = cost[c];
            for (int c = 0; c < 15; c++) mask[c] = cost[c] == cmx;
            int tb = first_set(rng, mask, 15);
            int mine[15], mmn = 127;
            for (int c = 0; c < 15; c++) {
                mine[c] = (cards(g, s, who, c) < 1 ? cards(g, s, who, c) : 1) * CARDS_COST[c];
                if (mine[c] == 0) mine[c] = 99;
            }
            for (int c = 0; c < 15; c++) if (mine[c] < mmn) mmn = mine[c];
            for (int c = 0; c < 15; c++) mask[c] = mine[c] == mmn;
            int mb = first_set(rng, mask, 15);
            AT(s, R_CARDS(n) + 15 * target + tb, 0) -= 1;
            AT(s, R_CARDS(n) + 15 * who + tb, 0) += 1;
            AT(s, R_CARDS(n) + 15 * who + mb, 0) -= 1;
            AT(s, R_CARDS(n) + 15 * target + mb, 0) += 1;
        }
        if (cards(g, s, who, CHAINE) > 0) {                                                                   /* _tv_channel :305-319 */
            int money[AZO_MAX_PLAYERS], mx = -128, mask[AZO_MAX_PLAYERS], w[AZO_MAX_PLAYERS], wmx = -128;
            for (int p = 0; p < n; p++) money[p] = AT(s, R_MONEY + p, 0);
            money[who] = 0;
            for (int p = 0; p < n; p++) if (money[p] > mx) mx = money[p];
            if (mx > 5) mx = 5;
            for (int p = 0; p < n; p++) w[p] = (money[p] == mx || money[p] >= 5) ? (int8_t)mv_wealth(g, s, p) : 0;
            for (int p = 0; p < n; p++) if (w[p] > wmx) wmx = w[p];
            for (int p = 0; p < n; p++) mask[p] = w[p] == wmx;
            int target = first_set(rng, mask, n);
            int amount = AT(s, R_MONEY + target, 0) < 5 ? AT(s, R_MONEY + target, 0) : 5;
            add_money(s, target, -amount);
            add_money(s, who, amount);
        }
        break;
    case 7: current_receive(g, s, who, FROMAGERIE, 3 * cards(g, s, who, FERME), 0); break;
    case 8: current_receive(g, s, who, MEUBLES, 3 * (cards(g, s, who, FORET) + cards(g, s, who, MINE)), 0); break;
    case 9: current_give(g, s, who, RESTAURANT, 2, 1); all_receive(g, s, MINE, 5); break;
    case 10: current_give(g, s, who, RESTAURANT, 2, 1); all_receive(g, s, VERGER, 3); break;
    case 11: case 12: current_receive(g, s, who, MARCHE, 2 * (cards(g, s, who, CHAMPS) + cards(g, s, who, VERGER)), 0); break;
    default: break;
    }
}

void minivilles_valid_moves(const azo_game* g, const int8_t* s, int player, uint8_t* out) {                    /* :104-110,220-247 */
    const int money = AT(s, R_MONEY + player, 0);
    for (int c = 0; c < 15; c++) out[c] = money >= CARDS_COST[c] && AT(s, R_MARKET + c, 0) > 0;
    if (cards(g, s, player, STADE) > 0) out[STADE] = 0;
    if (cards(g, s, player, AFFAIRES) > 0) out[AFFAIRES] = 0;
    if (cards(g, s, player, CHAINE) > 0) out[CHAINE] = 0;
    for (int m = 0; m < 4; m++) out[15 + m] = money >= MONU_COST[m] && monu(g, s, player, m) == 0;
    /* `self.players_monuments[4*player+3,0] and self.player_state[0]%2 == 0` (:245-247): index 3 = PARC as written */
    out[19] = (monu(g, s, player, 3) != 0) && (AT(s, 2, 0) % 2 == 0);
    out[20] = 1;
}

int minivilles_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t random_seed, azo_rng* rng) {   /* :112-160 */
    const int n = g->P, rows = 18 + 20 * n;
    (void)random_seed;
    if (move < 15) {                                                                                          /* _buy_card :249-252 */
        add_money(s, player, -CARDS_COST[move]);
        AT(s, R_MARKET + move, 0) -= 1;
        AT(s, R_CARDS(n) + 15 * player + move, 0) += 1;
    } else if (move < 19) {                                                                                   /* _buy_monument :254-256 */
        add_money(s, player, -MONU_COST[move - 15]);
        AT(s, R_MONU(n) + 4 * player + (move - 15), 0) += 1;
    } else if (move == 19) {                                                                                  /* _dice_again :258-263 */
        for (int r = R_MARKET; r < rows; r++) AT(s, r, 0) = AT(s, r, 1);
        AT(s, 0, 0) = AT(s, 0, 1);
    }
    int next;
    if (move == 19) next = player;
    else if (AT(s, 2, 0) >= 2) { AT(s, 0, 0) = (int8_t)(AT(s, 0, 0) + 1); next = player; }
    else { AT(s, 0, 0) = (int8_t)(AT(s, 0, 0) + 1); next = (player + 1) % n; }
    if (move != 19) {
        for (int r = R_MARKET; r < rows; r++) AT(s, r, 1) = AT(s, r, 0);
        AT(s, 0, 1) = AT(s, 0, 0);
    }
    int dice, identical;
    roll_dice(g, s, next, rng, &dice, &identical);
    AT(s, 1, 0) = (int8_t)dice;
    dice_effect(g, s, dice, next, rng);
    AT(s, 2, 0) = (int8_t)((move == 19 ? 1 : 0) + (identical ? 2 : 0));
    return next;
}

void minivilles_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) {                 /* :177-185 */
    const int n = g->P;
    int sc[AZO_MAX_PLAYERS], mx = -128, rich = 0, cnt = 0;
    (void)next_player;
    for (int p = 0; p < n; p++) { sc[p] = (int8_t)mv_score(g, s, p); if (sc[p] > mx) mx = sc[p]; }
    for (int p = 0; p < n; p++) rich |= AT(s, R_MONEY + p, 0) >= 126;
    if (mx < 52 && AT(s, 0, 0) < 126 && !rich) { for (int p = 0; p < n; p++) out[p] = 0.f; return; }
    for (int p = 0; p < n; p++) cnt += sc[p] == mx;
    for (int p = 0; p < n; p++) out[p] = sc[p] == mx ? (cnt == 1 ? 1.f : 0.01f) : -1.f;
}

void minivilles_swap_players(const azo_game* g, int8_t* s, int k) {                                           /* :189-198 */
    const int n = g->P;
    int8_t tmp[2 * 20 * AZO_MAX_PLAYERS];
    struct { int row0, rows, unit; } blk[3] = {{R_MONEY, n, 1}, {R_CARDS(n), 15 * n, 15}, {R_MONU(n), 4 * n, 4}};
    for (int b = 0; b < 3; b++) {
        const int size = blk[b].rows, shift = blk[b].unit * k;
        memcpy(tmp, s + 2 * blk[b].row0, (size_t)(2 * size));
        for (int i = 0; i < size; i++) {
            const int src = ((i + shift) % size + size) % size;
            AT(s, blk[b].row0 + i, 0) = tmp[2 * src];
            AT(s, blk[b].row0 + i, 1) = tmp[2 * src + 1];
        }
    }
}

int minivilles_get_round(const azo_game* g, const int8_t* s) { (void)g; return AT(s, 0, 0); }
int minivilles_get_score(const azo_game* g, const int8_t* s, int p) { return mv_score(g, s, p); }

void minivilles_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                                      /* init_game :86-102 */
    const int n = g->P;
    memset(s, 0, (size_t)g->S);
    for (int c = 0; c < 15; c++) { AT(s, R_MARKET + c, 0) = 6; AT(s, R_MARKET + c, 1) = 6; }
    for (int c = 6; c < 9; c++) { AT(s, R_MARKET + c, 0) = 4; AT(s, R_MARKET + c, 1) = 4; }
    for (int p = 0; p < n; p++) {
        AT(s, R_MONEY + p, 0) = AT(s, R_MONEY + p, 1) = 3;
        for (int c = 0; c < 2; c++) AT(s, R_CARDS(n) + 15 * p + c, 0) = AT(s, R_CARDS(n) + 15 * p + c, 1) = 1;
    }
    int dice, identical;
    roll_dice(g, s, 0, rng, &dice, &identical);
    AT(s, 1, 0) = (int8_t)dice;
    dice_effect(g, s, dice, 0, rng);
}

int minivilles_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                          uint8_t* ov, int max_sym) {                                                          /* :200-202 */
    if (max_sym < 1) return 0;
    memcpy(os, s, (size_t)g->S);
    memcpy(op, pi, sizeof(float) * (size_t)g->A);
    memcpy(ov, valids, (size_t)g->A);
    return 1;
}
