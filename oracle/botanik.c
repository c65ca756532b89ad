/* ORACLE (test infrastructure).  Botanik env step: a scalar restatement of botanik/BotanikLogicNumba.py (Board :143-646, helpers
 * :648-787) with the shipped constants (botanik/BotanikConstants.py: MACHINE_SIZE = 7, two players).
 *
 * State = int8 [66][5][7] (copy_state :165-189), a "card" = 7 bytes {colour, flowers, type, N, E, S, W}:
 *   block 0 misc (5 x 7): [0][0] round, [0][1] status within the round, [0][2] main player, [0][3..4] = 1, [1][p] score of p,
 *           [3][c] / [4][c] high / low byte of the 13-bit mask of the cards of colour c still in the deck (card 0 = bit 12)
 *   block 1 arrival zone (3 cards), 2 / 3 registers of P0 / P1 (5), 4 middle row (5), 5 freed cards (P0: 0..1, P1: 2..3)
 *   blocks 6.. : six [7][7][7] arrays packed in 350-byte slabs: machine P0, machine P1, optim_neighbors P0 / P1 ([0] = cell is a
 *           candidate, [3..6] = has a neighbour N/E/S/W), optim_needpipes P0 / P1 ([3..6] = that neighbour has a pipe towards me)
 * 428 actions (:60-88): arrival card -> own register slot 0..14, -> middle row slot 15..29, swap mecabot with middle slot 30..34,
 * freed card k on cell yx with orientation o = 35 + 4 (49 k + yx) + o, throw the freed cards away 427.
 *
 * Randomness: make_move ignores random_seed; every card drawn from the deck (_draw_cards :414-438) takes one uniform of the
 * global RNG through my_random_choice (:112-115) = searchsorted(cumsum(mask / mask.sum()), u, 'right'), in MCTS simulations too.
 * The quirks of the source are kept: _compute_open_pipes scans 5 x 5 cells (:675-676), _swap_freed strides by 25 (:331-332),
 * the connected-component scoring follows the directed label equivalences exactly (:715-787). */
#include <string.h>
#include "azg_oracle.h"

enum { EMPTY = 0, SOURCE = 1 };
enum { PIPE2_ANGLE = 0, PIPE2_STRAIGHT = 1, PIPE3 = 2, PIPE4 = 3, PLANT = 4, VEGET = 5, MECABOT = 6 };
enum { TO_REGISTER = 0, OTHERP_EXPAND = 1, OTHERP_SWAP = 2, MAINPL_EXPAND = 3, MAINPL_SWAP = 4 };
enum { NORTH = 3, EAST = 4, SOUTH = 5, WEST = 6 };
#define MS 7
#define MM 49
#define CARD(s, block, i) ((s) + ((block) * 5 + (i)) * 7)
#define MISC(s, i, z) ((s)[(i) * 7 + (z)])
#define ARRIVAL(s, i) CARD(s, 1, i)
#define REG(s, p, i) CARD(s, 2 + (p), i)
#define MIDDLE(s, i) CARD(s, 4, i)
#define FREED(s, i) CARD(s, 5, i)
#define MACH(s, p) ((s) + 210 + 350 * (p))
#define NEIGH(s, p) ((s) + 910 + 350 * (p))
#define NEEDP(s, p) ((s) + 1610 + 350 * (p))
#define CELL(m, y, x) ((m) + ((y) * MS + (x)) * 7)

/* np_all_cards[colour][k] (BotanikConstants.py:57-80): {flowers, type, N, E, S, W} */
static const int8_t GENERIC[13][6] = {{0, 0, 0, 1, 1, 0}, {0, 0, 0, 1, 1, 0}, {1, 0, 0, 1, 1, 0}, {0, 1, 1, 0, 1, 0}, {0, 1, 1, 0, 1, 0},
                                      {1, 1, 1, 0, 1, 0}, {0, 2, 0, 1, 1, 1}, {0, 2, 0, 1, 1, 1}, {1, 2, 0, 1, 1, 1}, {0, 3, 1, 1, 1, 1},
                                      {3, 4, 0, 0, 1, 0}, {3, 5, 0, 0, 1, 0}, {0, 6, 0, 0, 0, 0}};

static int is_empty(const int8_t* c) { return c[0] == EMPTY; }
static int is_mecabot(const int8_t* c) { return c[2] == MECABOT; }

/* _draw_cards(1) :414-438; returns 0 when the deck is empty (nothing drawn, no uniform consumed) */
static int draw_card(int8_t* s, azo_rng* rng, int8_t* out) {
    int avail[65], k = 0;
    for (int c = 0; c < 5; c++) {
        const unsigned m = ((unsigned)(uint8_t)MISC(s, 3, c) << 8) | (uint8_t)MISC(s, 4, c);
        for (int i = 0; i < 13; i++) { avail[13 * c + i] = (m >> (12 - i)) & 1; k += avail[13 * c + i]; }
    }
    if (k == 0) return 0;
    const double u = azo_rng_u01(rng);
    double acc = 0.0;
    int pick = 64;
    for (int i = 0; i < 65; i++) {
        acc += (avail[i] ? 1.0 : 0.0) / (double)k;
        if (acc > u) { pick = i; break; }
    }
    const int c = pick / 13, i = pick % 13;
    unsigned m = ((unsigned)(uint8_t)MISC(s, 3, c) << 8) | (uint8_t)MISC(s, 4, c);
    m &= ~(1u << (12 - i));
    MISC(s, 3, c) = (int8_t)(uint8_t)(m >> 8);
    MISC(s, 4, c) = (int8_t)(uint8_t)(m & 255u);
    out[0] = (int8_t)(c + 2);
    memcpy(out + 1, GENERIC[i], 6);
    return 1;
}
/* _draw_cards_to_arrival_zone :440-443: three cards, or none when the deck is empty (the deck holds 5 + 3 x 20 cards) */
static void draw_arrival(int8_t* s, azo_rng* rng) {
    int8_t cards[3][7];
    for (int i = 0; i < 3; i++)
        if (!draw_card(s, rng, cards[i])) return;
    for (int i = 0; i < 3; i++) memcpy(ARRIVAL(s, i), cards[i], 7);
}

static void update_optims(int8_t* s, int p, int y, int x) {                                      /* :615-627 */
    static const int DY[4] = {-1, 0, 1, 0}, DX[4] = {0, 1, 0, -1};
    int8_t *mach = MACH(s, p), *nei = NEIGH(s, p), *need = NEEDP(s, p);
    for (int o = 0; o < 4; o++) {
        const int ny = y + DY[o], nx = x + DX[o];
        if (ny < 0 || ny >= MS || nx < 0 || nx >= MS) continue;
        const int opp = (o + 2) % 4 + 3;
        CELL(nei, ny, nx)[0] = (int8_t)is_empty(CELL(mach, ny, nx));
        CELL(nei, ny, nx)[opp] = 1;
        CELL(need, ny, nx)[opp] = (int8_t)(CELL(mach, y, x)[3 + o] > 0);
    }
    memset(CELL(nei, y, x), 0, 7);
    memset(CELL(need, y, x), 0, 7);
}

static int open_pipes(const int8_t* mach) {                                                       /* _compute_open_pipes :672-686 */
    int n = 0;
    const int m = MS - 1;
    for (int y = 0; y < 5; y++)
        for (int x = 0; x < 5; x++) {
            const int8_t* c = CELL(mach, y, x);
            if (is_empty(c)) continue;
            if (y > 0 && is_empty(CELL(mach, y - 1, x)) && c[NORTH] > 0) n++;
            if (x < m && is_empty(CELL(mach, y, x + 1)) && c[EAST] > 0) n++;
            if (y < m && is_empty(CELL(mach, y + 1, x)) && c[SOUTH] > 0) n++;
            if (x > 0 && is_empty(CELL(mach, y, x - 1)) && c[WEST] > 0) n++;
        }
    return n;
}

/* _check_card_on_machine :688-713 for one orientation */
static int check_card(const int8_t* card, int y, int x, const int8_t* need, const int8_t* nei, int initial_open, int orient) {
    const int m = MS - 1;
    if (card[2] == PIPE2_STRAIGHT && orient >= 2) return 0;
    if (card[2] == PIPE4 && orient >= 1) return 0;
    const int inb[4] = {y > 0, x < m, y < m, x > 0};
    int card_pipes = 0, closed = 0;
    for (int i = 0; i < 4; i++) {
        const int oc = card[3 + ((i - orient + 4) & 3)];                                         /* np.roll(card[NORTH:], orient) */
        const int pwn = oc * nei[3 + i];
        if (pwn != need[3 + i]) return 0;
        card_pipes += oc * inb[i];
        closed += pwn;
    }
    return initial_open - closed + (card_pipes - closed) > 0;
}

/* ---- _compute_score :715-787 ---- */
typedef struct { uint8_t visited[MM]; int8_t labels[MM]; uint64_t equiv[MM + 1]; int ncards[MM + 1], nflow[MM + 1], n; } score_ctx;
static void dfs(const int8_t* mach, int y, int x, score_ctx* k) {
    const int m = MS - 1;
    const int8_t* c = CELL(mach, y, x);
    int ny[4], nx[4], nn = 0;
    k->visited[y * MS + x] = 1;
    if (y > 0 && c[NORTH] > 0) { ny[nn] = y - 1; nx[nn] = x; nn++; }
    if (x < m && c[EAST] > 0) { ny[nn] = y; nx[nn] = x + 1; nn++; }
    if (y < m && c[SOUTH] > 0) { ny[nn] = y + 1; nx[nn] = x; nn++; }
    if (x > 0 && c[WEST] > 0) { ny[nn] = y; nx[nn] = x - 1; nn++; }
    int nl[4], nnl = 0, best = 99;
    for (int i = 0; i < nn; i++)
        if (CELL(mach, ny[i], nx[i])[0] == c[0]) { nl[nnl] = k->labels[ny[i] * MS + nx[i]]; if (nl[nnl] < best) best = nl[nnl]; nnl++; }
    if (best == 99) {
        best = k->n;
        k->equiv[k->n] = 1ull << k->n; k->ncards[k->n] = 1; k->nflow[k->n] = c[1];
        k->n++;
    } else {
        for (int i = 0; i < nnl; i++)
            if (nl[i] != 99) k->equiv[nl[i]] |= 1ull << best;
        k->ncards[best] += 1; k->nflow[best] += c[1];
    }
    k->labels[y * MS + x] = (int8_t)best;
    for (int i = 0; i < nn; i++)
        if (!is_empty(CELL(mach, ny[i], nx[i])) && !k->visited[ny[i] * MS + nx[i]]) dfs(mach, ny[i], nx[i], k);
}
static void score_sum(const score_ctx* k, uint8_t* seen, uint64_t to_visit, int* cards, int* flowers) {
    for (int i = 0; i < k->n; i++)
        if (((to_visit >> i) & 1) && !seen[i]) {
            *cards += k->ncards[i]; *flowers += k->nflow[i];
            seen[i] = 1;
            score_sum(k, seen, k->equiv[i], cards, flowers);
        }
}
static int compute_score(const int8_t* mach) {
    score_ctx k;
    memset(&k, 0, sizeof(k));
    memset(k.labels, 99, sizeof(k.labels));
    dfs(mach, MS / 3, MS / 2, &k);
    uint8_t seen[MM + 1] = {0};
    int total = 0;
    for (int a = 1; a < k.n; a++) {
        int c = 0, f = 0;
        score_sum(&k, seen, 1ull << a, &c, &f);
        total += c >= 3 ? c + f : f;
    }
    return total;
}

static void free_card_if_needed(int8_t* s, int slot) {                                           /* :505-547 */
    const int mc = MIDDLE(s, slot)[0], mt = MIDDLE(s, slot)[2];
    for (int p = 0; p < 2; p++) {
        int8_t* reg = REG(s, p, slot);
        if (is_empty(reg) || reg[0] == mc || reg[2] == mt) continue;
        const int ns = is_empty(FREED(s, 2 * p)) ? 0 : 1;                                         /* both busy: slot 1 is overwritten, as written */
        memcpy(FREED(s, 2 * p + ns), reg, 7);
        memset(reg, 0, 7);
        const int is_main = p == MISC(s, 0, 2);
        int status;
        if (is_mecabot(FREED(s, 2 * p + ns))) {
            status = is_main ? MAINPL_SWAP : OTHERP_SWAP;
            if (ns != 0) {                                                                        /* the mecabot goes first */
                int8_t t[7];
                memcpy(t, FREED(s, 2 * p + 1), 7);
                memcpy(FREED(s, 2 * p + 1), FREED(s, 2 * p), 7);
                memcpy(FREED(s, 2 * p), t, 7);
            }
        } else {
            status = is_main ? MAINPL_EXPAND : OTHERP_EXPAND;
        }
        if (status > MISC(s, 0, 1)) MISC(s, 0, 1) = (int8_t)status;
    }
}
static void next_status(int8_t* s) {                                                              /* :592-604, 632-644 */
    const int mainpl = MISC(s, 0, 2), otherp = 1 - mainpl;
    if (is_mecabot(FREED(s, 2 * mainpl))) MISC(s, 0, 1) = MAINPL_SWAP;                            /* (the source raises here) */
    else if (!is_empty(FREED(s, 2 * mainpl))) MISC(s, 0, 1) = MAINPL_EXPAND;
    else if (is_mecabot(FREED(s, 2 * otherp))) MISC(s, 0, 1) = OTHERP_SWAP;
    else if (!is_empty(FREED(s, 2 * otherp))) MISC(s, 0, 1) = OTHERP_EXPAND;
    else MISC(s, 0, 1) = TO_REGISTER;
}

void botanik_valid_moves(const azo_game* g, const int8_t* s, int player, uint8_t* out) {          /* :191-201, 445-486 */
    memset(out, 0, (size_t)g->A);
    const int status = MISC(s, 0, 1);
    if (status == TO_REGISTER) {
        for (int i = 0; i < 3; i++) {
            const int8_t* card = ARRIVAL(s, i);
            if (is_empty(card)) continue;
            for (int k = 0; k < 5; k++) {
                out[5 * i + k] = (uint8_t)(is_empty(REG(s, player, k)) && (MIDDLE(s, k)[0] == card[0] || MIDDLE(s, k)[2] == card[2]));
                out[15 + 5 * i + k] = 1;
            }
        }
    } else if (status == MAINPL_SWAP || status == OTHERP_SWAP) {
        for (int k = 0; k < 5; k++) out[30 + k] = (uint8_t)(MIDDLE(s, k)[2] != MECABOT);
    } else if (status == MAINPL_EXPAND || status == OTHERP_EXPAND) {
        const int8_t *mach = MACH(s, player), *nei = NEIGH(s, player), *need = NEEDP(s, player);
        const int nb_open = open_pipes(mach);
        int any = 0;
        for (int k = 0; k < 2; k++) {
            const int8_t* card = FREED(s, 2 * player + k);
            if (is_empty(card)) continue;
            for (int y = 0; y < MS; y++)
                for (int x = 0; x < MS; x++) {
                    if (!CELL(nei, y, x)[0]) continue;
                    for (int o = 0; o < 4; o++) {
                        const int v = check_card(card, y, x, CELL(need, y, x), CELL(nei, y, x), nb_open, o);
                        out[35 + 4 * (MM * k + MS * y + x) + o] = (uint8_t)v;
                        any |= v;
                    }
                }
        }
        if (!any) out[g->A - 1] = 1;
    }
}

int botanik_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t seed, azo_rng* rng) {   /* :203-230 */
    (void)seed;
    if (move < 15) {                                                                              /* _move_to_register :488-495 */
        memcpy(REG(s, player, move % 5), ARRIVAL(s, move / 5), 7);
        memset(ARRIVAL(s, move / 5), 0, 7);
    } else if (move < 30) {                                                                       /* _move_to_middle_row_and_unlink :497-503 */
        const int ci = (move - 15) / 5, slot = (move - 15) % 5;
        memcpy(MIDDLE(s, slot), ARRIVAL(s, ci), 7);
        memset(ARRIVAL(s, ci), 0, 7);
        free_card_if_needed(s, slot);
    } else if (move < 35) {                                                                       /* _swap_mecabot :549-567 */
        const int slot = move - 30;
        int8_t t[7];
        memcpy(t, FREED(s, 2 * player), 7);
        memcpy(FREED(s, 2 * player), MIDDLE(s, slot), 7);
        memcpy(MIDDLE(s, slot), t, 7);
        if (MISC(s, 0, 1) == MAINPL_SWAP) MISC(s, 0, 1) = MAINPL_EXPAND;
        else if (MISC(s, 0, 1) == OTHERP_SWAP) MISC(s, 0, 1) = OTHERP_EXPAND;
        free_card_if_needed(s, slot);
    } else if (move < g->A - 1) {                                                                 /* _expand_machine :569-604 */
        const int ci = (move - 35) / (4 * MM), rem = (move - 35) % (4 * MM), slot = rem / 4, orient = rem % 4;
        const int y = slot / MS, x = slot % MS;
        int8_t* cell = CELL(MACH(s, player), y, x);
        const int8_t* card = FREED(s, 2 * player + ci);
        cell[0] = card[0]; cell[1] = card[1]; cell[2] = card[2];
        for (int i = 0; i < 4; i++) cell[3 + i] = card[3 + ((i - orient + 4) & 3)];
        memset(FREED(s, 2 * player + ci), 0, 7);
        update_optims(s, player, y, x);
        if (ci == 0 && !is_empty(FREED(s, 2 * player + 1))) {
            memcpy(FREED(s, 2 * player), FREED(s, 2 * player + 1), 7);
            memset(FREED(s, 2 * player + 1), 0, 7);
        }
        MISC(s, 1, player) = (int8_t)compute_score(MACH(s, player));
        next_status(s);
    } else {                                                                                      /* _throw_cards_away :629-644 */
        memset(FREED(s, 2 * player), 0, 14);
        next_status(s);
    }
    const int status = MISC(s, 0, 1);
    int mainpl = MISC(s, 0, 2);
    if (status == TO_REGISTER) {
        if (is_empty(ARRIVAL(s, 0)) && is_empty(ARRIVAL(s, 1)) && is_empty(ARRIVAL(s, 2))) draw_arrival(s, rng);
        MISC(s, 0, 0) = (int8_t)(MISC(s, 0, 0) + 1);
        mainpl = 1 - mainpl;
        MISC(s, 0, 2) = (int8_t)mainpl;
        return mainpl;
    }
    return (status == MAINPL_EXPAND || status == MAINPL_SWAP) ? mainpl : 1 - mainpl;
}

void botanik_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) {        /* :235-252 */
    (void)g; (void)next_player;
    out[0] = out[1] = 0.f;
    for (int z = 0; z < 7; z++) if (MISC(s, 3, z) != 0 || MISC(s, 4, z) != 0) return;
    for (int i = 0; i < 3; i++) if (!is_empty(ARRIVAL(s, i))) return;
    for (int i = 0; i < 4; i++) if (!is_empty(FREED(s, i))) return;
    int a = MISC(s, 1, 0), b = MISC(s, 1, 1);
    if (a == b) {
        a = b = 0;
        for (int i = 0; i < MM; i++) { a += MACH(s, 0)[7 * i] != 0; b += MACH(s, 1)[7 * i] != 0; }
    }
    if (a > b) { out[0] = 1.f; out[1] = -1.f; }
    else if (a < b) { out[0] = -1.f; out[1] = 1.f; }
    else out[0] = out[1] = 0.01f;
}

void botanik_swap_players(const azo_game* g, int8_t* s, int k) {                                  /* :254-284 */
    (void)g;
    if (k != 1) return;
    int8_t t[350];
    memcpy(t, REG(s, 0, 0), 35); memcpy(REG(s, 0, 0), REG(s, 1, 0), 35); memcpy(REG(s, 1, 0), t, 35);
    memcpy(t, FREED(s, 0), 14); memcpy(FREED(s, 0), FREED(s, 2), 14); memcpy(FREED(s, 2), t, 14);
    if (MISC(s, 0, 1) > TO_REGISTER) MISC(s, 0, 1) = (int8_t)((MISC(s, 0, 1) + 1) % 4 + 1);
    MISC(s, 0, 2) = (int8_t)(1 - MISC(s, 0, 2));
    const int8_t sc = MISC(s, 1, 0); MISC(s, 1, 0) = MISC(s, 1, 1); MISC(s, 1, 1) = sc;
    /* the machine arrays are [7][7][7] = 343 bytes at the start of their 350-byte slabs: the 7 tail bytes are not swapped */
    memcpy(t, MACH(s, 0), 343); memcpy(MACH(s, 0), MACH(s, 1), 343); memcpy(MACH(s, 1), t, 343);
    memcpy(t, NEIGH(s, 0), 343); memcpy(NEIGH(s, 0), NEIGH(s, 1), 343); memcpy(NEIGH(s, 1), t, 343);
    memcpy(t, NEEDP(s, 0), 343); memcpy(NEEDP(s, 0), NEEDP(s, 1), 343); memcpy(NEEDP(s, 1), t, 343);
}

int botanik_get_round(const azo_game* g, const int8_t* s) { (void)g; return MISC(s, 0, 0); }
int botanik_get_score(const azo_game* g, const int8_t* s, int p) { (void)g; return MISC(s, 1, p); }

void botanik_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                             /* :152-163, 606-613 */
    memset(s, 0, (size_t)g->S);
    for (int c = 0; c < 5; c++) { MISC(s, 3, c) = 0x1F; MISC(s, 4, c) = (int8_t)0xFF; }
    for (int i = 0; i < 5; i++) draw_card(s, rng, MIDDLE(s, i));
    draw_arrival(s, rng);
    static const int8_t SRC[7] = {SOURCE, 0, 0, 0, 0, 1, 0};
    for (int p = 0; p < 2; p++) memcpy(CELL(MACH(s, p), MS / 3, MS / 2), SRC, 7);
    MISC(s, 0, 3) = 1; MISC(s, 0, 4) = 1;
    for (int p = 0; p < 2; p++) update_optims(s, p, MS / 3, MS / 2);
}

/* ---- get_symmetries :286-409 (always on the canonical board) ---- */
static void mirror_machine(int8_t* mach) {                                                        /* :293-305 */
    const int m = MS - 1;
    for (int y = 0; y < MS; y++)
        for (int x = 0; x < (MS + 1) / 2; x++) {
            int8_t *a = CELL(mach, y, x), *b = CELL(mach, y, m - x), t[7];
            if (m - x != x) {
                memcpy(t, a, 7); memcpy(a, b, 7); memcpy(b, t, 7);
                int8_t w = b[EAST]; b[EAST] = b[WEST]; b[WEST] = w;
            }
            int8_t w = a[EAST]; a[EAST] = a[WEST]; a[WEST] = w;
        }
}
static void roll_colors(int8_t* cards, int n, int nroll) {                                        /* :356-367 */
    for (int i = 0; i < n; i++) {
        const int c = cards[7 * i];
        if (c != EMPTY && c != SOURCE) cards[7 * i] = (int8_t)((c - 2 + nroll) % 5 + 2);
    }
}
static const int PERM_ARRIVAL[3][3] = {{0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
static const int PERM_REG[5][5] = {{0, 3, 2, 4, 1}, {1, 0, 3, 2, 4}, {2, 4, 1, 0, 3}, {3, 2, 4, 1, 0}, {4, 1, 0, 3, 2}};

int botanik_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                       uint8_t* ov, int max_sym) {
    const int S = g->S, A = g->A, m = MS - 1;
    int k = 0;
#define BEGIN() if (k >= max_sym) return k; int8_t* st = os + (size_t)k * S; float* p = op + (size_t)k * A; uint8_t* v = ov + (size_t)k * A; \
                memcpy(st, s, (size_t)S); memcpy(p, pi, sizeof(float) * (size_t)A); memcpy(v, valids, (size_t)A)
    { BEGIN(); (void)st; (void)p; (void)v; k++; }
    {   /* mirror of machine 0 + policy / valids (:307-324) */
        BEGIN();
        mirror_machine(MACH(st, 0));
        for (int y = 0; y < MS; y++)
            for (int x = 0; x < (MS + 1) / 2; x++)
                for (int ci = 0; ci < 2; ci++) {
                    const int type = FREED(st, ci)[2];
                    for (int o = 0; o < 4; o++) {
                        static const int ANGLE[4] = {1, 0, 3, 2}, OTHER[4] = {0, 3, 2, 1};
                        const int no = type == PIPE2_ANGLE ? ANGLE[o] : OTHER[o];
                        const int a1 = 35 + 4 * (MM * ci + MS * y + x), a2 = 35 + 4 * (MM * ci + MS * y + m - x);
                        if (m - x != x) {
                            p[a2 + no] = pi[a1 + o]; p[a1 + no] = pi[a2 + o];
                            v[a2 + no] = valids[a1 + o]; v[a1 + no] = valids[a2 + o];
                        } else {
                            p[a1 + no] = pi[a1 + o]; v[a1 + no] = valids[a1 + o];
                        }
                    }
                }
        k++;
    }
    { BEGIN(); (void)p; (void)v; mirror_machine(MACH(st, 1)); k++; }
    if (!is_empty(FREED(s, 0)) && !is_empty(FREED(s, 1))) {                                       /* _swap_freed :326-337 */
        BEGIN();
        for (int yx = 0; yx < MM; yx++)
            for (int o = 0; o < 4; o++) {
                const int a1 = 35 + 4 * yx + o, a2 = 35 + 4 * (25 + yx) + o;
                p[a2] = pi[a1]; p[a1] = pi[a2];
                v[a2] = valids[a1]; v[a1] = valids[a2];
            }
        int8_t t[7];
        memcpy(t, FREED(st, 0), 7); memcpy(FREED(st, 0), FREED(st, 1), 7); memcpy(FREED(st, 1), t, 7);
        k++;
    }
    for (int q = 0; q < 3; q++) {                                                                 /* _permute_arrival :339-345 */
        BEGIN();
        for (int i = 0; i < 3; i++) {
            const int ni = PERM_ARRIVAL[q][i];
            memcpy(ARRIVAL(st, ni), ARRIVAL(s, i), 7);
            for (int j = 0; j < 5; j++) {
                p[5 * ni + j] = pi[5 * i + j]; p[5 * ni + 15 + j] = pi[5 * i + 15 + j];
                v[5 * ni + j] = valids[5 * i + j]; v[5 * ni + 15 + j] = valids[5 * i + 15 + j];
            }
        }
        k++;
    }
    for (int q = 0; q < 5; q++) {                                                                 /* _permute_registers :347-354 */
        BEGIN();
        for (int i = 0; i < 5; i++) {
            const int ni = PERM_REG[q][i];
            memcpy(REG(st, 0, ni), REG(s, 0, i), 7); memcpy(REG(st, 1, ni), REG(s, 1, i), 7); memcpy(MIDDLE(st, ni), MIDDLE(s, i), 7);
            for (int z = 0; z < 7; z++) { p[z * 5 + ni] = pi[z * 5 + i]; v[z * 5 + ni] = valids[z * 5 + i]; }
        }
        k++;
    }
    for (int q = 0; q < 2; q++) {                                                                 /* colour rolls :393-403 */
        BEGIN();
        (void)p; (void)v;
        const int nroll = q == 0 ? 2 : 4;
        roll_colors(ARRIVAL(st, 0), 5, nroll); roll_colors(REG(st, 0, 0), 5, nroll); roll_colors(REG(st, 1, 0), 5, nroll);
        roll_colors(MIDDLE(st, 0), 5, nroll); roll_colors(FREED(st, 0), 5, nroll);
        roll_colors(MACH(st, 0), MM, nroll); roll_colors(MACH(st, 1), MM, nroll);
        k++;
    }
#undef BEGIN
    return k;
}
