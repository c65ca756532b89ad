/* ORACLE (test infrastructure).  Smallworld env step: a scalar restatement of smallworld/SmallworldLogicNumba.py (Board :141-1508)
 * for NUMBER_PLAYERS = 2 (the shipped constant), 3 and 4 with their maps (smallworld/SmallworldMaps_<n>pl.py: 23 / 30 / 39 areas,
 * 10 / 10 / 9 rounds).
 *
 * State = int8 [NB_AREAS + 5 n + 7][8] (:11-60; indices below for 2 players): rows 0..22 territories {nb people, people type (negative in decline, -15 lost tribe), power, defence
 * due to the people, defence due to the power, total defence, points if scored now, owner (-1 none)}; rows 23 + 3 p + id peoples of
 * player p (id 0 declined-spirit, 1 declined, 2 active) {nb in hand, type, power, people data, power data, -, points, player};
 * rows 29..34 visible deck {nb, type, power, -, -, -, coins, -1}; rows 35 + p round status {people on the map, -, -, #NETWDT, phase,
 * total defence, points preview, player}; rows 37 + p game status {-, -, -, round, id of the people playing (-1 not my turn), -, score
 * - 128, player}; row 39 invisible deck {2 bytes of available peoples, 3 bytes of available powers (MSB first), dice count, deck draw
 * count, -}.  131 actions (:76-88): abandon 0..22, attack 23..45, people action 46..68, power action 69..91, redeploy 92..122
 * (92 skip, 93..99 n on each area, 100..122 one on area), choose 123..128, decline 129, end 130.
 *
 * Randomness: with random_seed != 0 (MCTS simulations) dice and deck draws are functions of the seed and of the two counters in the
 * invisible deck (:421-424, 1378-1381); with random_seed == 0 (real moves, init) they are np.random.choice(x), defined by the RNG
 * contract as x[floor(u len)]; get_symmetries (:281-299) draws two score offsets with np.random.randint(lo, hi) = lo + floor(u (hi - lo))
 * (tools/refshim/harness.py CounterRandom feeds the reference the same). */
#include <string.h>
#include "azg_oracle.h"
#include "smallworld_tables.h"

enum { FORESTT, FARMLAND, HILLT, SWAMPT, MOUNTAIN, WATER };
enum { DECK_SIZE = 6, SCORE_INIT = 5, IMMUNITY = 20, MAX_REDEPLOY = 8, MAX_DICE = 3, MAXNA = 39, MAXNP = 4 };
/* the map of the bound game (NUMBER_PLAYERS picks it, SmallworldMaps.py): set by sw_bind() at every entry point */
static __thread int NA = SW_NA_2, NP = 2, NB_ROUNDS = SW_ROUNDS_2;
static __thread const uint8_t* SW_DESCR = SW_DESCR_2;
static __thread const uint64_t* SW_CONN = SW_CONN_2;
static void sw_bind(const azo_game* g) {
    NP = g->P;
    if (NP == 2) { NA = SW_NA_2; NB_ROUNDS = SW_ROUNDS_2; SW_DESCR = SW_DESCR_2; SW_CONN = SW_CONN_2; }
    else if (NP == 3) { NA = SW_NA_3; NB_ROUNDS = SW_ROUNDS_3; SW_DESCR = SW_DESCR_3; SW_CONN = SW_CONN_3; }
    else { NA = SW_NA_4; NB_ROUNDS = SW_ROUNDS_4; SW_DESCR = SW_DESCR_4; SW_CONN = SW_CONN_4; }
}
enum { DECLINED_SPIRIT = 0, DECLINED = 1, ACTIVE = 2 };
enum { PHASE_READY = 1, PHASE_CHOOSE, PHASE_ABANDON, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS, PHASE_REDEPLOY,
       PHASE_STOUT_TO_DECLINE, PHASE_WAIT };
enum { NOPPL = 0, AMAZON, DWARF, ELF, GHOUL, GIANT, HALFLING, HUMAN, ORC, RATMAN, SKELETON, SORCERER, TRITON, TROLL, WIZARD, LOST_TRIBE = -15 };
enum { MAX_SKELETONS = 20, MAX_SORCERERS = 18 };
enum { NOPOWER = 0, ALCHEMIST, BERSERK, BIVOUACKING, COMMANDO, DIPLOMAT, DRAGONMASTER, FLYING, FOREST, FORTIFIED, HEROIC, HILL, MERCHANT,
       MOUNTED, PILLAGING, SEAFARING, SPIRIT, STOUT, SWAMP, UNDERWORLD, WEALTHY };
static const int8_t DICE_VALUES[6] = {0, 0, 0, 1, 2, 3};

#define T(s, a) ((s) + 8 * (a))
#define PPL(s, p, id) ((s) + 8 * (NA + 3 * (p) + (id)))
#define DECK(s, i) ((s) + 8 * (NA + 3 * NP + (i)))
#define RS(s, p) ((s) + 8 * (NA + 3 * NP + DECK_SIZE + (p)))
#define GS(s, p) ((s) + 8 * (NA + 4 * NP + DECK_SIZE + (p)))
#define INV(s) ((s) + 8 * (NA + 5 * NP + DECK_SIZE))
#define TERRAIN(a) (SW_DESCR[a] & 7)
#define CAVERN(a) ((SW_DESCR[a] >> 3) & 1)
#define MAGIC(a) ((SW_DESCR[a] >> 4) & 1)
#define MINE(a) ((SW_DESCR[a] >> 5) & 1)
#define HAS_TRIBE(a) ((SW_DESCR[a] >> 6) & 1)
#define AT_EDGE(a) ((SW_DESCR[a] >> 7) & 1)
#define PMOD(x) ((((x) % NP) + NP) % NP)

typedef struct { int8_t* s; int64_t seed; azo_rng* rng; } ctx;

static int split_a(int v) { return ((v % 64) + 64) % 64; }                /* _split_pwr_data :126-130 */
static int split_b(int v) { return (v - split_a(v)) / 64 != 0; }
static int in3(int x, int a, int b, int c) { return x == a || x == b || x == c; }
static int choice_idx(ctx* c, int n) {                                     /* np.random.choice over n entries */
    int k = (int)(azo_rng_u01(c->rng) * (double)n);
    return k >= n ? n - 1 : k;
}
static int64_t pmod64(int64_t a, int64_t m) { int64_t r = a % m; return r < 0 ? r + m : r; }

static int8_t* current_ppl(int8_t* s, int player) { return PPL(s, player, GS(s, player)[4]); }      /* :956-960 */
static uint64_t occupied_by(const int8_t* s, const int8_t* ppl) {                                     /* _are_occupied_by :973-974 */
    uint64_t m = 0;
    for (int a = 0; a < NA; a++) if (T(s, a)[1] == ppl[1]) m |= 1ull << a;
    return m;
}
static int8_t* ppl_owner_of(int8_t* s, int area, int* owner) {                                        /* :962-968 */
    const int t = T(s, area)[1];
    *owner = -1;
    if (t == NOPPL || t == LOST_TRIBE) return NULL;
    for (int p = 0; p < NP; p++)
        for (int id = 0; id < 3; id++)
            if (PPL(s, p, id)[1] == t) { *owner = p; return PPL(s, p, id); }
    return NULL;
}
static int border_of(int area, int terrain) {                                                         /* _is_area_border_of :976-980 */
    for (int a = 0; a < NA; a++) if (((SW_CONN[area] >> a) & 1) && TERRAIN(a) == terrain) return 1;
    return 0;
}
static int minimum_ppl_for_attack(const int8_t* s, int area, const int8_t* cp) {                       /* :982-998 */
    int m = T(s, area)[5] + 2;
    if (cp[1] == TRITON && border_of(area, WATER)) m--;
    if (cp[1] == GIANT && border_of(area, MOUNTAIN)) m--;
    if (cp[2] == COMMANDO) m--;
    if (cp[2] == MOUNTED && (TERRAIN(area) == HILLT || TERRAIN(area) == FARMLAND)) m--;
    if (cp[2] == UNDERWORLD && CAVERN(area)) m--;
    return m > 1 ? m : 1;
}
static int total_number_of_ppl(const int8_t* s, const int8_t* cp, uint64_t terr) {                     /* :1047-1053 */
    int n = cp[0];
    for (int a = 0; a < NA; a++) if ((terr >> a) & 1) n += T(s, a)[0];
    return n;
}
static int limit_added_ppl(const int8_t* s, const int8_t* cp, int addition, int maximum, uint64_t terr) {   /* :1055-1057 */
    const int room = maximum - total_number_of_ppl(s, cp, terr);
    return addition < room ? addition : room;
}
static int surplus_on_board(const int8_t* s, uint64_t terr) {               /* my_dot(max(territories[:,0] - 1, 0), territories_of_player) */
    int n = 0;
    for (int a = 0; a < NA; a++) if (((terr >> a) & 1) && T(s, a)[0] > 1) n += T(s, a)[0] - 1;
    return n;
}
static int ppl_virtually_available(const int8_t* s, int player, const int8_t* cp, int next_status, uint64_t terr) {   /* :1206-1233 */
    const int old = RS((int8_t*)s, player)[4];
    int n = cp[0];
    if (old == PHASE_READY && in3(next_status, PHASE_ABANDON, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE)) n += surplus_on_board(s, terr);
    else if ((old == PHASE_READY || old == PHASE_ABANDON || old == PHASE_CONQUEST || old == PHASE_CONQ_WITH_DICE || old == PHASE_ABANDON_AMAZONS) &&
             next_status == PHASE_REDEPLOY) n += surplus_on_board(s, terr);
    if (cp[1] == AMAZON) {
        if (in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS) && next_status == PHASE_REDEPLOY) { if (cp[3] != 0) n -= cp[3]; }
        else if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next_status == PHASE_CONQUEST) { if (cp[3] == 0) n += 4; }
    } else if (cp[1] == SKELETON) {
        if ((in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) || in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS)) &&
            next_status == PHASE_REDEPLOY)
            if (cp[3] == 0) n += limit_added_ppl(s, cp, cp[3] / 2, MAX_SKELETONS, terr);      /* (cp[3] == 0: adds min(0, room), as written) */
    }
    return n;
}
static int enough_amazons_to_redeploy(const int8_t* s, int player, const int8_t* cp) {                  /* :1434-1440 */
    if (cp[1] == AMAZON && ppl_virtually_available(s, player, cp, PHASE_REDEPLOY, occupied_by(s, cp)) < 0) return 0;
    return 1;
}

static void update_territory_after_win_or_decline(int8_t* s, int8_t* cp, int player, int area) {        /* :1442-1476 */
    int8_t* t = T(s, area);
    if (cp[1] == HALFLING && cp[3] > 0) { t[3] = IMMUNITY; cp[3] = (int8_t)(cp[3] - 1); }
    else t[3] = 0;
    t[5] = (int8_t)(t[0] + t[3] + t[4]);
    if (TERRAIN(area) == MOUNTAIN) t[5] = (int8_t)(t[5] + 1);
    if (t[1] == TROLL || t[1] == -TROLL) t[5] = (int8_t)(t[5] + 1);
    int pts = 1;
    if (MINE(area) && (t[1] == DWARF || t[1] == -DWARF)) pts++;
    if (TERRAIN(area) == FARMLAND && t[1] == HUMAN) pts++;
    if (MAGIC(area) && t[1] == WIZARD) pts++;
    if (TERRAIN(area) == FORESTT && t[2] == FOREST) pts++;
    if (TERRAIN(area) == HILLT && t[2] == HILL) pts++;
    if (TERRAIN(area) == SWAMPT && t[2] == SWAMP) pts++;
    if (t[2] == MERCHANT) pts++;
    if (t[4] > 0 && t[2] == FORTIFIED) pts++;
    t[6] = (int8_t)pts;
    t[7] = (int8_t)player;
}
static void update_round_status(int8_t* s, int8_t* cp, int player) {                                    /* :1478-1508 */
    int8_t* rs = RS(s, player);
    cp[6] = 0; rs[0] = 0; rs[5] = 0; rs[6] = 0;
    for (int a = 0; a < NA; a++) if (T(s, a)[1] == cp[1]) cp[6] = (int8_t)(cp[6] + T(s, a)[6]);
    for (int a = 0; a < NA; a++)
        if (T(s, a)[7] == player) {
            rs[0] = (int8_t)(rs[0] + T(s, a)[0]);
            rs[5] = (int8_t)(rs[5] + T(s, a)[5]);
            if (rs[5] < 0) rs[5] = 127;
        }
    if (cp[1] >= 0) {
        if (cp[1] == ORC) cp[6] = (int8_t)(cp[6] + rs[3]);
        if (cp[2] == PILLAGING) cp[6] = (int8_t)(cp[6] + rs[3]);
        if (cp[2] == ALCHEMIST) cp[6] = (int8_t)(cp[6] + 2);
        if (cp[2] == WEALTHY && cp[4] > 0) cp[6] = (int8_t)(cp[6] + cp[4]);
    }
    rs[6] = (int8_t)(PPL(s, player, 0)[6] + PPL(s, player, 1)[6] + PPL(s, player, 2)[6]);
}
static void empty_area(int8_t* s, int area) {
    int8_t* t = T(s, area);
    t[0] = 0; t[1] = NOPPL; t[2] = NOPOWER; t[3] = 0; t[4] = 0; t[5] = (int8_t)(TERRAIN(area) == MOUNTAIN); t[6] = 0; t[7] = -1;
}
static void give_back_tokens(const int8_t* t, int8_t* owner) {
    if (t[2] == BIVOUACKING || t[2] == FORTIFIED) owner[4] = (int8_t)(owner[4] + t[4]);
    else if (t[2] == HEROIC && t[4] > 0) owner[4] = (int8_t)(owner[4] + 1);
}
static void leave_area(int8_t* s, int area) {                                                           /* :1000-1012 */
    int owner;
    int8_t* lp = ppl_owner_of(s, area, &owner);
    lp[0] = (int8_t)(lp[0] + T(s, area)[0]);
    give_back_tokens(T(s, area), lp);
    empty_area(s, area);
}
static void switch_territory(int8_t* s, int area, int player, int8_t* wp, int nb_attacking) {           /* :1014-1045 */
    int8_t* t = T(s, area);
    const int nb_initial = t[0];
    int loser_id;
    int8_t* lp = ppl_owner_of(s, area, &loser_id);
    if (lp) {
        lp[0] = (int8_t)(lp[0] + t[0] - (t[1] != ELF ? 1 : 0));
        give_back_tokens(t, lp);
        if (wp[2] == DIPLOMAT) wp[4] = (int8_t)(wp[4] | (1 << PMOD(player - loser_id)));
    }
    t[0] = (int8_t)nb_attacking; t[1] = wp[1]; t[2] = wp[2]; t[3] = t[4] = t[5] = t[6] = 0; t[7] = (int8_t)player;
    wp[0] = (int8_t)(wp[0] - nb_attacking);
    if (lp) update_round_status(s, lp, loser_id);
    update_territory_after_win_or_decline(s, wp, player, area);
    if (nb_initial > 0) RS(s, player)[3] = (int8_t)(RS(s, player)[3] + 1);
}
static void gather_current_ppl_but_one(int8_t* s, int8_t* cp) {                                          /* :1059-1067 */
    for (int a = 0; a < NA; a++)
        if (T(s, a)[1] == cp[1]) {
            const int n = T(s, a)[0] - 1;
            if (n > 0) { T(s, a)[0] = (int8_t)(T(s, a)[0] - n); T(s, a)[5] = (int8_t)(T(s, a)[5] - n); cp[0] = (int8_t)(cp[0] + n); }
        }
}

static int roll_dice(ctx* c) {                                                                           /* :417-425, 1193-1200 */
    int8_t* inv = INV(c->s);
    int dice;
    if (c->seed == 0) dice = DICE_VALUES[choice_idx(c, 6)];
    else dice = DICE_VALUES[pmod64(1981 * (c->seed + (int64_t)inv[5]) + 5, 6)];
    inv[5] = (int8_t)(inv[5] + 1);
    return dice;
}
static void switch_status_amazon(int8_t* cp, int old, int next) {                                        /* :1147-1156 */
    if (in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS) && next == PHASE_REDEPLOY) {
        if (cp[3] != 0) { cp[0] = (int8_t)(cp[0] - cp[3]); cp[3] = 0; }
    } else if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next == PHASE_CONQUEST) {
        if (cp[3] == 0) { cp[0] = (int8_t)(cp[0] + 4); cp[3] = 4; }
    }
}
static void switch_status_skeleton(int8_t* s, int player, int8_t* cp, int old, int next) {               /* :1158-1162 */
    if ((in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) || in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS)) &&
        next == PHASE_REDEPLOY && cp[3] == 0) {
        cp[0] = (int8_t)(cp[0] + limit_added_ppl(s, cp, RS(s, player)[3] / 2, MAX_SKELETONS, occupied_by(s, cp)));
        cp[3] = 1;
    }
}
static void switch_status_bivouacking_heroic(int8_t* s, int8_t* cp, int old, int next, int heroic) {     /* :1164-1180 */
    if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next == PHASE_CONQUEST)
        for (int a = 0; a < NA; a++)
            if (T(s, a)[1] == cp[1] && T(s, a)[4] > 0) {
                cp[4] = (int8_t)(cp[4] + (heroic ? 1 : T(s, a)[4]));
                T(s, a)[5] = (int8_t)(T(s, a)[5] - T(s, a)[4]);
                T(s, a)[4] = 0;
            }
}
static void switch_status_diplomat(int8_t* cp, int old, int next) {                                      /* :1182-1189 */
    if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next == PHASE_CONQUEST) cp[4] = 64;
    else if (old != PHASE_WAIT && next == PHASE_WAIT) { if (split_b(cp[4])) cp[4] = 0; }
}
static void switch_status_berserk(ctx* c, int8_t* cp, int next) {                                        /* :1191-1204 */
    if (next == PHASE_READY || next == PHASE_ABANDON || next == PHASE_CHOOSE || next == PHASE_CONQUEST) cp[4] = (int8_t)(roll_dice(c) + 64);
    else cp[4] = 0;
}
static void people_power_switch(ctx* c, int player, int8_t* cp, int old, int next, int ready_variant) {
    int8_t* s = c->s;
    if (cp[1] == AMAZON) switch_status_amazon(cp, old, next);
    else if (cp[1] == SKELETON) switch_status_skeleton(s, player, cp, old, next);
    if (cp[2] == BIVOUACKING) switch_status_bivouacking_heroic(s, cp, old, next, 0);
    else if (cp[2] == HEROIC) switch_status_bivouacking_heroic(s, cp, old, next, 1);
    else if (cp[2] == DIPLOMAT) switch_status_diplomat(cp, old, next);
    else if (cp[2] == BERSERK) {
        if (!ready_variant && next == PHASE_CONQUEST) { /* during an attack the dice is not pre-run yet (:1091-1092) */ }
        else switch_status_berserk(c, cp, next);
    }
}

static void compute_and_update_score(int8_t* s, int player) {                                            /* :1287-1334 */
    int8_t* cp = current_ppl(s, player);
    update_round_status(s, cp, player);
    int score = 0;
    for (int a = 0; a < NA; a++) {
        const int8_t* t = T(s, a);
        if (t[1] == NOPPL || !(t[1] == PPL(s, player, 0)[1] || t[1] == PPL(s, player, 1)[1] || t[1] == PPL(s, player, 2)[1])) continue;
        score++;
        if (MINE(a) && (t[1] == DWARF || t[1] == -DWARF)) score++;
        if (TERRAIN(a) == FARMLAND && t[1] == HUMAN) score++;
        if (MAGIC(a) && t[1] == WIZARD) score++;
        if (TERRAIN(a) == FORESTT && t[2] == FOREST) score++;
        if (TERRAIN(a) == HILLT && t[2] == HILL) score++;
        if (TERRAIN(a) == SWAMPT && t[2] == SWAMP) score++;
        if (t[2] == MERCHANT) score++;
        if (t[4] > 0 && t[2] == FORTIFIED) score++;
    }
    int8_t* ap = PPL(s, player, ACTIVE);
    if (ap[1] == ORC) score += RS(s, player)[3];
    if (ap[2] == PILLAGING) score += RS(s, player)[3];
    if (ap[2] == ALCHEMIST) score += 2;
    if (ap[2] == WEALTHY && ap[4] > 0) { score += ap[4]; ap[4] = 0; }
    const int8_t backup = GS(s, player)[6];
    GS(s, player)[6] = (int8_t)(backup + score);
    if (GS(s, player)[6] < backup) GS(s, player)[6] = 127;
}

static void switch_to_next(ctx* c, int player, int8_t* cp) {                                             /* :1235-1285 */
    int8_t* s = c->s;
    int next_player, next_id;
    if (GS(s, player)[4] != ACTIVE) { next_player = player; next_id = ACTIVE; }
    else {
        next_player = (player + 1) % NP;
        next_id = PPL(s, next_player, DECLINED_SPIRIT)[1] == -GHOUL ? DECLINED_SPIRIT : (PPL(s, next_player, DECLINED)[1] == -GHOUL ? DECLINED : ACTIVE);
        GS(s, player)[3] = (int8_t)(GS(s, player)[3] + 1);
        GS(s, player)[4] = -1;
        RS(s, player)[4] = PHASE_WAIT;
    }
    if (cp[1] == SKELETON || cp[1] == SORCERER) cp[3] = 0;
    if (cp[2] == WEALTHY || cp[2] == BIVOUACKING || cp[2] == HEROIC || cp[2] == DIPLOMAT) { /* kept */ }
    else if (cp[2] == FORTIFIED) cp[4] = (int8_t)split_a(cp[4]);
    else cp[4] = 0;
    RS(s, player)[3] = 0;
    int8_t* np_ = PPL(s, next_player, next_id);
    GS(s, next_player)[4] = (int8_t)next_id;
    RS(s, next_player)[4] = PHASE_READY;
    people_power_switch(c, next_player, np_, PHASE_READY, PHASE_READY, 1);                               /* _prepare_for_ready :1108-1125 */
}
static void prepare_for_new_status(ctx* c, int player, int8_t* cp, int next) {                           /* :1070-1105 */
    int8_t* s = c->s;
    const int old = RS(s, player)[4];
    if (old == PHASE_READY && in3(next, PHASE_ABANDON, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE)) gather_current_ppl_but_one(s, cp);
    else if ((old == PHASE_READY || old == PHASE_CONQUEST || old == PHASE_CONQ_WITH_DICE || old == PHASE_ABANDON_AMAZONS) && next == PHASE_REDEPLOY)
        gather_current_ppl_but_one(s, cp);
    people_power_switch(c, player, cp, old, next, 0);
    if (next == PHASE_STOUT_TO_DECLINE && cp[2] == STOUT) compute_and_update_score(s, player);
    if (next == PHASE_WAIT) {
        if (GS(s, player)[4] == ACTIVE && old != PHASE_STOUT_TO_DECLINE) compute_and_update_score(s, player);
        switch_to_next(c, player, cp);
    }
}

/* ---- valid moves ---- */
static int valid_attack_area(int8_t* s, int player, int area, const int8_t* cp, int avail) {             /* :393-405 */
    if (avail + (cp[2] == BERSERK ? 0 : MAX_DICE) < minimum_ppl_for_attack(s, area, cp)) return 0;
    if (T(s, area)[2] == DIPLOMAT && cp[1] > 0) {
        int loser;
        const int8_t* lp = ppl_owner_of(s, area, &loser);
        if (lp && lp[4] == PMOD(player - loser)) return 0;
    }
    return 1;
}
static void valids_attack(int8_t* s, int player, uint8_t* v) {                                           /* :342-391 */
    const int8_t* cp = current_ppl(s, player);
    const int phase = RS(s, player)[4];
    if (cp[1] == NOPPL) return;
    if (!(phase == PHASE_READY || phase == PHASE_CHOOSE || phase == PHASE_ABANDON || phase == PHASE_CONQUEST)) return;
    const uint64_t terr = occupied_by(s, cp);
    int avail = ppl_virtually_available(s, player, cp, PHASE_CONQUEST, terr);
    if (avail <= 0) return;
    if (cp[2] == BERSERK && split_b(cp[4])) avail += split_a(cp[4]);
    uint64_t neigh = 0;
    int cavern_owned = 0;
    for (int a = 0; a < NA; a++) if ((terr >> a) & 1) { neigh |= SW_CONN[a]; cavern_owned |= CAVERN(a); }
    for (int a = 0; a < NA; a++) {
        if ((terr >> a) & 1) continue;
        if (!(T(s, a)[5] < IMMUNITY)) continue;
        if (cp[2] != SEAFARING && TERRAIN(a) == WATER) continue;
        if (cp[2] != FLYING) {
            if (terr == 0) { if (cp[1] != HALFLING && !AT_EDGE(a)) continue; }
            else {
                int nb = (neigh >> a) & 1;
                if (cp[2] == UNDERWORLD && cavern_owned && CAVERN(a)) nb = 1;
                if (!nb) continue;
            }
        }
        v[a] = (uint8_t)valid_attack_area(s, player, a, cp, avail);
    }
}
static void valids_redeploy(int8_t* s, int player, uint8_t* v) {                                         /* :451-488 (v[31]) */
    const int8_t* cp = current_ppl(s, player);
    const int phase = RS(s, player)[4];
    if (cp[1] == NOPPL) return;
    if (phase == PHASE_WAIT || phase == PHASE_ABANDON_AMAZONS) return;
    const uint64_t terr = occupied_by(s, cp);
    const int nt = __builtin_popcountll(terr);
    if (nt == 0) { if (phase != PHASE_REDEPLOY) v[0] = 1; return; }
    const int avail = ppl_virtually_available(s, player, cp, PHASE_REDEPLOY, terr);
    if (avail == 0) { if (phase != PHASE_REDEPLOY) v[0] = 1; return; }
    if (avail < 0) return;
    int any = 0;
    for (int n = 1; n < MAX_REDEPLOY; n++) { v[n] = (uint8_t)(avail >= n * nt); any |= v[n]; }
    for (int a = 0; a < NA; a++) { v[MAX_REDEPLOY + a] = (uint8_t)((terr >> a) & 1); any |= v[MAX_REDEPLOY + a]; }
    if (!any && phase != PHASE_REDEPLOY) v[0] = 1;
}
static int valid_decline(int8_t* s, int player) {                                                        /* :522-532 */
    const int phase = RS(s, player)[4];
    if (GS(s, player)[4] != ACTIVE || PPL(s, player, ACTIVE)[1] == NOPPL) return 0;
    if (phase != PHASE_READY)
        if (!((phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE || phase == PHASE_REDEPLOY) && PPL(s, player, ACTIVE)[2] == STOUT)) return 0;
    return 1;
}
static void valids_choose(int8_t* s, int player, uint8_t* v) {                                           /* :582-599 */
    if (RS(s, player)[4] != PHASE_READY || GS(s, player)[4] != ACTIVE || PPL(s, player, ACTIVE)[1] != NOPPL) return;
    for (int i = 0; i < DECK_SIZE; i++) v[i] = (uint8_t)(DECK(s, i)[1] != NOPPL && GS(s, player)[6] + 128 >= i);
}
static void valids_abandon(int8_t* s, int player, uint8_t* v) {                                          /* :616-632 */
    const int8_t* cp = current_ppl(s, player);
    const int phase = RS(s, player)[4];
    if (!(phase == PHASE_READY || phase == PHASE_ABANDON || phase == PHASE_ABANDON_AMAZONS))
        if (!(cp[1] == AMAZON && (phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE) &&
              ppl_virtually_available(s, player, cp, PHASE_REDEPLOY, occupied_by(s, cp)) < 0)) return;
    if (cp[1] == NOPPL) return;
    for (int a = 0; a < NA; a++) v[a] = (uint8_t)(T(s, a)[1] == cp[1]);
}
static void valids_special_ppl(int8_t* s, int player, uint8_t* v) {                                      /* :651-701 */
    const int8_t* cp = current_ppl(s, player);
    const int phase = RS(s, player)[4];
    if (cp[1] != SORCERER) return;
    if (!(phase == PHASE_READY || phase == PHASE_CHOOSE || phase == PHASE_ABANDON || phase == PHASE_CONQUEST)) return;
    const uint64_t terr = occupied_by(s, cp);
    if (total_number_of_ppl(s, cp, terr) + 1 > MAX_SORCERERS) return;
    for (int a = 0; a < NA; a++) {
        const int8_t* t = T(s, a);
        if (TERRAIN(a) == WATER && cp[2] != SEAFARING) continue;
        if (t[0] != 1 || t[1] <= 0) continue;
        if (t[1] == cp[1]) continue;
        if (t[3] >= IMMUNITY || t[4] >= IMMUNITY) continue;
        if (cp[2] != FLYING && !(SW_CONN[a] & terr)) continue;
        int loser;
        const int8_t* lp = ppl_owner_of(s, a, &loser);
        if (cp[3] & (1 << PMOD(player - loser))) continue;
        if (lp[2] == BIVOUACKING && t[4] > 0) continue;
        v[a] = 1;
    }
}
static int valid_special_pwr_area(int8_t* s, int player, int area, const int8_t* cp) {                   /* :807-858 */
    const int8_t* t = T(s, area);
    switch (cp[2]) {
    case BIVOUACKING: return t[1] == cp[1];
    case FORTIFIED: case HEROIC: return t[1] == cp[1] && !(t[4] > 0);
    case DIPLOMAT: return !(cp[4] & (1 << PMOD(player - area)));
    case DRAGONMASTER: {
        const uint64_t terr = occupied_by(s, cp);
        if (TERRAIN(area) == WATER || ((terr >> area) & 1)) return 0;
        if (t[3] >= IMMUNITY || t[4] >= IMMUNITY) return 0;
        return (SW_CONN[area] & terr) != 0;
    }
    default: return 0;
    }
}
static void valids_special_pwr(int8_t* s, int player, uint8_t* v) {                                      /* :724-805 */
    const int8_t* cp = current_ppl(s, player);
    const int phase = RS(s, player)[4];
    const int late = phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE || phase == PHASE_REDEPLOY;
    int n = NA;
    if (cp[2] == BIVOUACKING || cp[2] == HEROIC) { if (!late || cp[4] <= 0 || !enough_amazons_to_redeploy(s, player, cp)) return; }
    else if (cp[2] == FORTIFIED) { if (!late || split_a(cp[4]) <= 0 || split_b(cp[4]) || !enough_amazons_to_redeploy(s, player, cp)) return; }
    else if (cp[2] == DIPLOMAT) {
        if (!(phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE) || !enough_amazons_to_redeploy(s, player, cp)) return;
        n = NP;
    } else if (cp[2] == DRAGONMASTER) {
        if (!(phase == PHASE_READY || phase == PHASE_CHOOSE || phase == PHASE_ABANDON || phase == PHASE_CONQUEST) || cp[4] > 0 || cp[0] < 1) return;
    } else return;
    for (int a = 0; a < n; a++) v[a] = (uint8_t)valid_special_pwr_area(s, player, a, cp);
}
static int valid_end_aux(int8_t* s, int player, const int8_t* cp) {                                      /* :929-946 */
    if (RS(s, player)[4] != PHASE_REDEPLOY || cp[1] == NOPPL) return 0;
    if (cp[0] > 0 && occupied_by(s, cp) != 0)
        if (!(cp[1] == AMAZON && cp[0] == cp[3])) return 0;
    return enough_amazons_to_redeploy(s, player, cp);
}

void smallworld_valid_moves(const azo_game* g, const int8_t* cs, int player, uint8_t* out) {             /* :197-208 */
    sw_bind(g);
    int8_t* s = (int8_t*)cs;                      /* (read-only use) */
    memset(out, 0, (size_t)g->A);
    valids_abandon(s, player, out);
    valids_attack(s, player, out + NA);
    valids_special_ppl(s, player, out + 2 * NA);
    valids_special_pwr(s, player, out + 3 * NA);
    valids_redeploy(s, player, out + 4 * NA);
    valids_choose(s, player, out + 5 * NA + MAX_REDEPLOY);
    out[5 * NA + MAX_REDEPLOY + DECK_SIZE] = (uint8_t)valid_decline(s, player);
    out[5 * NA + MAX_REDEPLOY + DECK_SIZE + 1] = (uint8_t)valid_end_aux(s, player, current_ppl(s, player));
}

/* ---- moves ---- */
static void do_end(ctx* c, int player) {                                                                 /* :948-952 */
    int8_t* cp = current_ppl(c->s, player);
    update_round_status(c->s, cp, player);
    prepare_for_new_status(c, player, cp, PHASE_WAIT);
}
static void end_turn_if_possible(ctx* c, int player, int8_t* cp) {                                       /* :1127-1145 */
    if (cp[0] > 0 || cp[2] == STOUT) return;
    if ((cp[2] == BIVOUACKING || cp[2] == FORTIFIED || cp[2] == HEROIC) && cp[4] > 0) return;
    if (!valid_end_aux(c->s, player, cp)) return;
    do_end(c, player);
}
static void do_attack(ctx* c, int player, int area) {                                                    /* :407-449 */
    int8_t* s = c->s;
    int8_t* cp = current_ppl(s, player);
    prepare_for_new_status(c, player, cp, PHASE_CONQUEST);
    const int nb = cp[0], need = minimum_ppl_for_attack(s, area, cp);
    const int use_dice = nb < need;
    int attacking;
    if (cp[2] == BERSERK && split_b(cp[4])) {
        const int dice = split_a(cp[4]);
        if (nb + dice < need) { RS(s, player)[4] = PHASE_CONQ_WITH_DICE; return; }
        attacking = need - dice > 1 ? need - dice : 1;
    } else if (use_dice) {
        const int dice = roll_dice(c);
        if (nb + dice < need) { RS(s, player)[4] = PHASE_CONQ_WITH_DICE; return; }
        attacking = nb;
    } else attacking = need;
    switch_territory(s, area, player, cp, attacking);
    if (cp[2] == BERSERK) switch_status_berserk(c, cp, PHASE_CONQUEST);
    RS(s, player)[4] = (int8_t)(use_dice ? PHASE_CONQ_WITH_DICE : PHASE_CONQUEST);
    update_round_status(s, cp, player);
}
static void do_redeploy(ctx* c, int player, int param) {                                                 /* :490-520 */
    int8_t* s = c->s;
    int8_t* cp = current_ppl(s, player);
    prepare_for_new_status(c, player, cp, PHASE_REDEPLOY);
    RS(s, player)[4] = PHASE_REDEPLOY;
    if (param != 0) {
        if (param < MAX_REDEPLOY) {
            const uint64_t terr = occupied_by(s, cp);
            cp[0] = (int8_t)(cp[0] - param * __builtin_popcountll(terr));
            for (int a = 0; a < NA; a++) if ((terr >> a) & 1) { T(s, a)[0] = (int8_t)(T(s, a)[0] + param); T(s, a)[5] = (int8_t)(T(s, a)[5] + param); }
        } else {
            const int a = param - MAX_REDEPLOY;
            cp[0] = (int8_t)(cp[0] - 1); T(s, a)[0] = (int8_t)(T(s, a)[0] + 1); T(s, a)[5] = (int8_t)(T(s, a)[5] + 1);
        }
    }
    update_round_status(s, cp, player);
    end_turn_if_possible(c, player, cp);
}
static void draw_combo(ctx* c, int slot, uint32_t* avp, uint32_t* avw) {          /* one (people, power) pair for deck slot `slot` */
    int8_t* s = c->s;
    int ppl_ids[16], pwr_ids[24], np_ = 0, nw = 0;
    for (int i = 0; i < 16; i++) if ((*avp >> i) & 1) ppl_ids[np_++] = i;
    for (int i = 0; i < 24; i++) if ((*avw >> i) & 1) pwr_ids[nw++] = i;
    int ppl, pwr;
    if (c->seed == 0) { ppl = ppl_ids[choice_idx(c, np_)]; pwr = pwr_ids[choice_idx(c, nw)]; }
    else {
        const int64_t x = 4594591 * (c->seed + (int64_t)INV(s)[6]);
        ppl = ppl_ids[pmod64(x, np_)]; pwr = pwr_ids[pmod64(x, nw)];
    }
    INV(s)[6] = (int8_t)(INV(s)[6] + 1);
    int8_t* d = DECK(s, slot);
    d[0] = (int8_t)(SW_NB_PEOPLE[ppl] + SW_NB_POWER[pwr]); d[1] = (int8_t)ppl; d[2] = (int8_t)pwr; d[3] = d[4] = d[5] = d[6] = 0; d[7] = -1;
    *avp &= ~(1u << ppl); *avw &= ~(1u << pwr);
}
static void read_avail(const int8_t* s, uint32_t* avp, uint32_t* avw) {            /* my_unpackbits of the bitfields, entry i = bit i */
    *avp = 0; *avw = 0;
    for (int i = 0; i < 16; i++) if (((uint8_t)INV((int8_t*)s)[i >> 3] >> (7 - (i & 7))) & 1) *avp |= 1u << i;
    for (int i = 0; i < 24; i++) if (((uint8_t)INV((int8_t*)s)[2 + (i >> 3)] >> (7 - (i & 7))) & 1) *avw |= 1u << i;
}
static void write_avail(int8_t* s, uint32_t avp, uint32_t avw) {
    for (int b = 0; b < 2; b++) { unsigned v = 0; for (int j = 0; j < 8; j++) if ((avp >> (8 * b + j)) & 1) v |= 128u >> j; INV(s)[b] = (int8_t)(uint8_t)v; }
    for (int b = 0; b < 3; b++) { unsigned v = 0; for (int j = 0; j < 8; j++) if ((avw >> (8 * b + j)) & 1) v |= 128u >> j; INV(s)[2 + b] = (int8_t)(uint8_t)v; }
}
static void update_deck_after_chose(ctx* c, int index) {                                                 /* :1358-1389 */
    int8_t* s = c->s;
    uint32_t avp, avw;
    read_avail(s, &avp, &avw);
    for (int i = index; i < DECK_SIZE - 1; i++) memcpy(DECK(s, i), DECK(s, i + 1), 8);
    for (int i = 0; i < index; i++) DECK(s, i)[6] = (int8_t)(DECK(s, i)[6] + 1);
    if (avp == 0) {
        int8_t* d = DECK(s, DECK_SIZE - 1);
        memset(d, 0, 8); d[7] = -1;
        avp &= ~1u; avw &= ~1u;
    } else draw_combo(c, DECK_SIZE - 1, &avp, &avw);
    write_avail(s, avp, avw);
}
static void update_deck_after_decline(ctx* c) {                                                          /* :1391-1432 */
    int8_t* s = c->s;
    uint32_t avp = 0x7FFEu, avw = 0x1FFFFEu;                      /* peoples 1..14, powers 1..20 */
    for (int i = 0; i < DECK_SIZE; i++) { avp &= ~(1u << DECK(s, i)[1]); avw &= ~(1u << DECK(s, i)[2]); }
    for (int p = 0; p < NP; p++)
        for (int id = 0; id < 3; id++) {
            const int t = PPL(s, p, id)[1], w = PPL(s, p, id)[2];
            if (t != NOPPL) avp &= ~(1u << (t < 0 ? -t : t));
            if (w != NOPOWER) avw &= ~(1u << (w < 0 ? -w : w));
        }
    if (avp != 0)
        for (int i = 0; i < DECK_SIZE; i++)
            if (DECK(s, i)[0] == NOPPL) draw_combo(c, i, &avp, &avw);
    write_avail(s, avp, avw);
}
static void do_decline(ctx* c, int player) {                                                             /* :534-580 */
    int8_t* s = c->s;
    int8_t* cp = PPL(s, player, ACTIVE);
    if (cp[2] == STOUT) { prepare_for_new_status(c, player, cp, PHASE_STOUT_TO_DECLINE); RS(s, player)[4] = PHASE_STOUT_TO_DECLINE; }
    const int did = cp[2] == SPIRIT ? DECLINED_SPIRIT : DECLINED;
    int8_t* dp = PPL(s, player, did);
    if (dp[1] != NOPPL) {
        for (int a = 0; a < NA; a++) if (T(s, a)[1] == dp[1]) empty_area(s, a);
        memset(dp, 0, 7);
        update_deck_after_decline(c);
    }
    if (cp[1] == GHOUL) dp[0] = cp[0];
    else gather_current_ppl_but_one(s, cp);
    dp[1] = cp[1];
    memset(cp, 0, 7);
    for (int a = 0; a < NA; a++)
        if (T(s, a)[1] == dp[1]) {
            int8_t* t = T(s, a);
            const int8_t b2 = t[2], b4 = t[4];
            t[1] = (int8_t)(-dp[1]);
            t[2] = t[3] = t[4] = t[5] = t[6] = 0;
            if (b2 == FORTIFIED) t[4] = b4;
            update_territory_after_win_or_decline(s, cp, player, a);
        }
    dp[1] = (int8_t)(-dp[1]); dp[2] = (int8_t)(-dp[2]);
    update_round_status(s, dp, player);
    prepare_for_new_status(c, player, cp, PHASE_WAIT);
    RS(s, player)[4] = PHASE_WAIT;
}
static void do_choose_ppl(ctx* c, int player, int index) {                                               /* :601-614 */
    int8_t* s = c->s;
    int8_t* cp = PPL(s, player, ACTIVE);
    cp[0] = DECK(s, index)[0]; cp[1] = DECK(s, index)[1]; cp[2] = DECK(s, index)[2];
    cp[3] = SW_TOKENS[cp[1]]; cp[4] = SW_TOKENS_PWR[cp[2]]; cp[5] = cp[6] = 0;
    GS(s, player)[6] = (int8_t)(GS(s, player)[6] + DECK(s, index)[6] - index);
    prepare_for_new_status(c, player, cp, PHASE_CHOOSE);
    RS(s, player)[4] = PHASE_CHOOSE;
    update_deck_after_chose(c, index);
}
static void do_abandon(ctx* c, int player, int area) {                                                   /* :634-649 */
    int8_t* s = c->s;
    int8_t* cp = current_ppl(s, player);
    const int phase = RS(s, player)[4];
    leave_area(s, area);
    int next = PHASE_ABANDON;
    if (phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE || phase == PHASE_ABANDON_AMAZONS)
        next = ppl_virtually_available(s, player, cp, PHASE_REDEPLOY, occupied_by(s, cp)) >= 0 ? PHASE_REDEPLOY : PHASE_ABANDON_AMAZONS;
    prepare_for_new_status(c, player, cp, next);
    RS(s, player)[4] = (int8_t)next;
    update_round_status(s, cp, player);
}
static void do_special_ppl(ctx* c, int player, int area) {                                               /* :703-722 (sorcerer) */
    int8_t* s = c->s;
    int8_t* cp = current_ppl(s, player);
    int loser;
    int8_t* lp = ppl_owner_of(s, area, &loser);
    prepare_for_new_status(c, player, cp, PHASE_CONQUEST);
    int8_t* t = T(s, area);
    t[0] = 1; t[1] = SORCERER; t[2] = cp[2]; t[3] = t[4] = t[5] = t[6] = 0; t[7] = (int8_t)player;
    cp[3] = (int8_t)(cp[3] | (1 << PMOD(player - loser)));
    RS(s, player)[4] = PHASE_CONQUEST;
    RS(s, player)[3] = (int8_t)(RS(s, player)[3] + 1);
    update_territory_after_win_or_decline(s, lp, loser, area);
    update_territory_after_win_or_decline(s, cp, player, area);
    update_round_status(s, cp, player);
}
static void do_special_pwr(ctx* c, int player, int area) {                                               /* :860-923 */
    int8_t* s = c->s;
    int8_t* cp = current_ppl(s, player);
    int8_t* t = T(s, area);
    switch (cp[2]) {
    case BIVOUACKING:
        t[4] = (int8_t)(t[4] + 1); t[5] = (int8_t)(t[5] + 1); cp[4] = (int8_t)(cp[4] - 1);
        break;
    case FORTIFIED:
        t[4] = (int8_t)(t[4] + 1); t[5] = (int8_t)(t[5] + 1); t[6] = (int8_t)(t[6] + 1);
        cp[4] = (int8_t)((cp[4] - 1) | 64);
        break;
    case HEROIC:
        t[5] = (int8_t)(t[5] + (IMMUNITY - t[4])); t[4] = IMMUNITY; cp[4] = (int8_t)(cp[4] - 1);
        break;
    case DIPLOMAT:
        cp[4] = (int8_t)area;
        prepare_for_new_status(c, player, cp, PHASE_REDEPLOY);
        RS(s, player)[4] = PHASE_REDEPLOY;
        return;
    case DRAGONMASTER:
        for (int a = 0; a < NA; a++)
            if (T(s, a)[1] == cp[1] && T(s, a)[4] != 0) { T(s, a)[5] = (int8_t)(T(s, a)[5] - T(s, a)[4]); T(s, a)[4] = 0; }
        prepare_for_new_status(c, player, cp, PHASE_CONQUEST);
        switch_territory(s, area, player, cp, 1);
        t[5] = (int8_t)(t[5] + IMMUNITY); t[4] = IMMUNITY;
        cp[4] = 1;
        RS(s, player)[4] = PHASE_CONQUEST;
        update_round_status(s, cp, player);
        return;
    default: return;
    }
    prepare_for_new_status(c, player, cp, PHASE_REDEPLOY);
    RS(s, player)[4] = PHASE_REDEPLOY;
    update_round_status(s, cp, player);
}

int smallworld_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t seed, azo_rng* rng) {   /* :210-240 */
    sw_bind(g);
    ctx c = {s, seed, rng};
    if (move < NA) do_abandon(&c, player, move);
    else if (move < 2 * NA) do_attack(&c, player, move - NA);
    else if (move < 3 * NA) do_special_ppl(&c, player, move - 2 * NA);
    else if (move < 4 * NA) do_special_pwr(&c, player, move - 3 * NA);
    else if (move < 5 * NA + MAX_REDEPLOY) do_redeploy(&c, player, move - 4 * NA);
    else if (move < 5 * NA + MAX_REDEPLOY + DECK_SIZE) do_choose_ppl(&c, player, move - 5 * NA - MAX_REDEPLOY);
    else if (move == 5 * NA + MAX_REDEPLOY + DECK_SIZE) do_decline(&c, player);
    else do_end(&c, player);
    return GS(s, player)[4] >= 0 ? player : (player + 1) % NP;
}

int smallworld_get_round(const azo_game* g, const int8_t* s) {                                           /* :245-246 */
    sw_bind(g);
    int r = GS((int8_t*)s, 0)[3];
    for (int p = 1; p < NP; p++) if (GS((int8_t*)s, p)[3] < r) r = GS((int8_t*)s, p)[3];
    return r;
}
int smallworld_get_score(const azo_game* g, const int8_t* s, int p) { sw_bind(g); return GS((int8_t*)s, p)[6] + 128; }

void smallworld_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) {            /* :248-257 */
    (void)next_player;
    for (int p = 0; p < NP; p++) out[p] = 0.f;
    if (smallworld_get_round(g, s) <= NB_ROUNDS) return;
    int best = -1000, cnt = 0;
    for (int p = 0; p < NP; p++) if (GS((int8_t*)s, p)[6] > best) best = GS((int8_t*)s, p)[6];
    for (int p = 0; p < NP; p++) cnt += GS((int8_t*)s, p)[6] == best;
    for (int p = 0; p < NP; p++) out[p] = GS((int8_t*)s, p)[6] == best ? (cnt > 1 ? 0.01f : 1.f) : -1.f;
}

void smallworld_swap_players(const azo_game* g, int8_t* s, int k) {                                      /* :260-279 */
    sw_bind(g);
    k = PMOD(k);
    if (k == 0) return;
    for (int a = 0; a < NA; a++) if (T(s, a)[7] >= 0) T(s, a)[7] = (int8_t)PMOD(T(s, a)[7] - k);
    int8_t tmp[3 * MAXNP * 8];
    memcpy(tmp, RS(s, 0), NP * 8);
    for (int p = 0; p < NP; p++) memcpy(RS(s, p), tmp + 8 * ((p + k) % NP), 7);
    memcpy(tmp, GS(s, 0), NP * 8);
    for (int p = 0; p < NP; p++) memcpy(GS(s, p), tmp + 8 * ((p + k) % NP), 7);
    memcpy(tmp, PPL(s, 0, 0), 3 * NP * 8);
    for (int p = 0; p < NP; p++)
        for (int id = 0; id < 3; id++) memcpy(PPL(s, p, id), tmp + 8 * (3 * ((p + k) % NP) + id), 7);
}

void smallworld_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                                 /* :150-174, 1339-1356 */
    sw_bind(g);
    memset(s, 0, (size_t)g->S);
    ctx c = {s, 0, rng};
    for (int a = 0; a < NA; a++) {
        empty_area(s, a);
        if (HAS_TRIBE(a)) { T(s, a)[0] = SW_NB_PEOPLE[15]; T(s, a)[1] = LOST_TRIBE; T(s, a)[5] = (int8_t)(SW_NB_PEOPLE[15] + (TERRAIN(a) == MOUNTAIN)); }
    }
    uint32_t avp = 0x7FFEu, avw = 0x1FFFFEu;
    for (int i = 0; i < DECK_SIZE; i++) {
        int ids[24], n = 0;
        for (int j = 0; j < 15; j++) if ((avp >> j) & 1) ids[n++] = j;
        const int ppl = ids[choice_idx(&c, n)];
        n = 0;
        for (int j = 0; j < 21; j++) if ((avw >> j) & 1) ids[n++] = j;
        const int pwr = ids[choice_idx(&c, n)];
        int8_t* d = DECK(s, i);
        d[0] = (int8_t)(SW_NB_PEOPLE[ppl] + SW_NB_POWER[pwr]); d[1] = (int8_t)ppl; d[2] = (int8_t)pwr; d[7] = -1;
        avp &= ~(1u << ppl); avw &= ~(1u << pwr);
    }
    write_avail(s, avp, avw);
    for (int p = 0; p < NP; p++) {
        RS(s, p)[4] = (int8_t)(p == 0 ? PHASE_READY : PHASE_WAIT); RS(s, p)[7] = (int8_t)p;
        GS(s, p)[4] = (int8_t)(p == 0 ? ACTIVE : -1); GS(s, p)[6] = (int8_t)(SCORE_INIT - 128); GS(s, p)[7] = (int8_t)p;
        for (int id = 0; id < 3; id++) PPL(s, p, id)[7] = (int8_t)p;
        GS(s, p)[3] = 1;                                                                   /* _update_round */
    }
}

/* get_symmetries :281-299: the identity + two copies whose scores are shifted by a random offset in [-127 - min, 127 - max) */
int smallworld_symmetries_rng(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op, uint8_t* ov,
                              int max_sym, azo_rng* rng) {
    const int S = g->S, A = g->A;
    sw_bind(g);
    int k = 0;
    for (int f = 0; f < 3; f++) {
        int off = 0;
        if (f > 0) {
            int mn = 1000, mx = -1000;
            for (int p = 0; p < NP; p++) { const int v = GS((int8_t*)s, p)[6]; if (v < mn) mn = v; if (v > mx) mx = v; }
            const int lo = -127 - mn, hi = 127 - mx;
            if (lo >= hi) continue;
            int d = (int)(azo_rng_u01(rng) * (double)(hi - lo));
            if (d > hi - lo - 1) d = hi - lo - 1;
            off = lo + d;
        }
        if (k >= max_sym) continue;
        int8_t* st = os + (size_t)k * S;
        memcpy(st, s, (size_t)S); memcpy(op + (size_t)k * A, pi, sizeof(float) * (size_t)A); memcpy(ov + (size_t)k * A, valids, (size_t)A);
        for (int p = 0; p < NP; p++) GS(st, p)[6] = (int8_t)(GS(st, p)[6] + off);
        k++;
    }
    return k;
}
int smallworld_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op, uint8_t* ov,
                          int max_sym) {
    azo_rng r;
    memset(&r, 0, sizeof(r));
    return smallworld_symmetries_rng(g, s, pi, valids, os, op, ov, max_sym, &r);
}
