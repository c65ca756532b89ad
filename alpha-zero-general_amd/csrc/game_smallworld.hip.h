// game_smallworld.hip.h -- Smallworld env step on the device plugin interface: smallworld/SmallworldLogicNumba.py
// (Board :141-1508) for NUMBER_PLAYERS = 2 (the shipped constant), 3 and 4 with their maps (SmallworldMaps_<n>pl.py: 23 / 30 / 39 areas,
// 10 / 10 / 9 rounds).
//
// State int8 [NB_AREAS + 5 n + 7][8] (:11-60; indices below for 2 players): rows 0..22 territories {nb people, people type (negative in decline, -15 lost tribe), power, defence due
// to the people, defence due to the power, total defence, points if scored now, owner (-1 none)}; rows 23 + 3 p + id peoples of player p
// (id 0 declined-spirit, 1 declined, 2 active) {nb in hand, type, power, people data, power data, -, points, player}; rows 29..34 visible
// deck {nb, type, power, -, -, -, coins, -1}; rows 35 + p round status {people on the map, -, -, #NETWDT, phase, total defence, points
// preview, player}; rows 37 + p game status {-, -, -, round, id of the people playing (-1 not my turn), -, score - 128, player}; row 39
// invisible deck {2 bytes of available peoples, 3 bytes of available powers (MSB first), dice count, deck draw count, -}.
// 131 actions (:76-88): abandon 0..22, attack 23..45, people action 46..68, power action 69..91, redeploy 92..122 (92 skip, 93..99 n on
// each area, 100..122 one on area), choose 123..128, decline 129, end 130.
//
// The env step is a function of (state, action, random_seed) for random_seed != 0 -- dice and deck draws are functions of the seed and of
// the two counters in the invisible deck (:421-424, 1378-1381) -- so edges are memoised like Splendor's; real moves and init
// (random_seed == 0) draw np.random.choice(x) = x[floor(u len)] from the tree's counter stream.  get_symmetries (:281-299) adds two random
// score offsets (np.random.randint): BUILT symmetric forms on the caller's stream (k_env_symmetries_built).
//
// The rules (14 peoples x 20 powers, 9 phases) are branchy scalar code on 320 bytes: make_move / init run on lane 0 over the LDS state;
// the valid-move mask is one action per lane (every lane recomputes the few shared quantities of its action class).
#pragma once
#include "azg_common.hip.h"
#include "smallworld_tables.h"

namespace azg {

template <int NPL>
struct SmallworldDev {
    static constexpr int NP = NPL, NA = NPL == 2 ? SW_NA_2 : (NPL == 3 ? SW_NA_3 : SW_NA_4);
    static constexpr int NB_ROUNDS = NPL == 2 ? SW_ROUNDS_2 : (NPL == 3 ? SW_ROUNDS_3 : SW_ROUNDS_4);
    static constexpr int P = NPL;
    static constexpr int ROWS = NA + 5 * NPL + 7, COLS = 8;
    static constexpr int S = ROWS * 8;
    static constexpr int SP = RoundUp16<S>::value;
    static constexpr int A = 5 * NA + 16;
    static constexpr int AW = (A + 63) / 64;
    static constexpr bool STOCHASTIC = false;
    static constexpr bool RANDOM_SYM = true;      // symmetric forms are built by lane 0 (k_env_symmetries_built) and draw randomness
    static constexpr bool SYM_DEDUP = false;
    enum { FORESTT, FARMLAND, HILLT, SWAMPT, MOUNTAIN, WATER };
    enum { DECK_SIZE = 6, SCORE_INIT = 5, IMMUNITY = 20, MAX_REDEPLOY = 8, MAX_DICE = 3 };
    enum { DECLINED_SPIRIT = 0, DECLINED = 1, ACTIVE = 2 };
    enum { PHASE_READY = 1, PHASE_CHOOSE, PHASE_ABANDON, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS, PHASE_REDEPLOY,
           PHASE_STOUT_TO_DECLINE, PHASE_WAIT };
    enum { NOPPL = 0, AMAZON, DWARF, ELF, GHOUL, GIANT, HALFLING, HUMAN, ORC, RATMAN, SKELETON, SORCERER, TRITON, TROLL, WIZARD, LOST_TRIBE = -15 };
    enum { MAX_SKELETONS = 20, MAX_SORCERERS = 18 };
    enum { NOPOWER = 0, ALCHEMIST, BERSERK, BIVOUACKING, COMMANDO, DIPLOMAT, DRAGONMASTER, FLYING, FOREST, FORTIFIED, HEROIC, HILL, MERCHANT,
           MOUNTED, PILLAGING, SEAFARING, SPIRIT, STOUT, SWAMP, UNDERWORLD, WEALTHY };
    struct Ctx { int8_t* s; long long seed; Rng& rng; };

    template <class T_> __device__ static __forceinline__ T_* T(T_* s, int a) { return s + 8 * a; }
    template <class T_> __device__ static __forceinline__ T_* PPL(T_* s, int p, int id) { return s + 8 * (NA + 3 * p + id); }
    template <class T_> __device__ static __forceinline__ T_* DECK(T_* s, int i) { return s + 8 * (NA + 3 * NP + i); }
    template <class T_> __device__ static __forceinline__ T_* RS(T_* s, int p) { return s + 8 * (NA + 3 * NP + DECK_SIZE + p); }
    template <class T_> __device__ static __forceinline__ T_* GS(T_* s, int p) { return s + 8 * (NA + 4 * NP + DECK_SIZE + p); }
    template <class T_> __device__ static __forceinline__ T_* INV(T_* s) { return s + 8 * (NA + 5 * NP + DECK_SIZE); }
    __device__ static __forceinline__ int SW_DESCR_(int a) { if constexpr (NPL == 2) return SW_DESCR_2[a]; else if constexpr (NPL == 3) return SW_DESCR_3[a]; else return SW_DESCR_4[a]; }
    __device__ static __forceinline__ uint64_t SW_CONN_(int a) { if constexpr (NPL == 2) return SW_CONN_2[a]; else if constexpr (NPL == 3) return SW_CONN_3[a]; else return SW_CONN_4[a]; }
    __device__ static __forceinline__ int TERRAIN(int a) { return SW_DESCR_(a) & 7; }
    __device__ static __forceinline__ int CAVERN(int a) { return (SW_DESCR_(a) >> 3) & 1; }
    __device__ static __forceinline__ int MAGIC(int a) { return (SW_DESCR_(a) >> 4) & 1; }
    __device__ static __forceinline__ int MINE(int a) { return (SW_DESCR_(a) >> 5) & 1; }
    __device__ static __forceinline__ int HAS_TRIBE(int a) { return (SW_DESCR_(a) >> 6) & 1; }
    __device__ static __forceinline__ int AT_EDGE(int a) { return (SW_DESCR_(a) >> 7) & 1; }
    __device__ static __forceinline__ int PMOD(int x) { return ((x % NP) + NP) % NP; }
    __device__ static __forceinline__ int dice_value(int i) { return i < 3 ? 0 : i - 2; }        // DICE_VALUES = {0, 0, 0, 1, 2, 3}

    __device__ static int split_a(int v) { return ((v % 64) + 64) % 64; }                /* _split_pwr_data :126-130 */
    __device__ static int split_b(int v) { return (v - split_a(v)) / 64 != 0; }
    __device__ static int in3(int x, int a, int b, int c) { return x == a || x == b || x == c; }
    __device__ static int choice_idx(Ctx& c, int n) {                                     /* np.random.choice over n entries */
        int k = (int)(c.rng.u01() * (double)n);
        return k >= n ? n - 1 : k;
    }
    __device__ static long long pmod64(long long a, long long m) { long long r = a % m; return r < 0 ? r + m : r; }

    __device__ static int8_t* current_ppl(int8_t* s, int player) { return PPL(s, player, GS(s, player)[4]); }      /* :956-960 */
    __device__ static uint64_t occupied_by(const int8_t* s, const int8_t* ppl) {                                     /* _are_occupied_by :973-974 */
        uint64_t m = 0;
        const int8_t t = ppl[1];
#pragma unroll
        for (int a = 0; a < NA; a++) if (T(s, a)[1] == t) m |= 1ull << a;
        return m;
    }
    __device__ static int8_t* ppl_owner_of(int8_t* s, int area, int* owner) {                                        /* :962-968 */
        const int t = T(s, area)[1];
        *owner = -1;
        if (t == NOPPL || t == LOST_TRIBE) return nullptr;
        for (int p = 0; p < NP; p++)
            for (int id = 0; id < 3; id++)
                if (PPL(s, p, id)[1] == t) { *owner = p; return PPL(s, p, id); }
        return nullptr;
    }
    __device__ static int border_of(int area, int terrain) {                                                         /* _is_area_border_of :976-980 */
        for (int a = 0; a < NA; a++) if (((SW_CONN_(area) >> a) & 1) && TERRAIN(a) == terrain) return 1;
        return 0;
    }
    __device__ static int minimum_ppl_for_attack(const int8_t* s, int area, const int8_t* cp) {                       /* :982-998 */
        int m = T(s, area)[5] + 2;
        if (cp[1] == TRITON && border_of(area, WATER)) m--;
        if (cp[1] == GIANT && border_of(area, MOUNTAIN)) m--;
        if (cp[2] == COMMANDO) m--;
        if (cp[2] == MOUNTED && (TERRAIN(area) == HILLT || TERRAIN(area) == FARMLAND)) m--;
        if (cp[2] == UNDERWORLD && CAVERN(area)) m--;
        return m > 1 ? m : 1;
    }
    __device__ static int total_number_of_ppl(const int8_t* s, const int8_t* cp, uint64_t terr) {                     /* :1047-1053 */
        int n = cp[0];
#pragma unroll
        for (int a = 0; a < NA; a++) { const int v = T(s, a)[0]; if ((terr >> a) & 1) n += v; }
        return n;
    }
    __device__ static int limit_added_ppl(const int8_t* s, const int8_t* cp, int addition, int maximum, uint64_t terr) {   /* :1055-1057 */
        const int room = maximum - total_number_of_ppl(s, cp, terr);
        return addition < room ? addition : room;
    }
    __device__ static int surplus_on_board(const int8_t* s, uint64_t terr) {               /* my_dot(max(territories[:,0] - 1, 0), territories_of_player) */
        int n = 0;
#pragma unroll
        for (int a = 0; a < NA; a++) { const int v = T(s, a)[0]; if (((terr >> a) & 1) && v > 1) n += v - 1; }
        return n;
    }
    __device__ static int ppl_virtually_available(const int8_t* s, int player, const int8_t* cp, int next_status, uint64_t terr) {   /* :1206-1233 */
        const int old = RS(s, player)[4];
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
    // the same with the two sums over the people's territories already known (the valid-move scan forms them in one pass over the areas)
    __device__ static int ppl_virtually_available_u(int old, const int8_t* cp, int next_status, int surplus, int total_ppl) {
        int n = cp[0];
        if (old == PHASE_READY && in3(next_status, PHASE_ABANDON, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE)) n += surplus;
        else if ((old == PHASE_READY || old == PHASE_ABANDON || old == PHASE_CONQUEST || old == PHASE_CONQ_WITH_DICE || old == PHASE_ABANDON_AMAZONS) &&
                 next_status == PHASE_REDEPLOY) n += surplus;
        if (cp[1] == AMAZON) {
            if (in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS) && next_status == PHASE_REDEPLOY) { if (cp[3] != 0) n -= cp[3]; }
            else if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next_status == PHASE_CONQUEST) { if (cp[3] == 0) n += 4; }
        } else if (cp[1] == SKELETON) {
            if ((in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) || in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS)) &&
                next_status == PHASE_REDEPLOY)
                if (cp[3] == 0) { const int add = cp[3] / 2, room = MAX_SKELETONS - total_ppl; n += add < room ? add : room; }
        }
        return n;
    }
    __device__ static int enough_amazons_to_redeploy(const int8_t* s, int player, const int8_t* cp) {                  /* :1434-1440 */
        if (cp[1] == AMAZON && ppl_virtually_available(s, player, cp, PHASE_REDEPLOY, occupied_by(s, cp)) < 0) return 0;
        return 1;
    }

    __device__ static void update_territory_after_win_or_decline(int8_t* s, int8_t* cp, int player, int area) {        /* :1442-1476 */
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
    // (The loops over the areas only LOAD: with a store to the state inside, every iteration had to wait for the one before -- the
    // compiler cannot tell cp[6] / rs[5] from the territory rows -- and a move runs three of these.  The int8 wrap-around of every
    // running sum and the saturation of the defence total are applied step by step, as written.)
    __device__ static void update_round_status(int8_t* s, int8_t* cp, int player) {                                    /* :1478-1508 */
        int8_t* rs = RS(s, player);
        const int8_t cp1 = cp[1];
        int8_t c6 = 0, r0 = 0, r5 = 0;
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const uint64_t row = *(const uint64_t*)T(s, a);
            const int8_t t0 = (int8_t)row, t1 = (int8_t)(row >> 8), t5 = (int8_t)(row >> 40), t6 = (int8_t)(row >> 48), t7 = (int8_t)(row >> 56);
            if (t1 == cp1) c6 = (int8_t)(c6 + t6);
            if (t7 == player) {
                r0 = (int8_t)(r0 + t0);
                r5 = (int8_t)(r5 + t5);
                if (r5 < 0) r5 = 127;
            }
        }
        rs[0] = r0; rs[5] = r5;
        if (cp1 >= 0) {
            if (cp1 == ORC) c6 = (int8_t)(c6 + rs[3]);
            if (cp[2] == PILLAGING) c6 = (int8_t)(c6 + rs[3]);
            if (cp[2] == ALCHEMIST) c6 = (int8_t)(c6 + 2);
            if (cp[2] == WEALTHY && cp[4] > 0) c6 = (int8_t)(c6 + cp[4]);
        }
        cp[6] = c6;
        rs[6] = (int8_t)(PPL(s, player, 0)[6] + PPL(s, player, 1)[6] + PPL(s, player, 2)[6]);
    }
    __device__ static void empty_area(int8_t* s, int area) {
        int8_t* t = T(s, area);
        t[0] = 0; t[1] = NOPPL; t[2] = NOPOWER; t[3] = 0; t[4] = 0; t[5] = (int8_t)(TERRAIN(area) == MOUNTAIN); t[6] = 0; t[7] = -1;
    }
    __device__ static void give_back_tokens(const int8_t* t, int8_t* owner) {
        if (t[2] == BIVOUACKING || t[2] == FORTIFIED) owner[4] = (int8_t)(owner[4] + t[4]);
        else if (t[2] == HEROIC && t[4] > 0) owner[4] = (int8_t)(owner[4] + 1);
    }
    __device__ static void leave_area(int8_t* s, int area) {                                                           /* :1000-1012 */
        int owner;
        int8_t* lp = ppl_owner_of(s, area, &owner);
        lp[0] = (int8_t)(lp[0] + T(s, area)[0]);
        give_back_tokens(T(s, area), lp);
        empty_area(s, area);
    }
    __device__ static void switch_territory(int8_t* s, int area, int player, int8_t* wp, int nb_attacking) {           /* :1014-1045 */
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
    __device__ static void gather_current_ppl_but_one(int8_t* s, int8_t* cp) {                                          /* :1059-1067 */
        const int8_t t = cp[1];
        int8_t c0 = cp[0];
        uint64_t rows[NA];
#pragma unroll
        for (int a = 0; a < NA; a++) rows[a] = *(const uint64_t*)T(s, a);
#pragma unroll
        for (int a = 0; a < NA; a++)
            if ((int8_t)(rows[a] >> 8) == t) {
                const int8_t t0 = (int8_t)rows[a], t5 = (int8_t)(rows[a] >> 40);
                const int n = t0 - 1;
                if (n > 0) { T(s, a)[0] = (int8_t)(t0 - n); T(s, a)[5] = (int8_t)(t5 - n); c0 = (int8_t)(c0 + n); }
            }
        cp[0] = c0;
    }

    __device__ static int roll_dice(Ctx& c) {                                                                           /* :417-425, 1193-1200 */
        int8_t* inv = INV(c.s);
        int dice;
        if (c.seed == 0) dice = dice_value(choice_idx(c, 6));
        else dice = dice_value((int)pmod64(1981 * (c.seed + (long long)inv[5]) + 5, 6));
        inv[5] = (int8_t)(inv[5] + 1);
        return dice;
    }
    __device__ static void switch_status_amazon(int8_t* cp, int old, int next) {                                        /* :1147-1156 */
        if (in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS) && next == PHASE_REDEPLOY) {
            if (cp[3] != 0) { cp[0] = (int8_t)(cp[0] - cp[3]); cp[3] = 0; }
        } else if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next == PHASE_CONQUEST) {
            if (cp[3] == 0) { cp[0] = (int8_t)(cp[0] + 4); cp[3] = 4; }
        }
    }
    __device__ static void switch_status_skeleton(int8_t* s, int player, int8_t* cp, int old, int next) {               /* :1158-1162 */
        if ((in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) || in3(old, PHASE_CONQUEST, PHASE_CONQ_WITH_DICE, PHASE_ABANDON_AMAZONS)) &&
            next == PHASE_REDEPLOY && cp[3] == 0) {
            cp[0] = (int8_t)(cp[0] + limit_added_ppl(s, cp, RS(s, player)[3] / 2, MAX_SKELETONS, occupied_by(s, cp)));
            cp[3] = 1;
        }
    }
    __device__ static void switch_status_bivouacking_heroic(int8_t* s, int8_t* cp, int old, int next, int heroic) {     /* :1164-1180 */
        if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next == PHASE_CONQUEST)
            for (int a = 0; a < NA; a++)
                if (T(s, a)[1] == cp[1] && T(s, a)[4] > 0) {
                    cp[4] = (int8_t)(cp[4] + (heroic ? 1 : T(s, a)[4]));
                    T(s, a)[5] = (int8_t)(T(s, a)[5] - T(s, a)[4]);
                    T(s, a)[4] = 0;
                }
    }
    __device__ static void switch_status_diplomat(int8_t* cp, int old, int next) {                                      /* :1182-1189 */
        if (in3(old, PHASE_READY, PHASE_CHOOSE, PHASE_ABANDON) && next == PHASE_CONQUEST) cp[4] = 64;
        else if (old != PHASE_WAIT && next == PHASE_WAIT) { if (split_b(cp[4])) cp[4] = 0; }
    }
    __device__ static void switch_status_berserk(Ctx& c, int8_t* cp, int next) {                                        /* :1191-1204 */
        if (next == PHASE_READY || next == PHASE_ABANDON || next == PHASE_CHOOSE || next == PHASE_CONQUEST) cp[4] = (int8_t)(roll_dice(c) + 64);
        else cp[4] = 0;
    }
    __device__ static void people_power_switch(Ctx& c, int player, int8_t* cp, int old, int next, int ready_variant) {
        int8_t* s = c.s;
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

    __device__ static void compute_and_update_score(int8_t* s, int player) {                                            /* :1287-1334 */
        int8_t* cp = current_ppl(s, player);
        update_round_status(s, cp, player);
        int score = 0;
        const int8_t p0 = PPL(s, player, 0)[1], p1 = PPL(s, player, 1)[1], p2 = PPL(s, player, 2)[1];
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const uint64_t row = *(const uint64_t*)T(s, a);
            const int8_t t1 = (int8_t)(row >> 8), t2 = (int8_t)(row >> 16), t4 = (int8_t)(row >> 32);
            if (t1 == NOPPL || !(t1 == p0 || t1 == p1 || t1 == p2)) continue;
            score++;
            if (MINE(a) && (t1 == DWARF || t1 == -DWARF)) score++;
            if (TERRAIN(a) == FARMLAND && t1 == HUMAN) score++;
            if (MAGIC(a) && t1 == WIZARD) score++;
            if (TERRAIN(a) == FORESTT && t2 == FOREST) score++;
            if (TERRAIN(a) == HILLT && t2 == HILL) score++;
            if (TERRAIN(a) == SWAMPT && t2 == SWAMP) score++;
            if (t2 == MERCHANT) score++;
            if (t4 > 0 && t2 == FORTIFIED) score++;
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

    __device__ static void switch_to_next(Ctx& c, int player, int8_t* cp) {                                             /* :1235-1285 */
        int8_t* s = c.s;
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
    __device__ static void prepare_for_new_status(Ctx& c, int player, int8_t* cp, int next) {                           /* :1070-1105 */
        int8_t* s = c.s;
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

    __device__ static int valid_attack_area(int8_t* s, int player, int area, const int8_t* cp, int avail) {             /* :393-405 */
        if (avail + (cp[2] == BERSERK ? 0 : MAX_DICE) < minimum_ppl_for_attack(s, area, cp)) return 0;
        if (T(s, area)[2] == DIPLOMAT && cp[1] > 0) {
            int loser;
            const int8_t* lp = ppl_owner_of(s, area, &loser);
            if (lp && lp[4] == PMOD(player - loser)) return 0;
        }
        return 1;
    }
    __device__ static int valid_special_pwr_area(int8_t* s, int player, int area, const int8_t* cp) {                   /* :807-858 */
        const int8_t* t = T(s, area);
        switch (cp[2]) {
        case BIVOUACKING: return t[1] == cp[1];
        case FORTIFIED: case HEROIC: return t[1] == cp[1] && !(t[4] > 0);
        case DIPLOMAT: return !(cp[4] & (1 << PMOD(player - area)));
        case DRAGONMASTER: {
            const uint64_t terr = occupied_by(s, cp);
            if (TERRAIN(area) == WATER || ((terr >> area) & 1)) return 0;
            if (t[3] >= IMMUNITY || t[4] >= IMMUNITY) return 0;
            return (SW_CONN_(area) & terr) != 0;
        }
        default: return 0;
        }
    }

    __device__ static int valid_end_aux(int8_t* s, int player, const int8_t* cp) {                                      /* :929-946 */
        if (RS(s, player)[4] != PHASE_REDEPLOY || cp[1] == NOPPL) return 0;
        if (cp[0] > 0 && occupied_by(s, cp) != 0)
            if (!(cp[1] == AMAZON && cp[0] == cp[3])) return 0;
        return enough_amazons_to_redeploy(s, player, cp);
    }

    /* ---- moves ---- */
    __device__ static void do_end(Ctx& c, int player) {                                                                 /* :948-952 */
        int8_t* cp = current_ppl(c.s, player);
        update_round_status(c.s, cp, player);
        prepare_for_new_status(c, player, cp, PHASE_WAIT);
    }
    __device__ static void end_turn_if_possible(Ctx& c, int player, int8_t* cp) {                                       /* :1127-1145 */
        if (cp[0] > 0 || cp[2] == STOUT) return;
        if ((cp[2] == BIVOUACKING || cp[2] == FORTIFIED || cp[2] == HEROIC) && cp[4] > 0) return;
        if (!valid_end_aux(c.s, player, cp)) return;
        do_end(c, player);
    }
    __device__ static void do_attack(Ctx& c, int player, int area) {                                                    /* :407-449 */
        int8_t* s = c.s;
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
    __device__ static void do_redeploy(Ctx& c, int player, int param) {                                                 /* :490-520 */
        int8_t* s = c.s;
        int8_t* cp = current_ppl(s, player);
        prepare_for_new_status(c, player, cp, PHASE_REDEPLOY);
        RS(s, player)[4] = PHASE_REDEPLOY;
        if (param != 0) {
            if (param < MAX_REDEPLOY) {
                const uint64_t terr = occupied_by(s, cp);
                cp[0] = (int8_t)(cp[0] - param * __popcll((unsigned long long)terr));
                for (int a = 0; a < NA; a++) if ((terr >> a) & 1) { T(s, a)[0] = (int8_t)(T(s, a)[0] + param); T(s, a)[5] = (int8_t)(T(s, a)[5] + param); }
            } else {
                const int a = param - MAX_REDEPLOY;
                cp[0] = (int8_t)(cp[0] - 1); T(s, a)[0] = (int8_t)(T(s, a)[0] + 1); T(s, a)[5] = (int8_t)(T(s, a)[5] + 1);
            }
        }
        update_round_status(s, cp, player);
        end_turn_if_possible(c, player, cp);
    }
    __device__ static void draw_combo(Ctx& c, int slot, uint32_t* avp, uint32_t* avw) {          /* one (people, power) pair for deck slot `slot` */
        int8_t* s = c.s;
        int ppl_ids[16], pwr_ids[24], np_ = 0, nw = 0;
        for (int i = 0; i < 16; i++) if ((*avp >> i) & 1) ppl_ids[np_++] = i;
        for (int i = 0; i < 24; i++) if ((*avw >> i) & 1) pwr_ids[nw++] = i;
        int ppl, pwr;
        if (c.seed == 0) { ppl = ppl_ids[choice_idx(c, np_)]; pwr = pwr_ids[choice_idx(c, nw)]; }
        else {
            const long long x = 4594591 * (c.seed + (long long)INV(s)[6]);
            ppl = ppl_ids[pmod64(x, np_)]; pwr = pwr_ids[pmod64(x, nw)];
        }
        INV(s)[6] = (int8_t)(INV(s)[6] + 1);
        int8_t* d = DECK(s, slot);
        d[0] = (int8_t)(SW_NB_PEOPLE[ppl] + SW_NB_POWER[pwr]); d[1] = (int8_t)ppl; d[2] = (int8_t)pwr; d[3] = d[4] = d[5] = d[6] = 0; d[7] = -1;
        *avp &= ~(1u << ppl); *avw &= ~(1u << pwr);
    }
    __device__ static void read_avail(const int8_t* s, uint32_t* avp, uint32_t* avw) {            /* my_unpackbits of the bitfields, entry i = bit i */
        *avp = 0; *avw = 0;
        for (int i = 0; i < 16; i++) if (((uint8_t)INV((int8_t*)s)[i >> 3] >> (7 - (i & 7))) & 1) *avp |= 1u << i;
        for (int i = 0; i < 24; i++) if (((uint8_t)INV((int8_t*)s)[2 + (i >> 3)] >> (7 - (i & 7))) & 1) *avw |= 1u << i;
    }
    __device__ static void write_avail(int8_t* s, uint32_t avp, uint32_t avw) {
        for (int b = 0; b < 2; b++) { unsigned v = 0; for (int j = 0; j < 8; j++) if ((avp >> (8 * b + j)) & 1) v |= 128u >> j; INV(s)[b] = (int8_t)(uint8_t)v; }
        for (int b = 0; b < 3; b++) { unsigned v = 0; for (int j = 0; j < 8; j++) if ((avw >> (8 * b + j)) & 1) v |= 128u >> j; INV(s)[2 + b] = (int8_t)(uint8_t)v; }
    }
    __device__ static void update_deck_after_chose(Ctx& c, int index) {                                                 /* :1358-1389 */
        int8_t* s = c.s;
        uint32_t avp, avw;
        read_avail(s, &avp, &avw);
        for (int i = index; i < DECK_SIZE - 1; i++) for (int z = 0; z < 8; z++) DECK(s, i)[z] = DECK(s, i + 1)[z];
        for (int i = 0; i < index; i++) DECK(s, i)[6] = (int8_t)(DECK(s, i)[6] + 1);
        if (avp == 0) {
            int8_t* d = DECK(s, DECK_SIZE - 1);
            for (int z = 0; z < 7; z++) d[z] = 0; d[7] = -1;
            avp &= ~1u; avw &= ~1u;
        } else draw_combo(c, DECK_SIZE - 1, &avp, &avw);
        write_avail(s, avp, avw);
    }
    __device__ static void update_deck_after_decline(Ctx& c) {                                                          /* :1391-1432 */
        int8_t* s = c.s;
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
    __device__ static void do_decline(Ctx& c, int player) {                                                             /* :534-580 */
        int8_t* s = c.s;
        int8_t* cp = PPL(s, player, ACTIVE);
        if (cp[2] == STOUT) { prepare_for_new_status(c, player, cp, PHASE_STOUT_TO_DECLINE); RS(s, player)[4] = PHASE_STOUT_TO_DECLINE; }
        const int did = cp[2] == SPIRIT ? DECLINED_SPIRIT : DECLINED;
        int8_t* dp = PPL(s, player, did);
        if (dp[1] != NOPPL) {
            for (int a = 0; a < NA; a++) if (T(s, a)[1] == dp[1]) empty_area(s, a);
            for (int z = 0; z < 7; z++) dp[z] = 0;
            update_deck_after_decline(c);
        }
        if (cp[1] == GHOUL) dp[0] = cp[0];
        else gather_current_ppl_but_one(s, cp);
        dp[1] = cp[1];
        for (int z = 0; z < 7; z++) cp[z] = 0;
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
    __device__ static void do_choose_ppl(Ctx& c, int player, int index) {                                               /* :601-614 */
        int8_t* s = c.s;
        int8_t* cp = PPL(s, player, ACTIVE);
        cp[0] = DECK(s, index)[0]; cp[1] = DECK(s, index)[1]; cp[2] = DECK(s, index)[2];
        cp[3] = SW_TOKENS[cp[1]]; cp[4] = SW_TOKENS_PWR[cp[2]]; cp[5] = cp[6] = 0;
        GS(s, player)[6] = (int8_t)(GS(s, player)[6] + DECK(s, index)[6] - index);
        prepare_for_new_status(c, player, cp, PHASE_CHOOSE);
        RS(s, player)[4] = PHASE_CHOOSE;
        update_deck_after_chose(c, index);
    }
    __device__ static void do_abandon(Ctx& c, int player, int area) {                                                   /* :634-649 */
        int8_t* s = c.s;
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
    __device__ static void do_special_ppl(Ctx& c, int player, int area) {                                               /* :703-722 (sorcerer) */
        int8_t* s = c.s;
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
    __device__ static void do_special_pwr(Ctx& c, int player, int area) {                                               /* :860-923 */
        int8_t* s = c.s;
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


    // ---- Board.valid_moves :197-208.  What every action of a class shares -- the current people, its territories, the people it could field
    // for a conquest / a redeployment, the neighbourhood of its territories -- is computed ONCE by the wave (uniform code on broadcast
    // reads); a lane then decides its own action from its own territory row.  (Round 4 had every lane recompute the shared quantities of
    // its class, and the wave runs all seven classes one after the other.) ----
    struct VShared {
        const int8_t* cp;
        int phase, avail_conq, avail_redeploy, total_ppl, nt, cavern_owned;
        bool early, enough_amazons;
        uint64_t terr, neigh, water_mask, mountain_mask;
    };
    __device__ static uint64_t terrain_mask(int terrain) {
        uint64_t m = 0;
#pragma unroll
        for (int a = 0; a < NA; a++) if (TERRAIN(a) == terrain) m |= 1ull << a;
        return m;
    }
    __device__ static VShared valid_shared(int8_t* s, int player) {
        VShared u;
        u.cp = current_ppl(s, player);
        u.phase = RS(s, player)[4];
        u.early = u.phase == PHASE_READY || u.phase == PHASE_CHOOSE || u.phase == PHASE_ABANDON || u.phase == PHASE_CONQUEST;
        // one pass over the areas (one 8-byte row each): the people's territories, their surplus and their head count, their neighbourhood
        u.terr = 0; u.neigh = 0; u.cavern_owned = 0;
        int surplus = 0, on_board = 0;
        const int8_t cp1 = u.cp[1];
#pragma unroll
        for (int a = 0; a < NA; a++) {
            const uint64_t row = *(const uint64_t*)T(s, a);
            const int t0 = (int8_t)row;
            if ((int8_t)(row >> 8) == cp1) {
                u.terr |= 1ull << a; u.neigh |= SW_CONN_(a); u.cavern_owned |= CAVERN(a);
                on_board += t0;
                if (t0 > 1) surplus += t0 - 1;
            }
        }
        u.nt = __popcll((unsigned long long)u.terr);
        u.total_ppl = u.cp[0] + on_board;
        u.avail_conq = ppl_virtually_available_u(u.phase, u.cp, PHASE_CONQUEST, surplus, u.total_ppl);
        u.avail_redeploy = ppl_virtually_available_u(u.phase, u.cp, PHASE_REDEPLOY, surplus, u.total_ppl);
        u.enough_amazons = !(u.cp[1] == AMAZON && u.avail_redeploy < 0);                   // _enough_amazons_to_redeploy :1434-1440
        u.water_mask = terrain_mask(WATER); u.mountain_mask = terrain_mask(MOUNTAIN);
        return u;
    }
    __device__ static int minimum_ppl_for_attack_u(const int8_t* s, int area, const VShared& u) {                      /* :982-998 */
        const int8_t* cp = u.cp;
        int m = T(s, area)[5] + 2;
        if (cp[1] == TRITON && (SW_CONN_(area) & u.water_mask)) m--;
        if (cp[1] == GIANT && (SW_CONN_(area) & u.mountain_mask)) m--;
        if (cp[2] == COMMANDO) m--;
        if (cp[2] == MOUNTED && (TERRAIN(area) == HILLT || TERRAIN(area) == FARMLAND)) m--;
        if (cp[2] == UNDERWORLD && CAVERN(area)) m--;
        return m > 1 ? m : 1;
    }
    __device__ static bool valid_action(const int8_t* cs, int a, int player, const VShared& u) {
        int8_t* s = (int8_t*)cs;                                             // (read-only use)
        const int8_t* cp = u.cp;
        const int phase = u.phase;
        const uint64_t terr = u.terr;
        if (a < NA) {                                                        // _valids_abandon :616-632
            if (!(phase == PHASE_READY || phase == PHASE_ABANDON || phase == PHASE_ABANDON_AMAZONS))
                if (!(cp[1] == AMAZON && (phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE) && u.avail_redeploy < 0)) return false;
            return cp[1] != NOPPL && ((terr >> a) & 1);
        }
        if (a < 2 * NA) {                                                    // _valids_attack :342-391
            const int area = a - NA;
            if (cp[1] == NOPPL || !u.early) return false;
            int avail = u.avail_conq;
            if (avail <= 0) return false;
            if (cp[2] == BERSERK && split_b(cp[4])) avail += split_a(cp[4]);
            if ((terr >> area) & 1) return false;
            if (!(T(s, area)[5] < IMMUNITY)) return false;
            if (cp[2] != SEAFARING && TERRAIN(area) == WATER) return false;
            if (cp[2] != FLYING) {
                if (terr == 0) { if (cp[1] != HALFLING && !AT_EDGE(area)) return false; }
                else {
                    int nb = (u.neigh >> area) & 1;
                    if (cp[2] == UNDERWORLD && u.cavern_owned && CAVERN(area)) nb = 1;
                    if (!nb) return false;
                }
            }
            if (avail + (cp[2] == BERSERK ? 0 : MAX_DICE) < minimum_ppl_for_attack_u(s, area, u)) return false;      // _valid_attack_area :393-405
            if (T(s, area)[2] == DIPLOMAT && cp[1] > 0) {
                int loser;
                const int8_t* lp = ppl_owner_of(s, area, &loser);
                if (lp && lp[4] == PMOD(player - loser)) return false;
            }
            return true;
        }
        if (a < 3 * NA) {                                                    // _valids_special_actionppl :651-701 (sorcerer)
            const int area = a - 2 * NA;
            if (cp[1] != SORCERER || !u.early) return false;
            if (u.total_ppl + 1 > MAX_SORCERERS) return false;
            const int8_t* t = T(s, area);
            if (TERRAIN(area) == WATER && cp[2] != SEAFARING) return false;
            if (t[0] != 1 || t[1] <= 0 || t[1] == cp[1]) return false;
            if (t[3] >= IMMUNITY || t[4] >= IMMUNITY) return false;
            if (cp[2] != FLYING && !(SW_CONN_(area) & terr)) return false;
            int loser;
            const int8_t* lp = ppl_owner_of(s, area, &loser);
            if (cp[3] & (1 << PMOD(player - loser))) return false;
            return !(lp[2] == BIVOUACKING && t[4] > 0);
        }
        if (a < 4 * NA) {                                                    // _valids_special_actionpwr :724-805
            const int area = a - 3 * NA;
            const bool late = phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE || phase == PHASE_REDEPLOY;
            int n = NA;
            if (cp[2] == BIVOUACKING || cp[2] == HEROIC) { if (!late || cp[4] <= 0 || !u.enough_amazons) return false; }
            else if (cp[2] == FORTIFIED) { if (!late || split_a(cp[4]) <= 0 || split_b(cp[4]) || !u.enough_amazons) return false; }
            else if (cp[2] == DIPLOMAT) {
                if (!(phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE) || !u.enough_amazons) return false;
                n = NP;
            } else if (cp[2] == DRAGONMASTER) { if (!u.early || cp[4] > 0 || cp[0] < 1) return false; }
            else return false;
            if (area >= n) return false;
            const int8_t* t = T(s, area);                                    // _valid_special_pwr_area :807-858
            switch (cp[2]) {
            case BIVOUACKING: return t[1] == cp[1];
            case FORTIFIED: case HEROIC: return t[1] == cp[1] && !(t[4] > 0);
            case DIPLOMAT: return !(cp[4] & (1 << PMOD(player - area)));
            default:                                                         // DRAGONMASTER
                if (TERRAIN(area) == WATER || ((terr >> area) & 1)) return false;
                if (t[3] >= IMMUNITY || t[4] >= IMMUNITY) return false;
                return (SW_CONN_(area) & terr) != 0;
            }
        }
        if (a < 5 * NA + MAX_REDEPLOY) {                                     // _valids_redeploy :451-488
            const int i = a - 4 * NA;
            if (cp[1] == NOPPL || phase == PHASE_WAIT || phase == PHASE_ABANDON_AMAZONS) return false;
            const int nt = u.nt;
            const int avail = nt == 0 ? 0 : u.avail_redeploy;
            if (nt == 0 || avail == 0) return i == 0 && phase != PHASE_REDEPLOY;     // nothing to deploy: only "skip", once
            if (avail < 0 || i == 0) return false;                                    // (a territory is always a valid target: no skip)
            return i < MAX_REDEPLOY ? avail >= i * nt : ((terr >> (i - MAX_REDEPLOY)) & 1) != 0;
        }
        if (a < 5 * NA + MAX_REDEPLOY + DECK_SIZE) {                         // _valids_choose_ppl :582-599
            const int i = a - 5 * NA - MAX_REDEPLOY;
            if (phase != PHASE_READY || GS(s, player)[4] != ACTIVE || PPL(s, player, ACTIVE)[1] != NOPPL) return false;
            return DECK(s, i)[1] != NOPPL && GS(s, player)[6] + 128 >= i;
        }
        if (a == 5 * NA + MAX_REDEPLOY + DECK_SIZE) {                        // _valid_decline :522-532
            if (GS(s, player)[4] != ACTIVE || PPL(s, player, ACTIVE)[1] == NOPPL) return false;
            if (phase != PHASE_READY)
                if (!((phase == PHASE_CONQUEST || phase == PHASE_CONQ_WITH_DICE || phase == PHASE_REDEPLOY) && PPL(s, player, ACTIVE)[2] == STOUT)) return false;
            return true;
        }
        if (phase != PHASE_REDEPLOY || cp[1] == NOPPL) return false;         // _valid_end :925-946
        if (cp[0] > 0 && terr != 0)
            if (!(cp[1] == AMAZON && cp[0] == cp[3])) return false;
        return u.enough_amazons;
    }
    __device__ static __attribute__((noinline)) void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
        const VShared u = valid_shared((int8_t*)st, player);
#pragma unroll 1
        for (int k = 0; k < AW; k++) {
            const int a = k * 64 + l;
            const uint64_t m = __ballot(a < A && valid_action(st, a < A ? a : 0, player, u));
            if (l == 0) mask_lds[k] = m;
        }
    }

    __device__ static __forceinline__ int get_score(const int8_t* st, int p) { return GS(st, p)[6] + 128; }
    __device__ static __forceinline__ int get_round(const int8_t* st) {        // :245-246
        int r = GS(st, 0)[3];
#pragma unroll
        for (int p = 1; p < NP; p++) r = GS(st, p)[3] < r ? GS(st, p)[3] : r;
        return r;
    }
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return get_round(st) & 255; }     // the round only grows
    __device__ static __forceinline__ bool move_uses_seed(int) { return true; }     // dice / deck draws depend on the state: be conservative

    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        return lane0_make_move<SmallworldDev<NPL>>(st, move, player, seed, rng);
    }
    // Board.make_move :210-240 -- lane 0 only
    __device__ static __attribute__((noinline)) int make_move(int8_t* s, int move, int player, long long seed, Rng& rng) {
        Ctx c{s, seed, rng};
        if (move < NA) do_abandon(c, player, move);
        else if (move < 2 * NA) do_attack(c, player, move - NA);
        else if (move < 3 * NA) do_special_ppl(c, player, move - 2 * NA);
        else if (move < 4 * NA) do_special_pwr(c, player, move - 3 * NA);
        else if (move < 5 * NA + MAX_REDEPLOY) do_redeploy(c, player, move - 4 * NA);
        else if (move < 5 * NA + MAX_REDEPLOY + DECK_SIZE) do_choose_ppl(c, player, move - 5 * NA - MAX_REDEPLOY);
        else if (move == 5 * NA + MAX_REDEPLOY + DECK_SIZE) do_decline(c, player);
        else do_end(c, player);
        return GS(s, player)[4] >= 0 ? player : (player + 1) % NP;
    }

    // Board.check_end_game :248-257 (uniform)
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        if (get_round(st) <= NB_ROUNDS) {
#pragma unroll
            for (int p = 0; p < NP; p++) out[p] = 0.f;
            return false;
        }
        int best = -1000, cnt = 0;
#pragma unroll
        for (int p = 0; p < NP; p++) best = GS(st, p)[6] > best ? GS(st, p)[6] : best;
#pragma unroll
        for (int p = 0; p < NP; p++) cnt += GS(st, p)[6] == best;
#pragma unroll
        for (int p = 0; p < NP; p++) out[p] = GS(st, p)[6] == best ? (cnt > 1 ? 0.01f : 1.f) : -1.f;
        return true;
    }

    // Board.swap_players :260-279: owners shift by k, the status rows and the people rows of player p come from player (p + k) mod n
    // except their last column (the player id)
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        k = PMOD(k);
        if (k == 0) return;
        for (int i = lane_id(); i < S; i += 64) tmp[i] = st[i];
        wave_sync();
        for (int i = lane_id(); i < S; i += 64) {
            const int r = i >> 3, z = i & 7;
            int8_t v = tmp[i];
            if (r < NA) { if (z == 7 && v >= 0) v = (int8_t)PMOD(v - k); }
            else if (z < 7) {
                if (r < NA + 3 * NP) { const int p = (r - NA) / 3, id = (r - NA) - 3 * p; v = tmp[(NA + 3 * ((p + k) % NP) + id) * 8 + z]; }
                else if (r >= NA + 3 * NP + DECK_SIZE && r < NA + 4 * NP + DECK_SIZE) v = tmp[(NA + 3 * NP + DECK_SIZE + (r - (NA + 3 * NP + DECK_SIZE) + k) % NP) * 8 + z];
                else if (r >= NA + 4 * NP + DECK_SIZE && r < NA + 5 * NP + DECK_SIZE) v = tmp[(NA + 4 * NP + DECK_SIZE + (r - (NA + 4 * NP + DECK_SIZE) + k) % NP) * 8 + z];
            }
            st[i] = v;
        }
        wave_sync();
    }

    // init_game :150-174, _init_deck :1339-1356 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* s, Rng& rng) {
        Ctx c{s, 0, rng};
        for (int a = 0; a < NA; a++) {
            empty_area(s, a);
            if (HAS_TRIBE(a)) { T(s, a)[0] = SW_NB_PEOPLE[15]; T(s, a)[1] = LOST_TRIBE; T(s, a)[5] = (int8_t)(SW_NB_PEOPLE[15] + (TERRAIN(a) == MOUNTAIN)); }
        }
        uint32_t avp = 0x7FFEu, avw = 0x1FFFFEu;
        for (int i = 0; i < DECK_SIZE; i++) {
            int k = choice_idx(c, __popc(avp)), ppl = 0, pwr = 0;
            for (int j = 0; j < 15; j++) if ((avp >> j) & 1) { if (k == 0) { ppl = j; break; } k--; }
            k = choice_idx(c, __popc(avw));
            for (int j = 0; j < 21; j++) if ((avw >> j) & 1) { if (k == 0) { pwr = j; break; } k--; }
            int8_t* d = DECK(s, i);
            d[0] = (int8_t)(SW_NB_PEOPLE[ppl] + SW_NB_POWER[pwr]); d[1] = (int8_t)ppl; d[2] = (int8_t)pwr; d[7] = -1;
            avp &= ~(1u << ppl); avw &= ~(1u << pwr);
        }
        write_avail(s, avp, avw);
        for (int p = 0; p < NP; p++) {
            RS(s, p)[4] = (int8_t)(p == 0 ? PHASE_READY : PHASE_WAIT); RS(s, p)[7] = (int8_t)p;
            GS(s, p)[4] = (int8_t)(p == 0 ? ACTIVE : -1); GS(s, p)[6] = (int8_t)(SCORE_INIT - 128); GS(s, p)[7] = (int8_t)p;
            for (int id = 0; id < 3; id++) PPL(s, p, id)[7] = (int8_t)p;
            GS(s, p)[3] = 1;                                                   // _update_round
        }
    }

    // get_symmetries :281-299: the identity + two copies whose scores are shifted by a random offset in [-127 - min, 127 - max)
    static constexpr int NSYM_CAND = 3;
    __device__ static bool sym_build(const int8_t* st, int c, int8_t* cand, int16_t* act_src, Rng& rng, const uint8_t* valids) {
        (void)valids;
        for (int a = 0; a < A; a++) act_src[a] = (int16_t)a;
        if (c == 0) return true;
        int mn = 1000, mx = -1000;
        for (int p = 0; p < NP; p++) { const int v = GS(st, p)[6]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        const int lo = -127 - mn, hi = 127 - mx;
        if (lo >= hi) return false;
        int d = (int)(rng.u01() * (double)(hi - lo));
        d = d > hi - lo - 1 ? hi - lo - 1 : d;
        for (int p = 0; p < NP; p++) GS(cand, p)[6] = (int8_t)(GS(cand, p)[6] + lo + d);
        return true;
    }
    // (the deterministic per-byte interface is unused: k_env_symmetries takes the built path)
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int c) { return c == 0; }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int, int i) { return st[i]; }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int, int a) { return a; }
};

}  // namespace azg
