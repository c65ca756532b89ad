/* ORACLE (test infrastructure, not product).  Splendor 2-4 players.
 * Scalar C restatement of splendor/SplendorLogicNumba.py `Board` (reference lines cited per function).
 * State = int8[(32+10n+n*n)][7], byte-identical to the reference's board.tobytes(). */
#include <string.h>
#include "azg_oracle.h"
#include "splendor_tables.h"

#define COLS 7
#define GOLD 5
#define PTS 6

typedef struct {
    int n, nn;                 /* players, nobles in play (n+1)                SplendorLogicNumba.py:141-146 */
    int r_nobles, r_gems, r_pnobles, r_pcards, r_res;   /* first row of each block  :212-219 */
} lay_t;

static lay_t lay(int n) {
    lay_t L;
    L.n = n; L.nn = n + 1;
    L.r_nobles = 31;
    L.r_gems = 32 + n;
    L.r_pnobles = 32 + 2 * n;
    L.r_pcards = 32 + 3 * n + n * n;
    L.r_res = 32 + 4 * n + n * n;
    return L;
}

#define ROW(st, r) ((st) + (r) * COLS)

static int sum5(const int8_t* row) { return row[0] + row[1] + row[2] + row[3] + row[4]; }
static int sum7(const int8_t* row) { return sum5(row) + row[5] + row[6]; }

/* int8 element-wise a-b-c evaluated with int8 wrap like the NumPy/Numba int8 arrays (:350,363) */
static int missing_colors(const int8_t* cost, const int8_t* gems, const int8_t* cards) {
    int s = 0;
    for (int c = 0; c < 5; c++) {
        int8_t d = (int8_t)((int8_t)(cost[c] - gems[c]) - cards[c]);
        if (d > 0) s += d;
    }
    return s;
}

void splendor_valid_moves(const azo_game* g, const int8_t* st, int player, uint8_t* out) {
    lay_t L = lay(g->P);
    const int8_t* bank = ROW(st, 0);
    const int8_t* gems = ROW(st, L.r_gems + player);
    const int8_t* cards = ROW(st, L.r_pcards + player);
    memset(out, 0, 81);
    /* _valid_buy :359-368 */
    for (int i = 0; i < 12; i++) {
        const int8_t* cost = ROW(st, 1 + 2 * i);
        out[i] = (missing_colors(cost, gems, cards) <= gems[GOLD]) && (sum5(cost) != 0);
    }
    /* _valid_reserve :375-380  (3rd reserve slot's GAIN row must be empty) */
    int empty_slot = sum5(ROW(st, L.r_res + 6 * player + 5)) == 0;
    for (int i = 0; i < 12; i++) out[12 + i] = empty_slot && (sum5(ROW(st, 1 + 2 * i)) != 0);
    for (int t = 0; t < 3; t++) out[24 + t] = empty_slot && (sum5(ROW(st, 25 + 2 * t)) != 0);
    /* _valid_buy_reserve :402-412 */
    for (int i = 0; i < 3; i++) {
        const int8_t* cost = ROW(st, L.r_res + 6 * player + 2 * i);
        out[27 + i] = (missing_colors(cost, gems, cards) <= gems[GOLD]) && (sum5(cost) != 0);
    }
    /* _valid_get_gems :422-427, _valid_get_gems_identical :429-434 */
    int total = sum7(gems);
    for (int i = 0; i < 25; i++) {
        int ok = 1, k = 0;
        for (int c = 0; c < 5; c++) {
            if ((int8_t)(bank[c] - SPL_GEMS3[i][c]) < 0) ok = 0;
            k += SPL_GEMS3[i][c];
        }
        out[30 + i] = ok && (total + k <= 10);
    }
    for (int c = 0; c < 5; c++) out[55 + c] = (bank[c] >= 4) && (total + 2 <= 10);
    /* _valid_give_gems :446-449, _valid_give_gems_identical :451-453 */
    for (int i = 0; i < 15; i++) {
        int ok = 1;
        for (int c = 0; c < 5; c++)
            if ((int8_t)(gems[c] - SPL_GEMS2[i][c]) < 0) ok = 0;
        out[60 + i] = ok;
    }
    for (int c = 0; c < 5; c++) out[75 + c] = gems[c] >= 2;
    out[80] = 1; /* :187 */
}

/* _get_deck_card :306-336.  Returns 0 if the deck is empty, else writes the 14 card bytes. */
static int get_deck_card(int8_t* st, int tier, int64_t random_seed, azo_rng* rng, int8_t* card14) {
    int8_t* cnt = ROW(st, 25 + 2 * tier);
    int8_t* bits = ROW(st, 26 + 2 * tier);
    int total = sum5(cnt);
    if (total == 0) return 0;
    int color = 0, card_index = 0;
    if (random_seed == 0) {
        /* :311-315 two-stage: colour ~ remaining counts, then card uniform among the set bits;
           my_random_choice = searchsorted(cumsum(p), u, side='right') :100-103 */
        double u = azo_rng_u01(rng), acc = 0.0;
        int k = 0;
        for (k = 0; k < 5; k++) {
            acc += (double)cnt[k] / (double)total;
            if (acc > u) break;
        }
        if (k >= 5) { for (k = 4; k > 0 && cnt[k] == 0; k--) {} }
        color = k;
        uint8_t b = (uint8_t)bits[color];
        int nb = __builtin_popcount(b);
        double u2 = azo_rng_u01(rng);
        acc = 0.0;
        int idx = -1, last = 0;
        for (int i = 0; i < 8; i++) {
            int set = (b >> (7 - i)) & 1;
            if (set) last = i;
            acc += (double)set / (double)nb;
            if (acc > u2) { idx = i; break; }
        }
        card_index = idx < 0 ? last : idx;
    } else {
        /* :316-323 seeded draw: candidates colour-major, MSB-first; LCG index */
        int cand_c[40], cand_i[40], n = 0;
        int64_t seedv = 0, pw = 1;
        for (int c = 0; c < 5; c++) {
            uint8_t b = (uint8_t)bits[c];
            for (int i = 0; i < 8; i++)
                if ((b >> (7 - i)) & 1) { cand_c[n] = c; cand_i[n] = i; n++; }
            seedv += (int64_t)b * pw;
            pw *= 32;
        }
        int64_t x = 4594591LL * (random_seed + seedv);
        int64_t r = x % n;
        if (r < 0) r += n; /* Python floor-mod */
        color = cand_c[r];
        card_index = cand_i[r];
    }
    uint8_t b = (uint8_t)bits[color];
    b &= (uint8_t)~(0x80u >> card_index);
    bits[color] = (int8_t)b;     /* int8 wrap :327 */
    cnt[color] -= 1;
    memcpy(card14, SPL_CARDS[tier][color][card_index], 14);
    return 1;
}

/* _fill_new_card :338-342 */
static void fill_new_card(int8_t* st, int tier, int index, int64_t seed, azo_rng* rng) {
    int8_t* dst = ROW(st, 1 + 8 * tier + 2 * index);
    int8_t card[14];
    memset(dst, 0, 14);
    if (get_deck_card(st, tier, seed, rng, card)) memcpy(dst, card, 14);
}

/* _give_nobles_if_earned :465-470 */
static void give_nobles(int8_t* st, const lay_t* L, int player) {
    const int8_t* cards = ROW(st, L->r_pcards + player);
    for (int i = 0; i < L->nn; i++) {
        int8_t* noble = ROW(st, L->r_nobles + i);
        if (sum5(noble) <= 0) continue;
        int ok = 1;
        for (int c = 0; c < 5; c++) if (cards[c] < noble[c]) ok = 0;
        if (ok) {
            memcpy(ROW(st, L->r_pnobles + L->nn * player + i), noble, COLS);
            memset(noble, 0, COLS);
        }
    }
}

/* _buy_card :344-357 */
static void buy_card(int8_t* st, const lay_t* L, const int8_t* card0, const int8_t* card1, int player) {
    int8_t* bank = ROW(st, 0);
    int8_t* gems = ROW(st, L->r_gems + player);
    int8_t* cards = ROW(st, L->r_pcards + player);
    int8_t cost[5], gain[7];
    memcpy(cost, card0, 5);
    memcpy(gain, card1, 7);
    int miss = missing_colors(cost, gems, cards);
    for (int c = 0; c < 5; c++) {
        int8_t need = (int8_t)(cost[c] - cards[c]);
        if (need < 0) need = 0;
        int8_t paid = need < gems[c] ? need : gems[c];
        gems[c] -= paid;
        bank[c] += paid;
    }
    gems[GOLD] = (int8_t)(gems[GOLD] - miss);
    bank[GOLD] = (int8_t)(bank[GOLD] + miss);
    for (int c = 0; c < COLS; c++) cards[c] += gain[c];
    give_nobles(st, L, player);
}

int splendor_make_move(const azo_game* g, int8_t* st, int move, int player, int64_t seed, azo_rng* rng) {
    lay_t L = lay(g->P);
    int8_t* bank = ROW(st, 0);
    int8_t* gems = ROW(st, L.r_gems + player);
    if (move < 12) {                                   /* _buy :370-373 */
        int tier = move / 4, index = move % 4;
        buy_card(st, &L, ROW(st, 1 + 2 * move), ROW(st, 2 + 2 * move), player);
        fill_new_card(st, tier, index, seed, rng);
    } else if (move < 27) {                            /* _reserve :382-400 */
        int i = move - 12;
        int8_t* res = ROW(st, L.r_res + 6 * player);
        int slot = -1;
        for (int s = 0; s < 3; s++)
            if (sum5(res + 2 * s * COLS) == 0) { slot = s; break; }
        if (slot < 0) slot = 2; /* invalid move in the reference (NameError); unreachable for valid play */
        int8_t* dst = res + 2 * slot * COLS;
        if (i < 12) {
            int tier = i / 4, index = i % 4;
            memcpy(dst, ROW(st, 1 + 8 * tier + 2 * index), 14);
            fill_new_card(st, tier, index, seed, rng);
        } else {
            int8_t card[14];
            if (get_deck_card(st, i - 12, seed, rng, card)) memcpy(dst, card, 14);
        }
        if (bank[GOLD] > 0 && sum7(gems) <= 9) { gems[GOLD] += 1; bank[GOLD] -= 1; }
    } else if (move < 30) {                            /* _buy_reserve :414-420 */
        int i = move - 27;
        int8_t* res = ROW(st, L.r_res + 6 * player);
        int8_t c0[7], c1[7];
        memcpy(c0, res + (2 * i) * COLS, 7);
        memcpy(c1, res + (2 * i + 1) * COLS, 7);
        buy_card(st, &L, c0, c1, player);
        if (i < 2) memmove(res + 2 * i * COLS, res + (2 * i + 2) * COLS, (size_t)(4 - 2 * i) * COLS);
        memset(res + 4 * COLS, 0, 2 * COLS);
    } else if (move < 60) {                            /* _get_gems :436-444 */
        int i = move - 30;
        for (int c = 0; c < 5; c++) {
            int8_t k = i < 25 ? SPL_GEMS3[i][c] : (int8_t)((c == i - 25) ? 2 : 0);
            bank[c] -= k; gems[c] += k;
        }
    } else if (move < 80) {                            /* _give_gems :455-463 */
        int i = move - 60;
        for (int c = 0; c < 5; c++) {
            int8_t k = i < 15 ? SPL_GEMS2[i][c] : (int8_t)((c == i - 15) ? 2 : 0);
            bank[c] += k; gems[c] -= k;
        }
    }
    bank[PTS] = (int8_t)(bank[PTS] + 1);               /* :203 move counter, int8 wrap */
    return (player + 1) % g->P;
}

int splendor_get_round(const azo_game* g, const int8_t* st) { (void)g; return (uint8_t)st[PTS]; } /* :303-304 */

int splendor_get_score(const azo_game* g, const int8_t* st, int p) {     /* :151-154 */
    lay_t L = lay(g->P);
    int s = ROW(st, L.r_pcards + p)[PTS];
    for (int i = 0; i < L.nn; i++) s += ROW(st, L.r_pnobles + L.nn * p + i)[PTS];
    return s;
}

void splendor_game_ended(const azo_game* g, const int8_t* st, int next_player, float* out) {  /* :221-240 */
    (void)next_player;
    lay_t L = lay(g->P);
    int n = g->P, round = splendor_get_round(g, st);
    for (int p = 0; p < n; p++) out[p] = 0.f;
    if (round % n != 0) return;
    float scores[AZO_MAX_PLAYERS], mx = -1e30f;
    for (int p = 0; p < n; p++) { scores[p] = (float)splendor_get_score(g, st, p); if (scores[p] > mx) mx = scores[p]; }
    int max_moves = 62 * n;
    if (!(mx >= 15.f || round >= max_moves)) return;
    int cnt = 0;
    for (int p = 0; p < n; p++) cnt += scores[p] == mx;
    int several = cnt > 1;
    if (several) {
        for (int p = 0; p < n; p++) {
            int nb = sum5(ROW(st, L.r_pcards + p));
            scores[p] = (float)((double)scores[p] - (double)nb / 100.);
        }
        mx = -1e30f;
        for (int p = 0; p < n; p++) if (scores[p] > mx) mx = scores[p];
        cnt = 0;
        for (int p = 0; p < n; p++) cnt += scores[p] == mx;
        several = cnt > 1;
    }
    for (int p = 0; p < n; p++) out[p] = (scores[p] == mx) ? (several ? 0.01f : 1.f) : -1.f;
}

static void roll_rows(int8_t* base, int rows, int shift) {
    int8_t tmp[64 * COLS];
    memcpy(tmp, base, (size_t)rows * COLS);
    for (int i = 0; i < rows; i++) memcpy(base + i * COLS, tmp + ((i + shift) % rows) * COLS, COLS);
}

void splendor_swap_players(const azo_game* g, int8_t* st, int k) {       /* :244-253 */
    lay_t L = lay(g->P);
    int n = g->P;
    roll_rows(ROW(st, L.r_gems), n, k);
    roll_rows(ROW(st, L.r_pnobles), n * L.nn, L.nn * k);
    roll_rows(ROW(st, L.r_pcards), n, k);
    roll_rows(ROW(st, L.r_res), 6 * n, 6 * k);
}

void splendor_init_board(const azo_game* g, int8_t* st, azo_rng* rng) { /* init_game :156-175 */
    int n = g->P;
    memset(st, 0, (size_t)g->S);
    int8_t* bank = ROW(st, 0);
    int gems_in_play = n == 2 ? 4 : (n == 3 ? 5 : 7);
    for (int c = 0; c < 5; c++) bank[c] = (int8_t)gems_in_play;
    bank[GOLD] = 5;
    for (int t = 0; t < 3; t++) {
        int len = SPL_DECK_LEN[t];
        for (int c = 0; c < 5; c++) {
            ROW(st, 25 + 2 * t)[c] = (int8_t)len;
            ROW(st, 26 + 2 * t)[c] = (int8_t)(uint8_t)(0xFF00u >> len);  /* packbits(ones(len)) MSB-first */
        }
    }
    for (int t = 0; t < 3; t++)
        for (int i = 0; i < 4; i++) fill_new_card(st, t, i, 0, rng);
    /* nobles: n+1 distinct out of 10 (np.random.choice(replace=False) in the reference; our stream: partial
       Fisher-Yates, j = i + floor(u*(10-i))) */
    int perm[10];
    for (int i = 0; i < 10; i++) perm[i] = i;
    for (int i = 0; i < n + 1; i++) {
        int j = i + (int)(azo_rng_u01(rng) * (10 - i));
        if (j > 9) j = 9;
        int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        memcpy(ROW(st, 31 + i), SPL_NOBLES[perm[i]], COLS);
    }
}

/* get_symmetries :255-301.  Order: identity, 3 tiers x 3 card permutations, then reserve permutations per player. */
int splendor_symmetries(const azo_game* g, const int8_t* st, const float* pi, const uint8_t* valids,
                        int8_t* out_states, float* out_pi, uint8_t* out_valids, int max_sym) {
    lay_t L = lay(g->P);
    int S = g->S, A = 81, k = 0;
#define EMIT_BASE() do { if (k >= max_sym) return k; memcpy(out_states + (size_t)k * S, st, S); \
        memcpy(out_pi + (size_t)k * A, pi, A * sizeof(float)); memcpy(out_valids + (size_t)k * A, valids, A); } while (0)
    EMIT_BASE(); k++;
    for (int tier = 0; tier < 3; tier++)
        for (int s = 0; s < 3; s++) {
            EMIT_BASE();
            int8_t* o = out_states + (size_t)k * S;
            for (int i = 0; i < 4; i++) {
                int p = SPL_CARD_SYM[s][i];
                memcpy(ROW(o, 1 + 8 * tier + 2 * i), ROW(st, 1 + 8 * tier + 2 * p), 14);
                out_pi[(size_t)k * A + 4 * tier + i] = pi[4 * tier + p];
                out_pi[(size_t)k * A + 12 + 4 * tier + i] = pi[12 + 4 * tier + p];
                out_valids[(size_t)k * A + 4 * tier + i] = valids[4 * tier + p];
                out_valids[(size_t)k * A + 12 + 4 * tier + i] = valids[12 + 4 * tier + p];
            }
            k++;
        }
    for (int pl = 0; pl < g->P; pl++) {
        const int8_t* res = ROW(st, L.r_res + 6 * pl);
        int nb = 3;
        for (int c = 0; c < 3; c++) if (sum5(res + 2 * c * COLS) == 0) { nb = c; break; }
        for (int s = 0; s < 2; s++) {
            if (SPL_RESERVE_SYM[nb][s][0] < 0) continue;
            EMIT_BASE();
            int8_t* o = out_states + (size_t)k * S;
            for (int i = 0; i < 3; i++) {
                int p = SPL_RESERVE_SYM[nb][s][i];
                memcpy(ROW(o, L.r_res + 6 * pl + 2 * i), res + 2 * p * COLS, 14);
                if (pl == 0) {
                    out_pi[(size_t)k * A + 27 + i] = pi[27 + p];
                    out_valids[(size_t)k * A + 27 + i] = valids[27 + p];
                }
            }
            k++;
        }
    }
#undef EMIT_BASE
    return k;
}

/* RNG-free start state of SURVEY.md Appendix C.1 (built through the Board API, bypassing init_game's RNG) */
void splendor_known_start(const azo_game* g, int8_t* st) {
    int n = g->P;
    memset(st, 0, (size_t)g->S);
    int8_t* bank = ROW(st, 0);
    int gems_in_play = n == 2 ? 4 : (n == 3 ? 5 : 7);
    for (int c = 0; c < 5; c++) bank[c] = (int8_t)gems_in_play;
    bank[GOLD] = 5;
    for (int t = 0; t < 3; t++) {
        int len = SPL_DECK_LEN[t];
        for (int c = 0; c < 5; c++) {
            ROW(st, 25 + 2 * t)[c] = (int8_t)len;
            ROW(st, 26 + 2 * t)[c] = (int8_t)(uint8_t)(0xFF00u >> len);
        }
    }
    for (int t = 0; t < 3; t++)
        for (int i = 0; i < 4; i++) fill_new_card(st, t, i, 31416, NULL);
    for (int i = 0; i < n + 1; i++) memcpy(ROW(st, 31 + i), SPL_NOBLES[i], COLS);
}
