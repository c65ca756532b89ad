/* ORACLE (test infrastructure).  The Little Prince ("Make me a planet") env step: a scalar restatement of
 * thelittleprince/TLPLogicNumba.py (Board :95-412) for 3-5 players.
 *
 * State = int8 [18 n + 1][15] (copy_state :147-156): row 0 = round_and_state (col 0 round, col 1 current player, col 2
 * bitfield of who can still play this turn (player p = bit 128 >> p), cols 3..12 bitfield of the 80 cards still in the deck,
 * MSB first); rows 1..n market; rows n+1..2n players_score (one column per attribute; the FACE_DOWN column carries the volcano
 * penalty); rows 2n+1.. players_cards, 16 slots per player.  A card row = 14 attribute counts + the card type
 * (25 centre, 50 uphill edge, 75 downhill edge, 100 + character for corners).
 * Action = card * n + player_delta (:20-33).
 *
 * Randomness: make_move ignores random_seed; the market refill (:366-392) draws 1 + n uniforms from the global RNG through
 * my_random_choice_and_normalize (:46-50) = searchsorted(cumsum(mask / mask.sum()), u, 'right'), in MCTS simulations too.
 * get_symmetries (:177-272) is itself random (np.random.shuffle of players / market cards / planet slots); the contract here
 * defines shuffle as Fisher-Yates from the top, j = floor(u * (i + 1)) for i = len-1 .. 1 (tools/refshim/harness.py
 * CounterRandom.shuffle feeds the reference the same). */
#include <string.h>
#include "azg_oracle.h"
#include "tlp_tables.h"

enum { FACE_DOWN, BAOBAB, VOLCANO, SUNSET, ROSE, LAMPPOST, BOX, BIG_STAR, FOX, ELEPHANT, SNAKE, SHEEP_WHITE, SHEEP_GREY, SHEEP_BROWN,
       CARD_TYPE };
enum { NONE, VAIN_MAN, GEOGRAPHER, ASTRONOMER, KING, LAMPLIGHTER, HUNTER, DRUNKARD, BUSINESSMAN_W, BUSINESSMAN_G, BUSINESSMAN_B,
       GARDENER, TURKISH, LITTLE_PRINCE };
#define COLS 15
#define ROW(s, r) ((s) + (r) * COLS)
#define MARKET(s, i) ROW(s, 1 + (i))
#define SCORE(s, n, p) ROW(s, (n) + 1 + (p))
#define CARD(s, n, i) ROW(s, 2 * (n) + 1 + (i))

static const int SLOTS[4][4] = {{5, 6, 9, 10}, {1, 7, 8, 14}, {2, 4, 11, 13}, {0, 3, 12, 15}};       /* slots_in_planet :76-88 */

static int can_play(const int8_t* s, int p) { return ((uint8_t)s[2] >> (7 - p)) & 1; }
static void set_can_play(int8_t* s, int p, int v) {
    uint8_t b = (uint8_t)s[2];
    b = (uint8_t)(v ? (b | (128u >> p)) : (b & ~(128u >> p)));
    s[2] = (int8_t)b;
}
static int8_t all_can_play(int n) { return (int8_t)(uint8_t)(0xFF00u >> n); }

/* my_random_choice_and_normalize :46-50 on a 0/1 mask */
static int choice(azo_rng* rng, const int* mask, int len) {
    int k = 0, i;
    for (i = 0; i < len; i++) k += mask[i] != 0;
    const double u = azo_rng_u01(rng);
    double c = 0.0;
    for (i = 0; i < len; i++) {
        c += (mask[i] ? 1.0 : 0.0) / (double)k;
        if (c > u) return i;
    }
    return len - 1;                                    /* u above the rounded total (probability ~1e-16): clamp */
}

static void card_row(int group, int idx, int8_t* out) {               /* np_all_cards[group][idx] */
    const uint32_t w = TLP_CARD_ATTR[20 * group + idx];
    for (int k = 0; k < 14; k++) out[k] = (int8_t)((w >> (2 * k)) & 3u);
    out[CARD_TYPE] = (int8_t)(group == 3 ? 100 + TLP_CORNER_CHAR[idx] : 25 * (group + 1));
}

static void fill_market_if_needed(const azo_game* g, int8_t* s, azo_rng* rng) {                       /* :366-392 */
    const int n = g->P;
    int i, all_full = 1;
    for (i = 0; i < n; i++) if (MARKET(s, i)[CARD_TYPE] != 0) return;
    for (i = 0; i < 16 * n; i++) if (!(CARD(s, n, i)[CARD_TYPE] > 0)) { all_full = 0; break; }
    if (all_full) return;
    int room[4] = {CARD(s, n, 10)[CARD_TYPE] == 0, CARD(s, n, 14)[CARD_TYPE] == 0, CARD(s, n, 13)[CARD_TYPE] == 0,
                   CARD(s, n, 15)[CARD_TYPE] == 0};
    const int group = choice(rng, room, 4);
    for (i = 0; i < n; i++) {
        int avail[20];
        for (int c = 0; c < 20; c++) {
            const int bit = 20 * group + c;
            avail[c] = ((uint8_t)s[3 + (bit >> 3)] >> (7 - (bit & 7))) & 1;
        }
        const int idx = choice(rng, avail, 20);
        card_row(group, idx, MARKET(s, i));
        const int bit = 20 * group + idx;
        s[3 + (bit >> 3)] = (int8_t)((uint8_t)s[3 + (bit >> 3)] & ~(128u >> (bit & 7)));
    }
    s[2] = all_can_play(n);
}

static void take_card(const azo_game* g, int8_t* s, int i, int p) {                                    /* :283-300 */
    const int n = g->P, ct = MARKET(s, i)[CARD_TYPE];
    const int grp = ct == 25 ? 0 : (ct == 50 ? 1 : (ct == 75 ? 2 : 3));
    int best = -1;
    for (int k = 0; k < 4; k++)
        if (CARD(s, n, 16 * p + SLOTS[grp][k])[CARD_TYPE] == 0) { best = 16 * p + SLOTS[grp][k]; break; }
    if (best < 0) best = 16 * n - 1;                   /* players_cards[-1] as written; unreachable in play */
    memcpy(CARD(s, n, best), MARKET(s, i), COLS);
    memset(MARKET(s, i), 0, COLS);
    int baobabs = 0;
    for (int c = 0; c < 16; c++) baobabs += CARD(s, n, 16 * p + c)[BAOBAB];
    if (baobabs >= 3)
        for (int c = 0; c < 16; c++) {
            int8_t* row = CARD(s, n, 16 * p + c);
            if (row[BAOBAB] >= 1) { memset(row, 0, CARD_TYPE); row[FACE_DOWN] = 1; }
        }
}

static void update_score(const azo_game* g, int8_t* s, int p) {                                        /* :303-364 */
    const int n = g->P;
    int sum[COLS] = {0};
    for (int c = 0; c < 16; c++)
        for (int k = 0; k < COLS; k++) sum[k] += CARD(s, n, 16 * p + c)[k];
    int8_t* sc = SCORE(s, n, p);
    memset(sc, 0, COLS);
    for (int k = 0; k < 4; k++) {
        const int ct = CARD(s, n, 16 * p + SLOTS[3][k])[CARD_TYPE];
        const int ch = ct - 100 > 0 ? ct - 100 : 0;
        if (ch == NONE) continue;
        switch (ch) {
        case VAIN_MAN: sc[SNAKE] = (int8_t)(sc[SNAKE] + 4 * sum[SNAKE]); break;
        case GEOGRAPHER:
            for (int c = 0; c < 16; c++)
                if (c != 0 && c != 3 && c != 12 && c != 15 && CARD(s, n, 16 * p + c)[VOLCANO] == 0) sc[VOLCANO] = (int8_t)(sc[VOLCANO] + 1);
            break;
        case ASTRONOMER: sc[SUNSET] = (int8_t)(sc[SUNSET] + 2 * sum[SUNSET]); break;
        case KING: { static const int roses[4] = {0, 14, 7, 0}; sc[ROSE] = (int8_t)(sc[ROSE] + roses[sum[ROSE] < 3 ? sum[ROSE] : 3]); break; }
        case LAMPLIGHTER: sc[LAMPPOST] = (int8_t)(sc[LAMPPOST] + sum[LAMPPOST]); break;
        case HUNTER:
            sc[SNAKE] = (int8_t)(sc[SNAKE] + (sum[SNAKE] > 0 ? 3 : 0));
            sc[ELEPHANT] = (int8_t)(sc[ELEPHANT] + (sum[ELEPHANT] > 0 ? 3 : 0));
            if (sum[SHEEP_WHITE] > 0) sc[SHEEP_WHITE] = (int8_t)(sc[SHEEP_WHITE] + 3);
            else if (sum[SHEEP_GREY] > 0) sc[SHEEP_GREY] = (int8_t)(sc[SHEEP_GREY] + 3);
            else if (sum[SHEEP_BROWN] > 0) sc[SHEEP_BROWN] = (int8_t)(sc[SHEEP_BROWN] + 3);
            break;
        case DRUNKARD: sc[BAOBAB] = (int8_t)(sc[BAOBAB] + 3 * sum[FACE_DOWN]); break;
        case BUSINESSMAN_W: sc[SHEEP_WHITE] = (int8_t)(sc[SHEEP_WHITE] + 2 * sum[SHEEP_WHITE]); break;
        case BUSINESSMAN_G: sc[SHEEP_GREY] = (int8_t)(sc[SHEEP_GREY] + 3 * sum[SHEEP_GREY]); break;
        case BUSINESSMAN_B: sc[SHEEP_BROWN] = (int8_t)(sc[SHEEP_BROWN] + 5 * sum[SHEEP_BROWN]); break;
        case GARDENER: sc[BAOBAB] = (int8_t)(sc[BAOBAB] + 7 * sum[BAOBAB]); break;
        case TURKISH: sc[BIG_STAR] = (int8_t)(sc[BIG_STAR] + sum[BIG_STAR]); break;
        case LITTLE_PRINCE:
            if (sum[SHEEP_WHITE] > 0) sc[SHEEP_WHITE] = (int8_t)(sc[SHEEP_WHITE] + 3);
            if (sum[SHEEP_GREY] > 0) sc[SHEEP_GREY] = (int8_t)(sc[SHEEP_GREY] + 3);
            if (sum[SHEEP_BROWN] > 0) sc[SHEEP_BROWN] = (int8_t)(sc[SHEEP_BROWN] + 3);
            sc[BOX] = (int8_t)(sc[BOX] + sum[BOX]);
            break;
        default: break;
        }
        /* volcano penalty of every player, kept in the FACE_DOWN column of the score rows (:351-357) */
        int nb[AZO_MAX_PLAYERS], mx = -1000;
        for (int q = 0; q < n; q++) {
            nb[q] = 0;
            for (int c = 0; c < 16; c++) nb[q] += CARD(s, n, 16 * q + c)[VOLCANO];
            if (nb[q] > mx) mx = nb[q];
        }
        for (int q = 0; q < n; q++) SCORE(s, n, q)[FACE_DOWN] = (int8_t)(nb[q] == mx ? -mx : 0);
    }
}

void tlp_valid_moves(const azo_game* g, const int8_t* s, int player, uint8_t* out) {                   /* :119-133 */
    const int n = g->P;
    int who[AZO_MAX_PLAYERS], any = 0;
    for (int p = 0; p < n; p++) { who[p] = p != player && can_play(s, p); any |= who[p]; }
    if (!any) who[player] = 1;
    memset(out, 0, (size_t)g->A);
    for (int d = 0; d < n; d++)
        if (who[(player + d) % n])
            for (int i = 0; i < n; i++)
                if (MARKET(s, i)[CARD_TYPE] != 0) out[i * n + d] = 1;
}

int tlp_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t seed, azo_rng* rng) {     /* :135-145 */
    const int n = g->P, card = move / n, next = (player + move % n) % n;
    (void)seed;
    take_card(g, s, card, player);
    update_score(g, s, player);
    fill_market_if_needed(g, s, rng);
    set_can_play(s, player, 0);
    s[0] = (int8_t)(s[0] + 1);
    s[1] = (int8_t)next;
    return next;
}

int tlp_get_score(const azo_game* g, const int8_t* s, int p) {                                          /* :104-105 */
    int t = 0;
    for (int k = 0; k < COLS; k++) t += SCORE(s, g->P, p)[k];
    return t;
}
int tlp_get_round(const azo_game* g, const int8_t* s) { (void)g; return s[0]; }

void tlp_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) {                  /* :158-166 */
    const int n = g->P;
    (void)next_player;
    for (int p = 0; p < n; p++) out[p] = 0.f;
    if (s[0] < 16 * n) return;
    int sc[AZO_MAX_PLAYERS], mx = -1000, cnt = 0;
    for (int p = 0; p < n; p++) { sc[p] = (int8_t)tlp_get_score(g, s, p); if (sc[p] > mx) mx = sc[p]; }
    for (int p = 0; p < n; p++) cnt += sc[p] == mx;
    for (int p = 0; p < n; p++) out[p] = sc[p] == mx ? (cnt == 1 ? 1.f : 0.01f) : -1.f;
}

void tlp_swap_players(const azo_game* g, int8_t* s, int k) {                                            /* :170-182 */
    const int n = g->P;
    int8_t tmp[16 * AZO_MAX_PLAYERS * COLS];
    memcpy(tmp, SCORE(s, n, 0), (size_t)n * COLS);
    for (int i = 0; i < n; i++) memcpy(SCORE(s, n, i), tmp + ((i + k) % n) * COLS, COLS);
    memcpy(tmp, CARD(s, n, 0), (size_t)16 * n * COLS);
    for (int i = 0; i < 16 * n; i++) memcpy(CARD(s, n, i), tmp + ((i + 16 * k) % (16 * n)) * COLS, COLS);
    s[1] = (int8_t)((s[1] - k + n) % n);
    uint8_t b = 0;
    for (int i = 0; i < n; i++) if (can_play(s, (i + k) % n)) b |= (uint8_t)(128u >> i);
    s[2] = (int8_t)b;
}

void tlp_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                                       /* :107-117 */
    memset(s, 0, (size_t)g->S);
    s[2] = all_can_play(g->P);
    for (int i = 3; i < 13; i++) s[i] = -1;
    fill_market_if_needed(g, s, rng);
}

/* ---- get_symmetries :177-272 ---- */
static void shuffle(azo_rng* rng, int* a, int len) {
    for (int i = len - 1; i > 0; i--) {
        int j = (int)(azo_rng_u01(rng) * (double)(i + 1));
        if (j > i) j = i;
        const int t = a[i]; a[i] = a[j]; a[j] = t;
    }
}
static void permute_players(const azo_game* g, azo_rng* rng, const int* players, int len, int8_t* st, float* pi, uint8_t* v) {
    const int n = g->P;
    int sh[AZO_MAX_PLAYERS];
    int8_t in[(18 * AZO_MAX_PLAYERS + 1) * COLS];
    float ipi[AZO_MAX_PLAYERS * AZO_MAX_PLAYERS];
    uint8_t iv[AZO_MAX_PLAYERS * AZO_MAX_PLAYERS];
    for (int i = 0; i < len; i++) sh[i] = players[i];
    shuffle(rng, sh, len);
    memcpy(in, st, (size_t)g->S); memcpy(ipi, pi, sizeof(float) * (size_t)g->A); memcpy(iv, v, (size_t)g->A);
    uint8_t bits = (uint8_t)in[2];
    for (int i = 0; i < len; i++) {
        const int o = players[i], w = sh[i];
        memcpy(SCORE(st, n, w), SCORE(in, n, o), COLS);
        memcpy(CARD(st, n, 16 * w), CARD(in, n, 16 * o), 16 * COLS);
        bits = (uint8_t)(can_play(in, o) ? (bits | (128u >> w)) : (bits & ~(128u >> w)));
        for (int c = 0; c < n; c++) { pi[c * n + w] = ipi[c * n + o]; v[c * n + w] = iv[c * n + o]; }
    }
    st[2] = (int8_t)bits;
}

int tlp_symmetries_rng(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                       uint8_t* ov, int max_sym, azo_rng* rng) {
    const int n = g->P, S = g->S, A = g->A;
    int k = 0;
    int8_t st[(18 * AZO_MAX_PLAYERS + 1) * COLS], in[(18 * AZO_MAX_PLAYERS + 1) * COLS];
    float p2[AZO_MAX_PLAYERS * AZO_MAX_PLAYERS];
    uint8_t v2[AZO_MAX_PLAYERS * AZO_MAX_PLAYERS];
#define EMIT(state, ppi, vv) do { int dup = 0; for (int e = 0; e < k; e++) if (!memcmp(os + (size_t)e * S, state, (size_t)S)) { dup = 1; break; } \
        if (!dup && k < max_sym) { memcpy(os + (size_t)k * S, state, (size_t)S); memcpy(op + (size_t)k * A, ppi, sizeof(float) * (size_t)A); \
                                   memcpy(ov + (size_t)k * A, vv, (size_t)A); k++; } } while (0)
    EMIT(s, pi, valids);
    const int cur = s[1];
    int played[AZO_MAX_PLAYERS], fresh[AZO_MAX_PLAYERS], np_ = 0, nf = 0;
    for (int i = 0; i < n; i++) {
        if (i == cur) continue;
        if (can_play(s, i)) fresh[nf++] = i; else played[np_++] = i;
    }
    for (int it = 0; it < n; it++) {
        memcpy(st, s, (size_t)S); memcpy(p2, pi, sizeof(float) * (size_t)A); memcpy(v2, valids, (size_t)A);
        permute_players(g, rng, played, np_, st, p2, v2);
        permute_players(g, rng, fresh, nf, st, p2, v2);
        EMIT(st, p2, v2);
    }
    for (int it = 0; it < n; it++) {
        int list[16], sh[16], len = 0;
        for (int i = 0; i < n; i++) if (MARKET(s, i)[CARD_TYPE] != 0) list[len++] = i;
        memcpy(sh, list, sizeof(int) * (size_t)len);
        shuffle(rng, sh, len);
        memcpy(st, s, (size_t)S);
        for (int i = 0; i < len; i++) memcpy(MARKET(st, sh[i]), MARKET(s, list[i]), COLS);
        for (int p = 0; p < n; p++)
            for (int ct = 1; ct <= 4; ct++) {
                len = 0;
                for (int i = 0; i < 16; i++) if (CARD(s, n, 16 * p + i)[CARD_TYPE] / 25 == ct) list[len++] = i;
                memcpy(sh, list, sizeof(int) * (size_t)len);
                shuffle(rng, sh, len);
                memcpy(in, st, (size_t)S);
                for (int i = 0; i < len; i++) memcpy(CARD(st, n, 16 * p + sh[i]), CARD(in, n, 16 * p + list[i]), COLS);
            }
        EMIT(st, pi, valids);                          /* pi and valids are returned unpermuted, as written (:223,237) */
    }
#undef EMIT
    return k;
}

int tlp_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op, uint8_t* ov,
                   int max_sym) {
    azo_rng r;
    memset(&r, 0, sizeof(r));
    return tlp_symmetries_rng(g, s, pi, valids, os, op, ov, max_sym, &r);
}
