/* ORACLE (test infrastructure).  Game dispatch: mirrors the <G>Game adaptors (splendor/SplendorGame.py:16-60,
 * santorini/SantoriniGame.py:16-60): every Game.py call = Board.copy_state + one Board method. */
#include <string.h>
#include "azg_oracle.h"

void splendor_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  splendor_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void splendor_game_ended(const azo_game*, const int8_t*, int, float*);
void splendor_swap_players(const azo_game*, int8_t*, int);
int  splendor_get_round(const azo_game*, const int8_t*);
int  splendor_get_score(const azo_game*, const int8_t*, int);
void splendor_init_board(const azo_game*, int8_t*, azo_rng*);
int  splendor_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void santorini_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  santorini_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void santorini_game_ended(const azo_game*, const int8_t*, int, float*);
void santorini_swap_players(const azo_game*, int8_t*, int);
int  santorini_get_round(const azo_game*, const int8_t*);
int  santorini_get_score(const azo_game*, const int8_t*, int);
void santorini_init_board(const azo_game*, int8_t*, azo_rng*);
int  santorini_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void azul_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  azul_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void azul_game_ended(const azo_game*, const int8_t*, int, float*);
void azul_swap_players(const azo_game*, int8_t*, int);
int  azul_get_round(const azo_game*, const int8_t*);
int  azul_get_score(const azo_game*, const int8_t*, int);
void azul_init_board(const azo_game*, int8_t*, azo_rng*);
void azul_known_start(const azo_game*, int8_t*);
int  azul_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void minivilles_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  minivilles_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void minivilles_game_ended(const azo_game*, const int8_t*, int, float*);
void minivilles_swap_players(const azo_game*, int8_t*, int);
int  minivilles_get_round(const azo_game*, const int8_t*);
int  minivilles_get_score(const azo_game*, const int8_t*, int);
void minivilles_init_board(const azo_game*, int8_t*, azo_rng*);
int  minivilles_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void abalone_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  abalone_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void abalone_game_ended(const azo_game*, const int8_t*, int, float*);
void abalone_swap_players(const azo_game*, int8_t*, int);
int  abalone_get_round(const azo_game*, const int8_t*);
int  abalone_get_score(const azo_game*, const int8_t*, int);
void abalone_init_board(const azo_game*, int8_t*, azo_rng*);
int  abalone_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void smallworld_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  smallworld_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void smallworld_game_ended(const azo_game*, const int8_t*, int, float*);
void smallworld_swap_players(const azo_game*, int8_t*, int);
int  smallworld_get_round(const azo_game*, const int8_t*);
int  smallworld_get_score(const azo_game*, const int8_t*, int);
void smallworld_init_board(const azo_game*, int8_t*, azo_rng*);
int  smallworld_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);
int  smallworld_symmetries_rng(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int, azo_rng*);

void akropolis_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  akropolis_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void akropolis_game_ended(const azo_game*, const int8_t*, int, float*);
void akropolis_swap_players(const azo_game*, int8_t*, int);
int  akropolis_get_round(const azo_game*, const int8_t*);
int  akropolis_get_score(const azo_game*, const int8_t*, int);
void akropolis_init_board(const azo_game*, int8_t*, azo_rng*);
int  akropolis_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void botanik_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  botanik_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void botanik_game_ended(const azo_game*, const int8_t*, int, float*);
void botanik_swap_players(const azo_game*, int8_t*, int);
int  botanik_get_round(const azo_game*, const int8_t*);
int  botanik_get_score(const azo_game*, const int8_t*, int);
void botanik_init_board(const azo_game*, int8_t*, azo_rng*);
int  botanik_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);

void tlp_valid_moves(const azo_game*, const int8_t*, int, uint8_t*);
int  tlp_make_move(const azo_game*, int8_t*, int, int, int64_t, azo_rng*);
void tlp_game_ended(const azo_game*, const int8_t*, int, float*);
void tlp_swap_players(const azo_game*, int8_t*, int);
int  tlp_get_round(const azo_game*, const int8_t*);
int  tlp_get_score(const azo_game*, const int8_t*, int);
void tlp_init_board(const azo_game*, int8_t*, azo_rng*);
int  tlp_symmetries(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int);
int  tlp_symmetries_rng(const azo_game*, const int8_t*, const float*, const uint8_t*, int8_t*, float*, uint8_t*, int, azo_rng*);

int azo_game_init(azo_game* g, int game_id, int variant) {
    memset(g, 0, sizeof(*g));
    g->id = game_id;
    g->variant = variant;
    if (game_id == AZO_SPLENDOR) {
        int n = variant ? variant : 2;
        if (n < 2 || n > 4) return -1;
        g->variant = n;
        g->P = n;
        g->rows = 32 + 10 * n + n * n;      /* observation_size, SplendorLogicNumba.py:90-92 */
        g->cols = 7;
        g->S = g->rows * g->cols;
        g->A = 81;
        return 0;
    }
    if (game_id == AZO_SANTORINI) {
        int gods = variant ? variant : 11;
        if (gods != 1 && gods != 11) return -1;
        g->variant = gods;
        g->P = 2;
        g->rows = 25; g->cols = 3;          /* (5,5,3) SantoriniLogicNumba.py:13-15 */
        g->S = 75;
        g->A = gods * 2 * 9 * 9;            /* :17-19 */
        return 0;
    }
    if (game_id == AZO_ABALONE) {
        g->variant = 1;                      /* INITIAL_LAYOUT = 1 (Belgian Daisy), no dynamic komi */
        g->P = 2;
        g->rows = 81; g->cols = 4;           /* observation_size (9, 9, 4), AbaloneLogicNumba.py:46-48 */
        g->S = 324;
        g->A = 3402;                         /* :50-52 */
        return 0;
    }
    if (game_id == AZO_SMALLWORLD) {
        const int n = variant ? variant : 2, na = n == 2 ? 23 : (n == 3 ? 30 : 39);      /* SmallworldMaps_<n>pl.py NB_AREAS */
        if (n < 2 || n > 4) return -1;
        g->variant = n;
        g->P = n;
        g->rows = na + 5 * n + 7; g->cols = 8;   /* observation_size, SmallworldLogicNumba.py:92-94 */
        g->S = g->rows * 8;
        g->A = 5 * na + 16;                  /* action_size :96-98 */
        return 0;
    }
    if (game_id == AZO_AKROPOLIS) {
        const int n = variant ? variant : 2;                                            /* AkropolisConstants.py N_PLAYERS */
        if (n < 2 || n > 4) return -1;
        g->variant = n;
        g->P = n;
        g->rows = 169; g->cols = 3 * n + 2;  /* observation_size (13, 13, 3 n + 2), AkropolisLogicNumba.py:66-68 */
        g->S = 169 * g->cols;
        g->A = (n + 2) * 1014;               /* :70-72 */
        return 0;
    }
    if (game_id == AZO_BOTANIK) {
        g->variant = 2;
        g->P = 2;
        g->rows = 66 * 5; g->cols = 7;       /* observation_size (66, 5, 7), BotanikLogicNumba.py:93-95 (MACHINE_SIZE = 7) */
        g->S = 2310;
        g->A = 428;                          /* :97-99 */
        return 0;
    }
    if (game_id == AZO_TLP) {
        int n = variant ? variant : 3;
        if (n < 3 || n > 5) return -1;
        g->variant = n;
        g->P = n;
        g->rows = 18 * n + 1; g->cols = 15;  /* observation_size, TLPLogicNumba.py:38-40 */
        g->S = g->rows * g->cols;
        g->A = n * n;                        /* :42-44 */
        return 0;
    }
    if (game_id == AZO_MINIVILLES) {
        int n = variant ? variant : 2;
        if (n < 2 || n > 4) return -1;
        g->variant = n;
        g->P = n;
        g->rows = 18 + 20 * n; g->cols = 2;  /* observation_size, MinivillesLogicNumba.py:39-41 */
        g->S = g->rows * g->cols;
        g->A = 21;                           /* :43-45 */
        return 0;
    }
    if (game_id == AZO_AZUL) {
        g->variant = 2;
        g->P = 2;
        g->rows = 23; g->cols = 6;          /* observation_size, AzulLogicNumba.py:50-52 */
        g->S = 138;
        g->A = 180;                         /* :55-57 */
        return 0;
    }
    return -1;
}

void azo_valid_moves(const azo_game* g, const int8_t* s, int p, uint8_t* out) {
    if (g->id == AZO_SMALLWORLD) smallworld_valid_moves(g, s, p, out);
    else if (g->id == AZO_AKROPOLIS) akropolis_valid_moves(g, s, p, out);
    else if (g->id == AZO_BOTANIK) botanik_valid_moves(g, s, p, out);
    else if (g->id == AZO_TLP) tlp_valid_moves(g, s, p, out);
    else if (g->id == AZO_ABALONE) abalone_valid_moves(g, s, p, out);
    else if (g->id == AZO_MINIVILLES) minivilles_valid_moves(g, s, p, out);
    else if (g->id == AZO_SPLENDOR) splendor_valid_moves(g, s, p, out);
    else if (g->id == AZO_AZUL) azul_valid_moves(g, s, p, out);
    else santorini_valid_moves(g, s, p, out);
}
int azo_make_move(const azo_game* g, int8_t* s, int mv, int p, int64_t seed, azo_rng* rng) {
    if (g->id == AZO_SMALLWORLD) return smallworld_make_move(g, s, mv, p, seed, rng);
    if (g->id == AZO_AKROPOLIS) return akropolis_make_move(g, s, mv, p, seed, rng);
    if (g->id == AZO_BOTANIK) return botanik_make_move(g, s, mv, p, seed, rng);
    if (g->id == AZO_TLP) return tlp_make_move(g, s, mv, p, seed, rng);
    if (g->id == AZO_ABALONE) return abalone_make_move(g, s, mv, p, seed, rng);
    if (g->id == AZO_MINIVILLES) return minivilles_make_move(g, s, mv, p, seed, rng);
    if (g->id == AZO_AZUL) return azul_make_move(g, s, mv, p, seed, rng);
    return g->id == AZO_SPLENDOR ? splendor_make_move(g, s, mv, p, seed, rng) : santorini_make_move(g, s, mv, p, seed, rng);
}
void azo_game_ended(const azo_game* g, const int8_t* s, int np, float* out) {
    if (g->id == AZO_SMALLWORLD) smallworld_game_ended(g, s, np, out);
    else if (g->id == AZO_AKROPOLIS) akropolis_game_ended(g, s, np, out);
    else if (g->id == AZO_BOTANIK) botanik_game_ended(g, s, np, out);
    else if (g->id == AZO_TLP) tlp_game_ended(g, s, np, out);
    else if (g->id == AZO_ABALONE) abalone_game_ended(g, s, np, out);
    else if (g->id == AZO_MINIVILLES) minivilles_game_ended(g, s, np, out);
    else if (g->id == AZO_SPLENDOR) splendor_game_ended(g, s, np, out);
    else if (g->id == AZO_AZUL) azul_game_ended(g, s, np, out);
    else santorini_game_ended(g, s, np, out);
}
void azo_swap_players(const azo_game* g, int8_t* s, int k) {
    if (g->id == AZO_SMALLWORLD) smallworld_swap_players(g, s, k);
    else if (g->id == AZO_AKROPOLIS) akropolis_swap_players(g, s, k);
    else if (g->id == AZO_BOTANIK) botanik_swap_players(g, s, k);
    else if (g->id == AZO_TLP) tlp_swap_players(g, s, k);
    else if (g->id == AZO_ABALONE) abalone_swap_players(g, s, k);
    else if (g->id == AZO_MINIVILLES) minivilles_swap_players(g, s, k);
    else if (g->id == AZO_SPLENDOR) splendor_swap_players(g, s, k);
    else if (g->id == AZO_AZUL) azul_swap_players(g, s, k);
    else santorini_swap_players(g, s, k);
}
int azo_get_round(const azo_game* g, const int8_t* s) {
    if (g->id == AZO_SMALLWORLD) return smallworld_get_round(g, s);
    if (g->id == AZO_AKROPOLIS) return akropolis_get_round(g, s);
    if (g->id == AZO_BOTANIK) return botanik_get_round(g, s);
    if (g->id == AZO_TLP) return tlp_get_round(g, s);
    if (g->id == AZO_ABALONE) return abalone_get_round(g, s);
    if (g->id == AZO_MINIVILLES) return minivilles_get_round(g, s);
    if (g->id == AZO_AZUL) return azul_get_round(g, s);
    return g->id == AZO_SPLENDOR ? splendor_get_round(g, s) : santorini_get_round(g, s);
}
int azo_get_score(const azo_game* g, const int8_t* s, int p) {
    if (g->id == AZO_SMALLWORLD) return smallworld_get_score(g, s, p);
    if (g->id == AZO_AKROPOLIS) return akropolis_get_score(g, s, p);
    if (g->id == AZO_BOTANIK) return botanik_get_score(g, s, p);
    if (g->id == AZO_TLP) return tlp_get_score(g, s, p);
    if (g->id == AZO_ABALONE) return abalone_get_score(g, s, p);
    if (g->id == AZO_MINIVILLES) return minivilles_get_score(g, s, p);
    if (g->id == AZO_AZUL) return azul_get_score(g, s, p);
    return g->id == AZO_SPLENDOR ? splendor_get_score(g, s, p) : santorini_get_score(g, s, p);
}
void azo_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {
    if (g->id == AZO_SMALLWORLD) smallworld_init_board(g, s, rng);
    else if (g->id == AZO_AKROPOLIS) akropolis_init_board(g, s, rng);
    else if (g->id == AZO_BOTANIK) botanik_init_board(g, s, rng);
    else if (g->id == AZO_TLP) tlp_init_board(g, s, rng);
    else if (g->id == AZO_ABALONE) abalone_init_board(g, s, rng);
    else if (g->id == AZO_MINIVILLES) minivilles_init_board(g, s, rng);
    else if (g->id == AZO_SPLENDOR) splendor_init_board(g, s, rng);
    else if (g->id == AZO_AZUL) azul_init_board(g, s, rng);
    else santorini_init_board(g, s, rng);
}
void azo_canonical(const azo_game* g, const int8_t* s, int player, int8_t* out) {
    /* getCanonicalForm: SplendorGame.py:42-48, SantoriniGame.py:42-48 */
    if (out != s) memcpy(out, s, (size_t)g->S);
    if (player != 0) azo_swap_players(g, out, player);
}
int azo_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                   uint8_t* ov, int max_sym) {
    if (g->id == AZO_SMALLWORLD) return smallworld_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    if (g->id == AZO_AKROPOLIS) return akropolis_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    if (g->id == AZO_BOTANIK) return botanik_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    if (g->id == AZO_TLP) return tlp_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    if (g->id == AZO_ABALONE) return abalone_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    if (g->id == AZO_MINIVILLES) return minivilles_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    if (g->id == AZO_AZUL) return azul_symmetries(g, s, pi, valids, os, op, ov, max_sym);
    return g->id == AZO_SPLENDOR ? splendor_symmetries(g, s, pi, valids, os, op, ov, max_sym)
                                 : santorini_symmetries(g, s, pi, valids, os, op, ov, max_sym);
}
/* get_symmetries of a game whose symmetries are themselves random (The Little Prince shuffles); other games ignore rng */
int azo_symmetries_rng(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                       uint8_t* ov, int max_sym, azo_rng* rng) {
    if (g->id == AZO_TLP) return tlp_symmetries_rng(g, s, pi, valids, os, op, ov, max_sym, rng);
    if (g->id == AZO_SMALLWORLD) return smallworld_symmetries_rng(g, s, pi, valids, os, op, ov, max_sym, rng);
    return azo_symmetries(g, s, pi, valids, os, op, ov, max_sym);
}

void splendor_known_start(const azo_game*, int8_t*);
void santorini_known_start(const azo_game*, int8_t*, int, int);
void azo_known_start(const azo_game* g, int8_t* s, int a, int b) {
    if (g->id == AZO_SMALLWORLD) { azo_rng r; memset(&r, 0, sizeof(r)); smallworld_init_board(g, s, &r); }
    else if (g->id == AZO_AKROPOLIS) { azo_rng r; memset(&r, 0, sizeof(r)); akropolis_init_board(g, s, &r); }
    else if (g->id == AZO_BOTANIK) { azo_rng r; memset(&r, 0, sizeof(r)); botanik_init_board(g, s, &r); }
    else if (g->id == AZO_TLP) { azo_rng r; memset(&r, 0, sizeof(r)); tlp_init_board(g, s, &r); }                      /* first market from stream (0,0) */
    else if (g->id == AZO_ABALONE) abalone_init_board(g, s, NULL);
    else if (g->id == AZO_MINIVILLES) { azo_rng r; memset(&r, 0, sizeof(r)); minivilles_init_board(g, s, &r); }   /* first dice from stream (0,0) */
    else if (g->id == AZO_SPLENDOR) splendor_known_start(g, s);
    else if (g->id == AZO_AZUL) azul_known_start(g, s);
    else santorini_known_start(g, s, a, b);
}

const char* azo_version(void) { return "azg-oracle r2"; }
