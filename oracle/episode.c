/* ORACLE (test infrastructure).  Coach.executeEpisode (Coach.py:37-84) + temp_for_selfplay (:266-271) +
 * applyTemperatureAndNormalize / random_pick (:278-292), driven by the counter-based RNG contract (rng.c).
 *
 * Per-ply draw order (shared with the HIP engine):
 *   1. u_full  -> `rng.random() < prob_fullMCTS`                      MCTS.py:58
 *   2. u_pick  -> np.random.choice(len(p), p=p) == searchsorted(cumsum(p)/sum, u, 'right')   Coach.py:289-292
 *   3. the uniforms consumed by make_move(random_seed=0)              Coach.py:71
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "azg_oracle.h"

static double temp_for_selfplay(const azo_episode_cfg* c, int n) {
    double tb = c->temp_begin, te = c->temp_end, hl = c->tempThreshold;
    if (hl < 0) return (n > -hl) ? te : tb;
    return te + (tb - te) * pow(0.5, (double)n / hl);
}

static int random_pick(const double* probs, int A, double temperature, double u) {
    double* w = (double*)malloc(sizeof(double) * (size_t)A);
    if (temperature == 0) {
        double mx = -1;
        int nb = 0;
        for (int a = 0; a < A; a++) if (probs[a] > mx) mx = probs[a];
        for (int a = 0; a < A; a++) nb += probs[a] == mx;
        int k = (int)(u * nb), pick = 0;
        if (k >= nb) k = nb - 1;
        for (int a = 0; a < A; a++) if (probs[a] == mx) { if (k-- == 0) { pick = a; break; } }
        free(w);
        return pick;
    }
    double s = 0;
    for (int a = 0; a < A; a++) { w[a] = pow(probs[a], 1. / temperature); s += w[a]; }
    double cdf = 0, tot = 0;
    for (int a = 0; a < A; a++) { w[a] = w[a] / s; tot += w[a]; }
    int pick = -1, last = 0;
    for (int a = 0; a < A; a++) {
        cdf += w[a];
        if (w[a] > 0) last = a;
        if (cdf / tot > u) { pick = a; break; }
    }
    free(w);
    return pick < 0 ? last : pick;
}

int azo_episode_run(const azo_game* g, const azo_episode_cfg* cfg, const int8_t* init_board, uint64_t seed,
                    uint64_t stream, azo_predict_fn predict, void* ctx, int8_t* out_canonical, double* out_pi,
                    float* out_q, int32_t* out_action, int32_t* out_player, int32_t* out_full, float* out_result,
                    int8_t* out_final_board) {
    const int S = g->S, A = g->A, P = g->P;
    azo_rng rng;
    memset(&rng, 0, sizeof(rng));
    rng.mode = 0; rng.seed = seed; rng.stream = stream; rng.counter = 0;
    int8_t* board = (int8_t*)malloc((size_t)S);
    int8_t* canon = (int8_t*)malloc((size_t)S);
    if (init_board) memcpy(board, init_board, (size_t)S);
    else azo_init_board(g, board, &rng);
    azo_mcts* m = azo_mcts_create(g, &cfg->mcts, 0);
    azo_mcts_set_rng(m, &rng);                  /* search-time env randomness (Minivilles) draws from the episode's stream */
    int cur = 0, step = 0, plies = 0;
    float r[AZO_MAX_PLAYERS];
    for (;;) {
        if (plies >= cfg->max_plies) { plies = -2; break; }
        step++;
        azo_canonical(g, board, cur, canon);                                        /* Coach.py:61 */
        double u_full = azo_rng_u01(&rng);
        double* pi = out_pi + (size_t)plies * A;
        float* q = out_q + (size_t)plies * P;
        int is_full = azo_mcts_get_action_prob(m, canon, 1.0, 0, u_full, NULL, predict, ctx, pi, q);   /* :62 */
        if (is_full < 0) { plies = -1; break; }
        double u_pick = azo_rng_u01(&rng);
        const double temp = temp_for_selfplay(cfg, step);
        int action = random_pick(pi, A, temp, u_pick);                              /* :63 */
        /* temperature == 0 (Coach.py:278-292): applyTemperatureAndNormalize draws np.random.choice(bests) -- u_pick above, the k-th of
           the nb maxima with k = floor(u * nb), the contract's unweighted choice -- and random_pick then still calls
           np.random.choice(len(p), p = the one-hot result): a second uniform is consumed, its outcome is the one-hot index */
        if (temp == 0) (void)azo_rng_u01(&rng);
        memcpy(out_canonical + (size_t)plies * S, canon, (size_t)S);
        out_action[plies] = action; out_player[plies] = cur; out_full[plies] = is_full;
        plies++;
        cur = azo_make_move(g, board, action, cur, 0, &rng);                        /* :71 true random */
        azo_game_ended(g, board, cur, r);                                           /* :73 */
        int any = 0;
        for (int p = 0; p < P; p++) any |= r[p] != 0.f;
        if (any) break;
    }
    if (plies > 0) {
        memcpy(out_result, r, sizeof(float) * (size_t)P);
        memcpy(out_final_board, board, (size_t)S);
    }
    azo_mcts_destroy(m);
    free(board); free(canon);
    return plies;
}
