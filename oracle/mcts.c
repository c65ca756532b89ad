/* ORACLE (test infrastructure, not product).  Scalar C restatement of the reference's MCTS.py.
 *
 *   nodes_data dict keyed by board.tobytes()          MCTS.py:39,125-126   -> open-addressing table, full-key compare
 *   getActionProb                                     MCTS.py:49-103
 *   search (recursive)                                MCTS.py:105-184      -> iterative descent + explicit path
 *   applyDirNoise / softmax / normalise / np_roll     MCTS.py:187-197,205-207,250-261
 *   pick_highest_UCB                                  MCTS.py:210-230
 *   get_next_best_action_and_canonical_state          MCTS.py:233-248
 *
 * Dtypes follow the reference exactly: Ps f32[A], Qsa f64[A] (sentinel -42.), Nsa i64[A], Ns int, Qs f32.
 * No FMA contraction anywhere (compiled with -ffp-contract=off): the Python/NumPy reference rounds every operation. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "azg_oracle.h"

#define NANQ (-42.0)
#define EPS 1e-8
#define KFORCED 0.5
static const int64_t MAGIC_SEEDS[8] = {31416, 1, 14142, 42, 27183, 2, 16180, 7};  /* MCTS.py:14 */

typedef struct node {
    int8_t* key;
    float Es[AZO_MAX_PLAYERS];
    int has_policy;       /* Ps is not None */
    uint8_t* Vs;
    float* Ps;
    int64_t Ns;
    double* Qsa;
    int64_t* Nsa;
    int r;
    float Qs;
    uint64_t hash;
} node;

struct azo_mcts {
    azo_game g;
    azo_mcts_args a;
    int dirichlet_noise;
    node** tab;
    size_t cap, count;
    int last_cleaning;
    /* search-in-progress */
    int8_t* root;
    int step, nb_sims, is_full, forced;
    const double* dir_noise;
    int64_t random_seed;
    azo_rng* search_rng;   /* env steps inside the search (games with true randomness in make_move), borrowed */
    azo_rng own_rng;
    /* sim-in-progress */
    node** path_node;
    int* path_a;
    int* path_np;
    int depth, path_cap;
    int8_t* cur;           /* state being expanded (leaf board) */
    uint8_t* leaf_valids;
    int leaf_r;
    int leaf_dir;          /* apply noise to this leaf (root on step 0) */
    /* counters */
    uint64_t c_sims, c_levels, c_exp, c_sumvalid, c_term;
};

static uint64_t hash_bytes(const int8_t* p, int n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int i = 0; i < n; i++) { h ^= (uint8_t)p[i]; h *= 0x100000001b3ULL; }
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
    return h;
}

static node* tab_find(const azo_mcts* m, const int8_t* key, uint64_t h) {
    size_t mask = m->cap - 1, i = h & mask;
    while (m->tab[i]) {
        if (m->tab[i]->hash == h && memcmp(m->tab[i]->key, key, (size_t)m->g.S) == 0) return m->tab[i];
        i = (i + 1) & mask;
    }
    return NULL;
}

static void tab_insert_raw(node** tab, size_t cap, node* n) {
    size_t mask = cap - 1, i = n->hash & mask;
    while (tab[i]) i = (i + 1) & mask;
    tab[i] = n;
}

static void tab_grow(azo_mcts* m) {
    size_t ncap = m->cap * 2;
    node** nt = (node**)calloc(ncap, sizeof(node*));
    for (size_t i = 0; i < m->cap; i++) if (m->tab[i]) tab_insert_raw(nt, ncap, m->tab[i]);
    free(m->tab);
    m->tab = nt; m->cap = ncap;
}

static void node_free(node* n) {
    free(n->key); free(n->Vs); free(n->Ps); free(n->Qsa); free(n->Nsa); free(n);
}

static node* node_new(azo_mcts* m, const int8_t* key, uint64_t h, int r) {
    node* n = (node*)calloc(1, sizeof(node));
    n->key = (int8_t*)malloc((size_t)m->g.S);
    memcpy(n->key, key, (size_t)m->g.S);
    n->hash = h; n->r = r;
    if ((m->count + 1) * 2 > m->cap) tab_grow(m);
    tab_insert_raw(m->tab, m->cap, n);
    m->count++;
    return n;
}

azo_mcts* azo_mcts_create(const azo_game* g, const azo_mcts_args* args, int dirichlet_noise) {
    azo_mcts* m = (azo_mcts*)calloc(1, sizeof(*m));
    m->g = *g; m->a = *args; m->dirichlet_noise = dirichlet_noise;
    m->cap = 1024;
    m->tab = (node**)calloc(m->cap, sizeof(node*));
    m->root = (int8_t*)malloc((size_t)g->S);
    m->cur = (int8_t*)malloc((size_t)g->S);
    m->leaf_valids = (uint8_t*)malloc((size_t)g->A);
    m->path_cap = 1024;
    m->path_node = (node**)malloc(sizeof(node*) * (size_t)m->path_cap);
    m->path_a = (int*)malloc(sizeof(int) * (size_t)m->path_cap);
    m->path_np = (int*)malloc(sizeof(int) * (size_t)m->path_cap);
    m->random_seed = -1;
    memset(&m->own_rng, 0, sizeof(m->own_rng));      /* default search stream: (seed 0, stream 0), counter 0 */
    m->search_rng = &m->own_rng;
    return m;
}

void azo_mcts_set_rng(azo_mcts* m, azo_rng* rng) { m->search_rng = rng ? rng : &m->own_rng; }

void azo_mcts_reset(azo_mcts* m) {   /* reset_all_search_trees MCTS.py:199-203 */
    for (size_t i = 0; i < m->cap; i++) if (m->tab[i]) { node_free(m->tab[i]); m->tab[i] = NULL; }
    m->count = 0; m->last_cleaning = 0;
}

void azo_mcts_destroy(azo_mcts* m) {
    if (!m) return;
    azo_mcts_reset(m);
    free(m->tab); free(m->root); free(m->cur); free(m->leaf_valids);
    free(m->path_node); free(m->path_a); free(m->path_np); free(m);
}

size_t azo_mcts_num_nodes(const azo_mcts* m) { return m->count; }

/* np.sum of a float32 array: NumPy's pairwise summation (8 accumulators, blocks of 128).  `normalise` MCTS.py:250-253
   calls np.sum; in pure-Python mode this is the order that is executed and pinned. */
static float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum_f32(a, n2) + np_sum_f32(a + n2, n - n2);
    }
}

static void normalise(float* v, int n) {
    float s = np_sum_f32(v, n);
    for (int i = 0; i < n; i++) v[i] /= s;
}

/* softmax(Ps, T) MCTS.py:255-261 */
static void softmax_temp(const azo_mcts* m, float* Ps, int n, double T) {
    if (T == 1.) return;
    if (m->a.numpy2_scalar_typing) {
        float e = (float)(1. / T);
        for (int i = 0; i < n; i++) Ps[i] = powf(Ps[i], e);
        normalise(Ps, n);
    } else {
        double* t = (double*)malloc(sizeof(double) * (size_t)n);
        double s = 0;
        for (int i = 0; i < n; i++) { t[i] = pow((double)Ps[i], 1. / T); s += t[i]; }
        for (int i = 0; i < n; i++) Ps[i] = (float)(t[i] / s);
        free(t);
    }
}

/* applyDirNoise MCTS.py:187-197 : Ps[idx] = 0.75*Ps[idx] + 0.25*dir  (f32*weak -> f32, + f64 -> f64, store f32) */
static void apply_dir_noise(float* Ps, const uint8_t* Vs, int n, const double* dir) {
    int k = 0;
    for (int i = 0; i < n; i++)
        if (Vs[i]) {
            float a = 0.75f * Ps[i];
            Ps[i] = (float)((double)a + 0.25 * dir[k]);
            k++;
        }
}

/* pick_highest_UCB MCTS.py:210-230 */
static int pick_highest_ucb(const azo_mcts* m, const node* nd, int forced, int n_iter) {
    const int A = m->g.A;
    const double cpuct = m->a.cpuct, fpu = m->a.fpu;
    int best = -1;
    double cur_best = -INFINITY;
    if (!m->a.numpy2_scalar_typing) {
        double fpu_init = fpu > 0 ? (double)nd->Qs - fpu : fpu;
        for (int a = 0; a < A; a++) {
            if (!nd->Vs[a]) continue;
            if (forced) {
                double t = KFORCED * (double)nd->Ps[a] * (double)n_iter;
                if (nd->Nsa[a] < (int64_t)sqrt(t)) return a;
            }
            double u;
            if (nd->Qsa[a] != NANQ)
                u = nd->Qsa[a] + cpuct * (double)nd->Ps[a] * sqrt((double)nd->Ns) / (double)(1 + nd->Nsa[a]);
            else
                u = fpu_init + cpuct * (double)nd->Ps[a] * sqrt((double)nd->Ns + EPS);
            if (u > cur_best) { cur_best = u; best = a; }
        }
    } else {
        float fpu_init32 = fpu > 0 ? nd->Qs - (float)fpu : (float)fpu;
        for (int a = 0; a < A; a++) {
            if (!nd->Vs[a]) continue;
            if (forced) {
                float t = ((float)KFORCED * nd->Ps[a]) * (float)n_iter;
                if (nd->Nsa[a] < (int64_t)sqrt((double)t)) return a;
            }
            double u;
            float t1 = (float)cpuct * nd->Ps[a];
            if (nd->Qsa[a] != NANQ) {
                float t2 = t1 * (float)sqrt((double)nd->Ns);
                u = nd->Qsa[a] + (double)t2 / (double)(1 + nd->Nsa[a]);
            } else {
                float t2 = t1 * (float)sqrt((double)nd->Ns + EPS);
                u = (double)(float)(fpu_init32 + t2);
            }
            if (u > cur_best) { cur_best = u; best = a; }
        }
    }
    return best;
}

void azo_mcts_search_begin(azo_mcts* m, const int8_t* canonical, int force_full_search, double u_full,
                           const double* dir_noise) {
    memcpy(m->root, canonical, (size_t)m->g.S);
    m->is_full = force_full_search || (u_full < m->a.prob_fullMCTS);           /* MCTS.py:58 */
    m->nb_sims = m->is_full ? m->a.numMCTSSims : m->a.numMCTSSims / m->a.ratio_fullMCTS;
    m->forced = m->is_full && m->a.forced_playouts;
    m->step = 0;
    m->dir_noise = dir_noise;
}

int azo_mcts_search_done(const azo_mcts* m) { return m->step >= m->nb_sims; }

static void roll_v(float* v, int P, int n) {   /* np.roll(v, n): out[i] = v[(i-n) mod P]  MCTS.py:205-207 */
    float t[AZO_MAX_PLAYERS];
    for (int i = 0; i < P; i++) t[i] = v[((i - n) % P + P) % P];
    memcpy(v, t, sizeof(float) * (size_t)P);
}

static void backup(azo_mcts* m, float* v) {     /* MCTS.py:176-183, unwinding the recursion */
    const int P = m->g.P;
    for (int d = m->depth - 1; d >= 0; d--) {
        node* nd = m->path_node[d];
        int a = m->path_a[d];
        roll_v(v, P, m->path_np[d]);
        nd->Qsa[a] = ((double)nd->Nsa[a] * nd->Qsa[a] + (double)v[0]) / (double)(nd->Nsa[a] + 1);
        float t = (float)(nd->Ns + 1) * nd->Qs;
        t = t + v[0];
        nd->Qs = t / (float)(nd->Ns + 2);
        nd->Nsa[a] += 1;
        nd->Ns += 1;
    }
}

/* One simulation up to the point where the net is needed.  Returns 1 if a leaf awaits (pi, v). */
int azo_mcts_sim_begin(azo_mcts* m) {
    const azo_game* g = &m->g;
    const int S = g->S, P = g->P;
    m->random_seed = m->a.universes > 0 ? MAGIC_SEEDS[m->step % m->a.universes] : -1;   /* MCTS.py:63 */
    int dir_root = (m->step == 0 && m->is_full && m->dirichlet_noise && m->dir_noise != NULL);  /* :64 */
    memcpy(m->cur, m->root, (size_t)S);
    m->depth = 0;
    m->c_sims++;
    for (;;) {
        int at_root = (m->depth == 0);
        uint64_t h = hash_bytes(m->cur, S);
        node* nd = tab_find(m, m->cur, h);
        float Es[AZO_MAX_PLAYERS];
        int any = 0;
        if (!nd) {
            int r = azo_get_round(g, m->cur);                                     /* :127-128 */
            azo_game_ended(g, m->cur, 0, Es);                                     /* :131 */
            for (int p = 0; p < P; p++) any |= (Es[p] != 0.f);
            if (any) {                                                            /* :132-135 terminal, stored */
                nd = node_new(m, m->cur, h, r);
                memcpy(nd->Es, Es, sizeof(float) * (size_t)P);
                m->c_term++;
                float v[AZO_MAX_PLAYERS];
                memcpy(v, Es, sizeof(v));
                backup(m, v);
                m->step++;
                return 0;
            }
            /* first visit: needs the net :140-146 */
            azo_valid_moves(g, m->cur, 0, m->leaf_valids);
            m->leaf_r = r;
            m->leaf_dir = at_root && dir_root;
            return 1;
        }
        for (int p = 0; p < P; p++) any |= (nd->Es[p] != 0.f);
        if (any) {                                                                /* :136-138 */
            m->c_term++;
            float v[AZO_MAX_PLAYERS];
            memcpy(v, nd->Es, sizeof(v));
            backup(m, v);
            m->step++;
            return 0;
        }
        if (at_root && dir_root) {                                                /* :156-160 */
            softmax_temp(m, nd->Ps, g->A, m->a.temperature[2]);
            apply_dir_noise(nd->Ps, nd->Vs, g->A, m->dir_noise);
            normalise(nd->Ps, g->A);
        }
        int a = pick_highest_ucb(m, nd, at_root && m->forced, m->step);          /* :164-173 (forced only at root :175) */
        if (m->depth >= m->path_cap) return -1;
        m->c_levels++;
        for (int i = 0; i < g->A; i++) m->c_sumvalid += nd->Vs[i];
        int np = azo_make_move(g, m->cur, a, 0, m->random_seed, m->search_rng);   /* :238-239 */
        if (np != 0) azo_swap_players(g, m->cur, np);                             /* :243-245 */
        m->path_node[m->depth] = nd; m->path_a[m->depth] = a; m->path_np[m->depth] = np;
        m->depth++;
    }
}

const int8_t* azo_mcts_leaf_board(const azo_mcts* m) { return m->cur; }
const uint8_t* azo_mcts_leaf_valids(const azo_mcts* m) { return m->leaf_valids; }

void azo_mcts_sim_finish(azo_mcts* m, const float* pi, const float* v_in) {
    const azo_game* g = &m->g;
    const int A = g->A, P = g->P;
    uint64_t h = hash_bytes(m->cur, g->S);
    node* nd = node_new(m, m->cur, h, m->leaf_r);
    nd->has_policy = 1;
    nd->Vs = (uint8_t*)malloc((size_t)A);
    memcpy(nd->Vs, m->leaf_valids, (size_t)A);
    nd->Ps = (float*)malloc(sizeof(float) * (size_t)A);
    memcpy(nd->Ps, pi, sizeof(float) * (size_t)A);
    if (m->leaf_dir) {                                                            /* :147-149 */
        softmax_temp(m, nd->Ps, A, m->a.temperature[2]);
        apply_dir_noise(nd->Ps, nd->Vs, A, m->dir_noise);
    }
    normalise(nd->Ps, A);                                                         /* :150 */
    nd->Qsa = (double*)malloc(sizeof(double) * (size_t)A);
    nd->Nsa = (int64_t*)calloc((size_t)A, sizeof(int64_t));
    for (int i = 0; i < A; i++) nd->Qsa[i] = NANQ;                                /* :40-41,152 */
    nd->Ns = 0;
    nd->Qs = v_in[0];                                                             /* :153 */
    m->c_exp++;
    float v[AZO_MAX_PLAYERS];
    memcpy(v, v_in, sizeof(float) * (size_t)P);
    backup(m, v);                                                                 /* leaf returns v un-negated :154 */
    m->step++;
}

int azo_mcts_search_end(azo_mcts* m, double temp, double* probs, float* q) {
    const azo_game* g = &m->g;
    const int A = g->A, P = g->P;
    node* root = tab_find(m, m->root, hash_bytes(m->root, g->S));
    if (!root) return -1;
    int64_t* counts = (int64_t*)malloc(sizeof(int64_t) * (size_t)A);
    for (int a = 0; a < A; a++) counts[a] = root->has_policy ? root->Nsa[a] : 0;        /* :68 */
    float q0 = root->Qs;                                                          /* :71-72 */
    for (int p = 0; p < P; p++) q[p] = p == 0 ? q0 : -q0 / (float)(P - 1);
    if (m->forced && root->has_policy) {                                          /* :75-80 policy target pruning */
        int64_t best = 0;
        for (int a = 0; a < A; a++) if (counts[a] > best) best = counts[a];
        for (int a = 0; a < A; a++) {
            int64_t c = counts[a];
            if (c != best) {
                /* k*Psa*nb_MCTS_sims: python_float*np.float32 -> float32 under NumPy>=2 (this code is plain Python
                   in the reference, so NumPy scalar typing applies in both oracle modes) */
                float t = ((float)KFORCED * root->Ps[a]) * (float)m->nb_sims;
                c = c - (int64_t)sqrt((double)t);
            }
            counts[a] = c > 1 ? c : 0;
        }
    }
    if (!m->a.no_mem_optim) {                                                     /* :86-91 */
        int r = azo_get_round(g, m->root);
        if (r > m->last_cleaning + 20) {
            node** nt = (node**)calloc(m->cap, sizeof(node*));
            size_t kept = 0;
            for (size_t i = 0; i < m->cap; i++) {
                node* n = m->tab[i];
                if (!n) continue;
                if (n->r < r - 5) node_free(n);
                else { tab_insert_raw(nt, m->cap, n); kept++; }
            }
            free(m->tab); m->tab = nt; m->count = kept;
            m->last_cleaning = r;
        }
    }
    if (temp <= 0.02) {                                                           /* :93-98 (first maximum; the
                                                                                     reference picks one at random) */
        int64_t best = -1; int ba = 0;
        for (int a = 0; a < A; a++) if (counts[a] > best) { best = counts[a]; ba = a; }
        for (int a = 0; a < A; a++) probs[a] = 0.;
        probs[ba] = 1.;
    } else {                                                                      /* :100-103 */
        double s = 0.;
        for (int a = 0; a < A; a++) { probs[a] = pow((double)counts[a], 1. / temp); s += probs[a]; }
        for (int a = 0; a < A; a++) probs[a] = probs[a] / s;
    }
    free(counts);
    return m->is_full;
}

int azo_mcts_get_action_prob(azo_mcts* m, const int8_t* canonical, double temp, int force_full_search, double u_full,
                             const double* dir_noise, azo_predict_fn predict, void* ctx, double* probs, float* q) {
    float* pi = (float*)malloc(sizeof(float) * (size_t)m->g.A);
    float v[AZO_MAX_PLAYERS];
    azo_mcts_search_begin(m, canonical, force_full_search, u_full, dir_noise);
    while (!azo_mcts_search_done(m)) {
        int need = azo_mcts_sim_begin(m);
        if (need < 0) { free(pi); return -1; }
        if (need) {
            predict(ctx, m->cur, m->leaf_valids, pi, v);
            azo_mcts_sim_finish(m, pi, v);
        }
    }
    free(pi);
    return azo_mcts_search_end(m, temp, probs, q);
}

int azo_mcts_node_stats(const azo_mcts* m, const int8_t* state, int64_t* Ns, float* Qs, int64_t* Nsa, double* Qsa,
                        float* Ps, float* Es, int* has_policy) {
    node* n = tab_find(m, state, hash_bytes(state, m->g.S));
    if (!n) return 0;
    if (Ns) *Ns = n->Ns;
    if (Qs) *Qs = n->Qs;
    if (has_policy) *has_policy = n->has_policy;
    if (Es) memcpy(Es, n->Es, sizeof(float) * (size_t)m->g.P);
    if (n->has_policy) {
        if (Nsa) memcpy(Nsa, n->Nsa, sizeof(int64_t) * (size_t)m->g.A);
        if (Qsa) memcpy(Qsa, n->Qsa, sizeof(double) * (size_t)m->g.A);
        if (Ps) memcpy(Ps, n->Ps, sizeof(float) * (size_t)m->g.A);
    }
    return 1;
}

size_t azo_mcts_dump_keys(const azo_mcts* m, int8_t* out, size_t max_nodes) {
    size_t k = 0;
    for (size_t i = 0; i < m->cap; i++)
        if (m->tab[i]) {
            if (k < max_nodes) memcpy(out + k * (size_t)m->g.S, m->tab[i]->key, (size_t)m->g.S);
            k++;
        }
    return k;
}

void azo_mcts_counters(const azo_mcts* m, uint64_t* sims, uint64_t* levels, uint64_t* expansions,
                       uint64_t* sum_valid_visited, uint64_t* terminal_hits) {
    if (sims) *sims = m->c_sims;
    if (levels) *levels = m->c_levels;
    if (expansions) *expansions = m->c_exp;
    if (sum_valid_visited) *sum_valid_visited = m->c_sumvalid;
    if (terminal_hits) *terminal_hits = m->c_term;
}

/* SURVEY.md Appendix C.3 hash-net */
void azo_hashnet_predict(void* ctx, const int8_t* board, const uint8_t* valids, float* pi, float* v) {
    const azo_game* g = (const azo_game*)ctx;
    int64_t s = 0;
    for (int i = 0; i < g->S; i++) s += (int64_t)board[i] * (int64_t)(i + 1);
    /* Python: (s * 2654435761) % 2**32 with floor-mod on a possibly negative s */
    uint32_t h = (uint32_t)((uint64_t)s * 2654435761ULL);
    float v0 = (float)((double)h / 2147483648.0 - 1.0);
    for (int p = 0; p < g->P; p++) v[p] = p == 0 ? v0 : (float)(-v0 / (float)(g->P - 1));
    int64_t sum = 0;
    for (int a = 0; a < g->A; a++) {
        uint32_t t = (uint32_t)((uint64_t)(h >> 8) + 2654435761ULL * (uint64_t)a);
        int64_t w = valids[a] ? 1 + (int64_t)(t % 13u) : 0;
        sum += w;
    }
    for (int a = 0; a < g->A; a++) {
        uint32_t t = (uint32_t)((uint64_t)(h >> 8) + 2654435761ULL * (uint64_t)a);
        int64_t w = valids[a] ? 1 + (int64_t)(t % 13u) : 0;
        pi[a] = (float)((double)w / (double)sum);
    }
}
