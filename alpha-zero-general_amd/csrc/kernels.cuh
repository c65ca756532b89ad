// kernels.cuh -- __global__ kernels of the engine, templated on the device game (SplendorDev<n>, SantoriniDev<g>).
// One workgroup = one wavefront (64 threads) = one state (env kernels) or one tree (forest kernels).
#pragma once
#include "forest.cuh"

namespace azg {

// ======================================================================================================================
// batched env step kernels  (azg_env_*)
// ======================================================================================================================
template <class G>
__global__ __launch_bounds__(64) void k_env_valid_moves(const int8_t* states, const int32_t* players, int n,
                                                        uint8_t* out) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    __shared__ __attribute__((aligned(16))) uint64_t mask[G::AW];
    int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    G::valid_mask(st, players ? players[t] : 0, mask);
    wave_sync();
    for (int a = lane_id(); a < G::A; a += 64) out[(size_t)t * G::A + a] = (uint8_t)((mask[a >> 6] >> (a & 63)) & 1);
}

template <class G>
__global__ __launch_bounds__(64) void k_env_next_state(const int8_t* states, const int32_t* players,
                                                       const int32_t* actions, const int64_t* seeds, int n,
                                                       int8_t* out_states, int32_t* out_next, uint64_t rng_seed,
                                                       uint64_t stream0, uint64_t* counters) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    int np = 0;
    if (lane_id() == 0) {
        Rng rng{rng_seed, stream0 + (uint64_t)t, counters ? counters[t] : 0ull};
        np = G::make_move(st, actions[t], players ? players[t] : 0, seeds ? (long long)seeds[t] : 0ll, rng);
        if (counters) counters[t] = rng.counter;
        out_next[t] = np;
    }
    wave_sync();
    Forest<G>::store_state_unpadded(out_states + (size_t)t * G::S, st);
}

template <class G>
__global__ __launch_bounds__(64) void k_env_game_ended(const int8_t* states, const int32_t* next_players, int n,
                                                       float* out_ended, int32_t* out_scores, int32_t* out_round) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    __shared__ __attribute__((aligned(16))) uint64_t mask[G::AW];
    int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    float es[G::P];
    G::game_ended(st, next_players ? next_players[t] : 0, es, mask);
    if (lane_id() == 0) {
        for (int p = 0; p < G::P; p++) {
            if (out_ended) out_ended[(size_t)t * G::P + p] = es[p];
            if (out_scores) out_scores[(size_t)t * G::P + p] = G::get_score(st, p);
        }
        if (out_round) out_round[t] = G::get_round(st);
    }
}

template <class G>
__global__ __launch_bounds__(64) void k_env_canonical(const int8_t* states, const int32_t* players, int n,
                                                      int8_t* out_states) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    __shared__ __attribute__((aligned(16))) int8_t tmp[G::SP];
    int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    int p = players[t];
    if (p != 0) G::swap_players(st, tmp, p);
    Forest<G>::store_state_unpadded(out_states + (size_t)t * G::S, st);
}

template <class G>
__global__ __launch_bounds__(64) void k_env_init_boards(int n, int8_t* out_states, uint64_t rng_seed, uint64_t stream0,
                                                        uint64_t* out_counters) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    int t = blockIdx.x;
    if (t >= n) return;
    for (int i = lane_id(); i < G::SP; i += 64) st[i] = 0;
    wave_sync();
    if (lane_id() == 0) {
        Rng rng{rng_seed, stream0 + (uint64_t)t, 0ull};
        G::init_board(st, rng);
        if (out_counters) out_counters[t] = rng.counter;
    }
    wave_sync();
    Forest<G>::store_state_unpadded(out_states + (size_t)t * G::S, st);
}

// ======================================================================================================================
// forest kernels
// ======================================================================================================================
template <class G>
__global__ __launch_bounds__(64) void k_forest_reset(ForestDev F) {
    int t = blockIdx.x;
    uint32_t* tab = Forest<G>::htab(F, t);
    for (int i = lane_id(); i < F.HT; i += 64) tab[i] = AZG_NONE;
    if (lane_id() == 0) {
        TreeHdr* H = &F.hdr[t];
        H->n_nodes = 0; H->heap_top = 0; H->root = AZG_NONE; H->status = ST_IDLE;
        H->sim_idx = 0; H->n_sims = 0; H->pending_leaf = AZG_NONE; H->path_len = 0;
    }
}

// Set up a search from the canonical state in LDS `sm.st` (MCTS.getActionProb prologue, MCTS.py:58-60).
template <class G>
__device__ void begin_search_from_lds(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm, bool full) {
    using FR = Forest<G>;
    FR::store_state(F.root_state + (size_t)t * G::SP, sm.st);
    uint64_t h = wave_hash_state((const uint32_t*)sm.st, FR::SPW);
    uint32_t free_slot;
    H.root = FR::probe(F, t, sm.st, h, &free_slot);
    H.root_round = (uint32_t)G::get_round(sm.st);
    H.is_full = full ? 1u : 0u;
    H.n_sims = (uint32_t)(full ? F.numMCTSSims : F.numMCTSSims / F.ratio_fullMCTS);
    H.forced = (full && F.forced_playouts) ? 1u : 0u;
    H.sim_idx = 0;
    H.status = ST_SEARCHING;
    H.pending_leaf = AZG_NONE;
}

template <class G>
__global__ __launch_bounds__(64) void k_begin_search(ForestDev F, const int8_t* roots, const uint8_t* full) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    int t = blockIdx.x;
    TreeHdr H = F.hdr[t];
    FR::load_state_unpadded(sm.st, roots + (size_t)t * G::S);
    begin_search_from_lds<G>(F, t, H, sm, full ? full[t] != 0 : true);
    if (lane_id() == 0) F.hdr[t] = H;
}

// pick_highest_UCB (MCTS.py:210-230) over the compact row of `node`; returns row index j (wave-uniform).
template <class G>
__device__ int pick_action(const ForestDev& F, int t, const NodeHdr& nh, const uint8_t* row, const RowLayout& L,
                           bool forced, uint32_t n_iter) {
    const int nv = nh.nv;
    const float* Prow = (const float*)row;
    const uint32_t* Nrow = (const uint32_t*)(row + L.offN);
    const double* Qrow = (const double*)(row + L.offQ);
    const double sqrtNs = sqrt((double)nh.Ns);
    const double sqrtNsEps = sqrt((double)nh.Ns + AZG_EPS);
    const double fpu_init = F.fpu > 0 ? (double)nh.Qs - F.fpu : F.fpu;
    double best_u = -INFINITY;
    int best_j = 0x7FFFFFFF;
    for (int base = 0; base < nv; base += 64) {
        int j = base + lane_id();
        bool act = j < nv;
        float p = act ? Prow[j] : 0.f;
        uint32_t n = act ? Nrow[j] : 0u;
        double q = act ? Qrow[j] : AZG_NANQ;
        if (forced) {                                                     // :218-220 first deficient action wins
            double thr = sqrt(0.5 * (double)p * (double)n_iter);
            uint64_t def = __ballot(act && ((long long)n < (long long)thr));
            if (def) return uni_i32(base + first_lane(def));
        }
        double u;
        if (q != AZG_NANQ) u = q + F.cpuct * (double)p * sqrtNs / (double)(1u + n);      // :223
        else u = fpu_init + F.cpuct * (double)p * sqrtNsEps;                            // :225
        if (!act) u = -INFINITY;
        int jj = act ? j : 0x7FFFFFFF;
        bool take = (u > best_u) || (u == best_u && jj < best_j);
        best_u = take ? u : best_u;
        best_j = take ? jj : best_j;
    }
    wave_argmax_f64(best_u, best_j);
    return uni_i32(best_j);
}

// Apply root Dirichlet noise to the existing root row (MCTS.py:156-160): scatter to dense, transform, gather back.
template <class G>
__device__ void noise_existing_root(const ForestDev& F, uint8_t* row, const RowLayout& L, int nv, float* dense,
                                    uint64_t* mask, const double* noise, bool normalised, double alpha, uint64_t gkey,
                                    uint64_t gctr) {
    float* Prow = (float*)row;
    const uint16_t* ids = (const uint16_t*)(row + L.offI);
    for (int i = lane_id(); i < G::A; i += 64) dense[i] = 0.f;
    if (lane_id() < G::AW) mask[lane_id()] = 0ull;
    wave_sync();
    for (int j = lane_id(); j < nv; j += 64) dense[ids[j]] = Prow[j];
    if (lane_id() == 0)
        for (int j = 0; j < nv; j++) mask[ids[j] >> 6] |= 1ull << (ids[j] & 63);
    wave_sync();
    Forest<G>::root_noise_dense(dense, mask, F.temp_root, noise, normalised, alpha, gkey, gctr);
    for (int j = lane_id(); j < nv; j += 64) Prow[j] = dense[ids[j]];
}

// One lock-step round, part 1 (MCTS.search descent, MCTS.py:105-175).
template <class G>
__global__ __launch_bounds__(64, 4) void k_select(ForestDev F, int8_t* leaf_states, uint8_t* leaf_valid,
                                               uint8_t* needs_eval, const double* root_noise, int noise_stride) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    const int t = blockIdx.x;
    const int l = lane_id();
    TreeHdr H = load_uniform(&F.hdr[t]);
    if (H.status != ST_SEARCHING) {
        if (l == 0) needs_eval[t] = 0;
        return;
    }
    uint8_t* hp = FR::heap(F, t);
    bool need_nn = false;
    Rng no_rng{0, 0, 0};
    while (true) {
        if (H.sim_idx >= H.n_sims) { H.status = ST_DONE; break; }
        if (H.err) { H.status = ST_DONE; break; }
        const int uidx = F.universes > 0 ? (int)(H.sim_idx % (uint32_t)F.universes) : 0;
        const long long seed = F.universes > 0 ? AZG_MAGIC_SEEDS[uidx] : -1ll;                 // MCTS.py:63
        // MCTS.py:64 -- noise source: caller tensor, or the engine's own Gamma sampler (root_noise == NULL, stride == -1)
        const bool gen_noise = (root_noise == nullptr && noise_stride == -1 && F.dirichletAlpha != 0.0);
        const bool dir_now = (H.sim_idx == 0 && H.is_full && (root_noise != nullptr || gen_noise));
        H.c_sims++;
        uint32_t node = H.root;
        int depth = 0;
        int pre = 0;
        bool have_leaf = false;
        H.leaf_is_root = 0;
        if (node == AZG_NONE) {
            // the root itself is not a node yet: it is the leaf of this simulation (MCTS.py:140-154)
            FR::load_state(sm.st, F.root_state + (size_t)t * G::SP);
            uint64_t h = wave_hash_state((const uint32_t*)sm.st, FR::SPW);
            uint32_t free_slot;
            uint32_t found = FR::probe(F, t, sm.st, h, &free_slot);
            if (found == AZG_NONE) {
                node = FR::create_node(F, t, H, sm.st, h, free_slot);
                if (node == AZG_NONE) continue;
                H.root = node;
                H.leaf_is_root = 1;
                have_leaf = true;
            } else {
                node = found;
                H.root = node;
            }
        }
        while (!have_leaf) {
            const NodeHdr nh = load_uniform(FR::nhdr(F, t, node));
            if (nh.flags & NF_TERMINAL) {                                                       // MCTS.py:136-138
                H.c_term++;
                float v[G::P];
#pragma unroll
                for (int p = 0; p < G::P; p++) v[p] = nh.Es[p];
                FR::backup(F, t, sm.path, depth, v);
                H.sim_idx++;
                break;
            }
            const RowLayout L(nh.nv, F.U);
            uint8_t* row = hp + (size_t)nh.row_off * 16u;
            if (depth == 0 && dir_now)
                noise_existing_root<G>(F, row, L, nh.nv, dense, sm.mask,
                                       root_noise ? root_noise + (size_t)t * (noise_stride < 0 ? -noise_stride : noise_stride)
                                                  : nullptr,
                                       root_noise != nullptr && noise_stride < 0, F.dirichletAlpha,
                                       mix64(mix64(F.rng_seed ^ 0xA5A5A5A55A5A5A5AULL) + F.stream0 + (uint64_t)t),
                                       H.c_sims << 20);
            const int j = pick_action<G>(F, t, nh, row, L, depth == 0 && H.forced, H.sim_idx);
            H.c_levels++;
            H.c_sumvalid += nh.nv;
            const uint16_t* ids = (const uint16_t*)(row + L.offI);
            uint32_t* crow = (uint32_t*)(row + L.offC);
            const int a = (int)uni_u32(ids[j]);
            uint32_t child = uni_u32(crow[j * F.U + uidx]);
            if (depth >= AZG_MAXD - 1) { H.err |= ERR_DEPTH_OVERFLOW; H.sim_idx = H.n_sims; break; }
            if (child == AZG_NONE) {
                // frontier edge: replay the env step from the parent's state (MCTS.py:233-248) and look the child up
                FR::load_state(sm.st, FR::nstate(F, t, node));
                int np = 0;
                if (l == 0) np = G::make_move(sm.st, a, 0, seed, no_rng);
                np = uni_i32(np);
                wave_sync();
                if (np != 0) G::swap_players(sm.st, sm.tmp, np);
                uint64_t h = wave_hash_state((const uint32_t*)sm.st, FR::SPW);
                uint32_t free_slot;
                uint32_t found = FR::probe(F, t, sm.st, h, &free_slot);
                if (found == AZG_NONE) {
                    found = FR::create_node(F, t, H, sm.st, h, free_slot);
                    if (found == AZG_NONE) { H.sim_idx = H.n_sims; break; }
                    have_leaf = true;
                }
                child = found | ((uint32_t)np << 30);
                if (l == 0) crow[j * F.U + uidx] = child;
            }
            const int np = (int)(child >> 30);
            if (l == 0) {
                PathEnt e; e.node = node; e.j = (uint16_t)j; e.np = (uint8_t)np; e.pre = (uint8_t)pre;
                sm.path[depth] = e;
            }
            wave_sync();
            pre = (pre + np) % G::P;
            depth++;
            node = child & AZG_CHILD_IDX_MASK;
        }
        if (!have_leaf) continue;
        // ---- new node `node`, its state is in sm.st: terminal test, valid moves, queue for the net ----
        float es[G::P];
        const bool ended = G::game_ended(sm.st, 0, es, sm.mask);                                 // MCTS.py:131
        NodeHdr* nhp = FR::nhdr(F, t, node);
        if (ended) {
            if (l == 0) {
                nhp->row_off = AZG_NONE; nhp->nv = 0; nhp->round = (uint8_t)G::get_round(sm.st);
                nhp->flags = NF_TERMINAL; nhp->Ns = 0; nhp->Qs = 0.f;
#pragma unroll
                for (int p = 0; p < G::P; p++) nhp->Es[p] = es[p];
            }
            H.c_term++;
            FR::backup(F, t, sm.path, depth, es);
            H.sim_idx++;
            continue;
        }
        G::valid_mask(sm.st, 0, sm.mask);                                                        // MCTS.py:142
        wave_sync();
        int nv = 0;
#pragma unroll
        for (int k = 0; k < G::AW; k++) nv += __popcll(sm.mask[k]);
        const RowLayout L(nv, F.U);
        if (H.heap_top + L.total / 16u > F.heap_units) { H.err |= ERR_HEAP_OVERFLOW; H.sim_idx = H.n_sims; continue; }
        const uint32_t row_off = H.heap_top;
        H.heap_top += L.total / 16u;
        uint8_t* row = hp + (size_t)row_off * 16u;
        uint16_t* ids = (uint16_t*)(row + L.offI);
        for (int a = l; a < G::A; a += 64) {
            uint64_t w = sm.mask[a >> 6];
            bool v = (w >> (a & 63)) & 1;
            if (v) {
                int rank = __popcll(w & ((1ull << (a & 63)) - 1ull));
                for (int k = 0; k < (a >> 6); k++) rank += __popcll(sm.mask[k]);
                ids[rank] = (uint16_t)a;
            }
            leaf_valid[(size_t)t * G::A + a] = (uint8_t)v;
        }
        if (l == 0) {
            nhp->row_off = row_off; nhp->nv = (uint16_t)nv; nhp->round = (uint8_t)G::get_round(sm.st);
            nhp->flags = 0; nhp->Ns = 0; nhp->Qs = 0.f;
#pragma unroll
            for (int p = 0; p < G::P; p++) nhp->Es[p] = 0.f;
        }
        FR::store_state_unpadded(leaf_states + (size_t)t * G::S, sm.st);
        PathEnt* gp = F.path + (size_t)t * AZG_MAXD;
        for (int d = l; d < depth; d += 64) gp[d] = sm.path[d];
        H.pending_leaf = node;
        H.path_len = (uint32_t)depth;
        H.status = ST_WAIT_NN;
        need_nn = true;
        break;
    }
    if (H.n_nodes > H.max_nodes_seen) H.max_nodes_seen = H.n_nodes;
    if (l == 0) {
        F.hdr[t] = H;
        needs_eval[t] = need_nn ? 1 : 0;
    }
}

// One lock-step round, part 2: store (Ps, v) on the pending leaf and back up (MCTS.py:144-154,176-183).
template <class G>
__global__ __launch_bounds__(64) void k_expand_backup(ForestDev F, const float* pi, const float* vin,
                                                      const double* root_noise, int noise_stride) {
    using FR = Forest<G>;
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    __shared__ __attribute__((aligned(16))) uint64_t mask[G::AW];
    __shared__ __attribute__((aligned(16))) PathEnt path[AZG_MAXD];
    const int t = blockIdx.x;
    const int l = lane_id();
    TreeHdr H = load_uniform(&F.hdr[t]);
    if (H.status != ST_WAIT_NN) return;
    const uint32_t leaf = H.pending_leaf;
    NodeHdr* nhp = FR::nhdr(F, t, leaf);
    const int nv = nhp->nv;
    const RowLayout L(nv, F.U);
    uint8_t* row = FR::heap(F, t) + (size_t)nhp->row_off * 16u;
    const uint16_t* ids = (const uint16_t*)(row + L.offI);
    for (int i = l; i < G::A; i += 64) dense[i] = pi[(size_t)t * G::A + i];
    const int depth = (int)H.path_len;
    const PathEnt* gp = F.path + (size_t)t * AZG_MAXD;
    for (int d = l; d < depth; d += 64) path[d] = gp[d];
    wave_sync();
    const bool gen_noise = (root_noise == nullptr && noise_stride == -1 && F.dirichletAlpha != 0.0);
    const bool dir_now = (H.leaf_is_root && H.sim_idx == 0 && H.is_full && (root_noise != nullptr || gen_noise));
    if (dir_now) {                                                                               // MCTS.py:147-149
        if (l < G::AW) mask[l] = 0ull;
        wave_sync();
        if (l == 0)
            for (int j = 0; j < nv; j++) mask[ids[j] >> 6] |= 1ull << (ids[j] & 63);
        wave_sync();
        FR::root_noise_dense(dense, mask, F.temp_root,
                             root_noise ? root_noise + (size_t)t * (noise_stride < 0 ? -noise_stride : noise_stride) : nullptr,
                             root_noise != nullptr && noise_stride < 0, F.dirichletAlpha,
                             mix64(mix64(F.rng_seed ^ 0xA5A5A5A55A5A5A5AULL) + F.stream0 + (uint64_t)t), H.c_sims << 20);
        float* Prow = (float*)row;
        for (int j = l; j < nv; j += 64) Prow[j] = dense[ids[j]];
    } else {
        const float s = np_sum_f32(dense, G::A);                                                 // normalise :150,250-253
        float* Prow = (float*)row;
        for (int j = l; j < nv; j += 64) Prow[j] = dense[ids[j]] / s;
    }
    uint32_t* Nrow = (uint32_t*)(row + L.offN);
    double* Qrow = (double*)(row + L.offQ);
    uint32_t* crow = (uint32_t*)(row + L.offC);
    for (int j = l; j < nv; j += 64) { Nrow[j] = 0u; Qrow[j] = AZG_NANQ; }                       // :40-41,152
    for (int j = l; j < (int)((L.offI - L.offC) / 4u); j += 64) crow[j] = AZG_NONE;     // whole aligned section
    float v[G::P];
#pragma unroll
    for (int p = 0; p < G::P; p++) v[p] = vin[(size_t)t * G::P + p];
    if (l == 0) { nhp->Ns = 0; nhp->Qs = v[0]; nhp->flags = NF_EXPANDED; }                        // :152-153
    FR::backup(F, t, path, depth, v);                                                            // leaf returns v :154
    if (l == 0) {
        TreeHdr* Hp = &F.hdr[t];
        Hp->sim_idx = H.sim_idx + 1;
        Hp->status = ST_SEARCHING;
        Hp->pending_leaf = AZG_NONE;
        Hp->c_exp = H.c_exp + 1;
        Hp->c_depth = H.c_depth + (uint64_t)depth;
    }
}

// MCTS.getActionProb epilogue (MCTS.py:67-103) into dense LDS buffers; returns false if the root has no policy.
// counts (after policy-target pruning) land in `cnt` (int), all lanes must call.
template <class G>
__device__ bool root_counts(const ForestDev& F, int t, const TreeHdr& H, int* cnt /*LDS [A]*/, float* q /*[P] regs*/) {
    using FR = Forest<G>;
    const int l = lane_id();
    for (int i = l; i < G::A; i += 64) cnt[i] = 0;
    wave_sync();
    if (H.root == AZG_NONE) return false;
    const NodeHdr nh = *FR::nhdr(F, t, H.root);
    const float q0 = nh.Qs;                                                                      // :71-72
#pragma unroll
    for (int p = 0; p < G::P; p++) q[p] = p == 0 ? q0 : -q0 / (float)(G::P - 1);
    if (!(nh.flags & NF_EXPANDED)) return false;
    const RowLayout L(nh.nv, F.U);
    const uint8_t* row = FR::heap(F, t) + (size_t)nh.row_off * 16u;
    const float* Prow = (const float*)row;
    const uint32_t* Nrow = (const uint32_t*)(row + L.offN);
    const uint16_t* ids = (const uint16_t*)(row + L.offI);
    int best = 0;
    for (int j = l; j < nh.nv; j += 64) { int n = (int)Nrow[j]; best = n > best ? n : best; }
    best = wave_max_i32(best);
    for (int j = l; j < nh.nv; j += 64) {
        int c = (int)Nrow[j];
        if (H.forced) {                                                                          // :75-80
            if (c != best) {
                float tq = (0.5f * Prow[j]) * (float)H.n_sims;       // plain-Python NumPy scalar typing (see oracle)
                c = c - (int)sqrt((double)tq);
            }
            c = c > 1 ? c : 0;
        }
        cnt[ids[j]] = c;
    }
    wave_sync();
    return true;
}

template <class G>
__global__ __launch_bounds__(64) void k_action_probs(ForestDev F, double temp, double* probs, float* qout,
                                                     uint8_t* is_full) {
    __shared__ int cnt[G::A];
    const int t = blockIdx.x;
    const int l = lane_id();
    const TreeHdr H = F.hdr[t];
    float q[G::P];
#pragma unroll
    for (int p = 0; p < G::P; p++) q[p] = 0.f;
    root_counts<G>(F, t, H, cnt, q);
    if (l == 0) {
        if (qout) for (int p = 0; p < G::P; p++) qout[(size_t)t * G::P + p] = q[p];
        if (is_full) is_full[t] = (uint8_t)H.is_full;
    }
    if (!probs) return;
    double* pr = probs + (size_t)t * G::A;
    if (temp <= 0.02) {                                                                          // :93-98
        int best = -1, ba = 0;
        for (int a = 0; a < G::A; a++) if (cnt[a] > best) { best = cnt[a]; ba = a; }
        for (int a = l; a < G::A; a += 64) pr[a] = a == ba ? 1.0 : 0.0;
        return;
    }
    double s = 0.0;                                                                              // :100-103 sequential sum
    const double e = 1.0 / temp;
    for (int a = 0; a < G::A; a++) s += (temp == 1.0) ? (double)cnt[a] : pow((double)cnt[a], e);
    for (int a = l; a < G::A; a += 64) pr[a] = ((temp == 1.0) ? (double)cnt[a] : pow((double)cnt[a], e)) / s;
}

template <class G>
__global__ __launch_bounds__(64) void k_root_stats(ForestDev F, int32_t* Ns, float* Qs, int32_t* Nsa, double* Qsa,
                                                   float* Ps, int32_t* n_nodes) {
    using FR = Forest<G>;
    const int t = blockIdx.x;
    const int l = lane_id();
    const TreeHdr H = F.hdr[t];
    if (n_nodes && l == 0) n_nodes[t] = (int32_t)H.n_nodes;
    for (int a = l; a < G::A; a += 64) {
        if (Nsa) Nsa[(size_t)t * G::A + a] = 0;
        if (Qsa) Qsa[(size_t)t * G::A + a] = AZG_NANQ;
        if (Ps) Ps[(size_t)t * G::A + a] = 0.f;
    }
    if (H.root == AZG_NONE) { if (l == 0) { if (Ns) Ns[t] = -1; if (Qs) Qs[t] = 0.f; } return; }
    const NodeHdr nh = *FR::nhdr(F, t, H.root);
    if (l == 0) { if (Ns) Ns[t] = (int32_t)nh.Ns; if (Qs) Qs[t] = nh.Qs; }
    if (!(nh.flags & NF_EXPANDED)) return;
    __syncthreads();
    const RowLayout L(nh.nv, F.U);
    const uint8_t* row = FR::heap(F, t) + (size_t)nh.row_off * 16u;
    const uint16_t* ids = (const uint16_t*)(row + L.offI);
    for (int j = l; j < nh.nv; j += 64) {
        int a = ids[j];
        if (Nsa) Nsa[(size_t)t * G::A + a] = (int32_t)((const uint32_t*)(row + L.offN))[j];
        if (Qsa) Qsa[(size_t)t * G::A + a] = ((const double*)(row + L.offQ))[j];
        if (Ps) Ps[(size_t)t * G::A + a] = ((const float*)row)[j];
    }
}

}  // namespace azg
