// kernels.hip.h -- __global__ kernels of the engine, templated on the device game (SplendorDev<n>, SantoriniDev<g>).
// One workgroup = one wavefront (64 threads) = one state (env kernels) or one tree (forest kernels).
#pragma once
#include <type_traits>
#include "forest.hip.h"

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
    G::valid_mask(st, players ? ld_agent_i32(players + t) : 0, mask);
    wave_sync();
    for (int a = lane_id(); a < G::A; a += 64) out[(size_t)t * G::A + a] = (uint8_t)((mask[a >> 6] >> (a & 63)) & 1);
}

template <class G>
__global__ __launch_bounds__(64) void k_env_next_state(const int8_t* states, const int32_t* players,
                                                       const int32_t* actions, const int64_t* seeds, int n,
                                                       int8_t* out_states, int32_t* out_next, uint64_t rng_seed,
                                                       uint64_t stream0, uint64_t* counters) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP + MoveScratch<G>::value];
    int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    Rng rng{rng_seed, stream0 + (uint64_t)t, counters ? ld_agent_u64(counters + t) : 0ull};
    const int np = G::wave_make_move(st, ld_agent_i32(actions + t), players ? ld_agent_i32(players + t) : 0,
                                     seeds ? (long long)ld_agent_u64((const uint64_t*)seeds + t) : 0ll, rng);
    if (lane_id() == 0) {
        if (counters) counters[t] = rng.counter;
        out_next[t] = np;
    }
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
    G::game_ended(st, next_players ? ld_agent_i32(next_players + t) : 0, es, mask);
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
    int p = ld_agent_i32(players + t);
    if (p != 0) G::swap_players(st, tmp, p);
    Forest<G>::store_state_unpadded(out_states + (size_t)t * G::S, st);
}

// Game.getSymmetries (Game.py:96-109; Coach.py:66-69 applies it to every recorded example): one wave per input triple,
// form k of (state, pi, valids) goes to row k of the outputs [n][max_sym][.]; out_count[t] = number of forms written.
template <class G>
__global__ __launch_bounds__(64) void k_env_symmetries(const int8_t* states, const float* pi, const uint8_t* valids, int n,
                                                        int max_sym, int8_t* out_states, float* out_pi, uint8_t* out_valids,
                                                        int32_t* out_count) {
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    const int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    const float* pin = pi + (size_t)t * G::A;
    const uint8_t* vin = valids + (size_t)t * G::A;
    int k = 0;
    for (int c = 0; c < G::NSYM_CAND && k < max_sym; c++) {
        if (!G::sym_exists(st, c)) continue;
        const size_t o = (size_t)t * max_sym + k;
        for (int i = lane_id(); i < G::S; i += 64) out_states[o * G::S + i] = G::sym_state_byte(st, c, i);
        for (int a = lane_id(); a < G::A; a += 64) {
            const int src = G::sym_action_src(st, c, a);          // < 0: no action maps onto a (Abalone groups off the grid)
            out_pi[o * G::A + a] = src >= 0 ? pin[src] : 0.f;
            out_valids[o * G::A + a] = src >= 0 ? vin[src] : (uint8_t)0;
        }
        k++;
    }
    if (lane_id() == 0) out_count[t] = k;
}

// Games whose symmetric forms are BUILT by lane 0 (G::RANDOM_SYM): get_symmetries is itself random (The Little Prince shuffles
// players, market cards and planet slots and drops duplicate states) or is a set of card-level edits rather than a byte map
// (Botanik).  One wave per input triple; all lanes copy the state into the candidate buffer, lane 0 turns it into form c
// (G::sym_build: state edits + the action map out[a] = in[act_src[a]] (0 where act_src[a] < 0), drawing from the counter stream (rng_seed, stream0 + t) if the
// game shuffles), all lanes write the form out.  With G::SYM_DEDUP a form is kept only when its state differs from every form kept so
// far (the kept forms stay in LDS for the comparison).
template <class G>
__global__ __launch_bounds__(64) void k_env_symmetries_built(const int8_t* states, const float* pi, const uint8_t* valids, int n,
                                                              int max_sym, int8_t* out_states, float* out_pi, uint8_t* out_valids,
                                                              int32_t* out_count, uint64_t rng_seed, uint64_t stream0) {
    constexpr int NKEEP = G::SYM_DEDUP ? G::NSYM_CAND : 1;
    __shared__ __attribute__((aligned(16))) int8_t st[G::SP];
    __shared__ __attribute__((aligned(16))) int8_t kept[NKEEP][G::SP];
    __shared__ int16_t act_src[G::A];
    __shared__ int exists_s;
    const int t = blockIdx.x;
    if (t >= n) return;
    Forest<G>::load_state_unpadded(st, states + (size_t)t * G::S);
    const float* pin = pi + (size_t)t * G::A;
    const uint8_t* vin = valids + (size_t)t * G::A;
    Rng rng{rng_seed, stream0 + (uint64_t)t, 0};
    int k = 0;
    for (int c = 0; c < G::NSYM_CAND; c++) {
        int8_t* cand = kept[G::SYM_DEDUP ? (k < NKEEP ? k : NKEEP - 1) : 0];
        for (int i = lane_id(); i < G::S; i += 64) cand[i] = st[i];
        __syncthreads();
        if (lane_id() == 0) exists_s = G::sym_build(st, c, cand, act_src, rng, vin) ? 1 : 0;     // (draws even when the form is dropped below)
        __syncthreads();
        bool keep = exists_s != 0 && k < max_sym;
        if (keep && G::SYM_DEDUP) {
            for (int e = 0; e < k && keep; e++) {
                bool diff = false;
                for (int i = lane_id(); i < G::S; i += 64) diff |= kept[e][i] != cand[i];
                keep = __ballot(diff) != 0;
            }
        }
        if (keep) {
            const size_t o = (size_t)t * max_sym + k;
            for (int i = lane_id(); i < G::S; i += 64) out_states[o * G::S + i] = cand[i];
            for (int a = lane_id(); a < G::A; a += 64) {
                const int src = act_src[a];                          // < 0: nothing maps onto a
                out_pi[o * G::A + a] = src >= 0 ? pin[src] : 0.f;
                out_valids[o * G::A + a] = src >= 0 ? vin[src] : (uint8_t)0;
            }
            k++;
        }
        __syncthreads();
    }
    if (lane_id() == 0) out_count[t] = k;
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
    for (int i = lane_id(); i < Forest<G>::n_classes(F); i += 64) Forest<G>::rec_free(F, t)[i] = AZG_NONE;
    if (lane_id() == 0) {
        TreeHdr* H = &F.hdr[t];
        H->n_nodes = 0; H->heap_top = 0; H->root = AZG_NONE; H->root_rec = AZG_NONE; H->status = ST_IDLE;
        H->id_top = 0; H->n_free_ids = 0; H->free_units = 0;
        H->sim_idx = 0; H->n_sims = 0; H->pending_leaf = AZG_NONE; H->path_len = 0; H->mid_sim = 0;
    }
}

// Set up a search from the canonical state in LDS `sm.st` (MCTS.getActionProb prologue, MCTS.py:58-60).
template <class G>
__device__ void begin_search_from_lds(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm, bool full) {
    using FR = Forest<G>;
    FR::store_state(F.root_state + (size_t)t * G::SP, sm.st);
    uint64_t h = FR::hash_state(sm.st);
    uint32_t free_slot;
    uint32_t found_rec = AZG_NONE;
    H.root = FR::probe(F, t, sm.st, h, &free_slot, &found_rec);
    H.root_rec = H.root == AZG_NONE ? AZG_NONE : found_rec;
    H.root_round = (uint32_t)G::gc_age(sm.st);
    H.is_full = full ? 1u : 0u;
    H.n_sims = (uint32_t)(full ? F.numMCTSSims : F.numMCTSSims / F.ratio_fullMCTS);
    H.forced = (full && F.forced_playouts) ? 1u : 0u;
    H.sim_idx = 0;
    H.status = ST_SEARCHING;
    H.pending_leaf = AZG_NONE;
    H.mid_sim = 0;
    // MCTS.py:64,156-160: an EXISTING root gets its noise before simulation 0 (applied by k_root_noise)
    H.noise_pending = (full && F.dirichletAlpha != 0.0 && H.root_rec != AZG_NONE) ? 1u : 0u;
}

// Root Dirichlet noise (MCTS.py:64,147-149,156-160,187-197) as its own small kernel so that the f64 pow/log/cos of the
// Gamma sampler never weigh on the registers of the descent kernel.  It runs right before k_select and touches only trees
// with noise_pending != 0: an existing root at the start of a full search (entries hold the normalised prior), or a root
// that was expanded by simulation 0 (k_expand_backup left the RAW net output in the entries).  Both get the reference's
// sequence softmax(T) -> 0.75*P + 0.25*Dir -> normalise.
// Applies the noise to the root record `root_rec` of tree t (all lanes must call); returns false when there is nothing to do.
template <class G>
__device__ __forceinline__ bool root_noise_tree(const ForestDev& F, int t, uint32_t root_rec, uint64_t c_sims,
                                                const double* root_noise, int noise_stride, float* dense /*LDS [A]*/,
                                                uint64_t* mask /*LDS [AW]*/) {
    using FR = Forest<G>;
    const int l = lane_id();
    if (root_rec == AZG_NONE) return false;
    uint8_t* rec = FR::rec_ptr(F, t, root_rec);
    const RecHdr rh = load_uniform((const RecHdr*)rec);
    if (!(rh.flags & NF_EXPANDED)) return false;
    const int nv = rh.nv;

    const RecGeom RG = FR::geom(F);
    const RecIds ids(rec, RG);
    for (int i = l; i < G::A; i += 64) dense[i] = 0.f;
    if (l < G::AW) mask[l] = 0ull;
    wave_sync();
    for (int j = l; j < nv; j += 64) dense[ids[j]] = *(const float*)(rec + RG.hot((uint32_t)j) + AZG_H_P);
    if (l == 0)
        for (int j = 0; j < nv; j++) mask[ids[j] >> 6] |= 1ull << (ids[j] & 63);
    wave_sync();
    const double* nz = root_noise ? root_noise + (size_t)t * (noise_stride < 0 ? -noise_stride : noise_stride) : nullptr;
    FR::root_noise_dense(dense, mask, F.temp_root, nz, root_noise != nullptr && noise_stride < 0, F.dirichletAlpha,
                         mix64(mix64(forest_seed(F) ^ 0xA5A5A5A55A5A5A5AULL) + F.stream0 + (uint64_t)t), c_sims << 20);
    for (int j = l; j < nv; j += 64) *(float*)(rec + RG.hot((uint32_t)j) + AZG_H_P) = dense[ids[j]];
    return true;
}

template <class G>
__global__ __launch_bounds__(64) void k_root_noise(ForestDev F, const double* root_noise, int noise_stride) {
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    __shared__ __attribute__((aligned(16))) uint64_t mask[G::AW];
    const int t = blockIdx.x;
    if (!ld_agent_u32(&F.hdr[t].noise_pending)) return;
    const uint32_t root_rec = ld_agent_u32(&F.hdr[t].root_rec);
    const uint64_t c_sims = ld_agent_u64(&F.hdr[t].c_sims);
    if (root_noise_tree<G>(F, t, root_rec, c_sims, root_noise, noise_stride, dense, mask) && lane_id() == 0)
        F.hdr[t].noise_pending = 0u;
}

// PUCT score of one entry (pick_highest_UCB body, MCTS.py:222-225), f64 with the Numba operand typing.
__device__ __forceinline__ double ucb_score(float p, uint32_t n, double q, double cpuct, double sqrtNs, double sqrtNsEps,
                                            double fpu_init) {
    if (q != AZG_NANQ) return q + cpuct * (double)p * sqrtNs / (double)(1u + n);
    return fpu_init + cpuct * (double)p * sqrtNsEps;
}

// games whose terminal test has to scan the valid moves anyway leave the mask behind (G::ENDED_FILLS_MASK)
template <class G, class = void> struct EndedFillsMask { static constexpr bool value = false; };
template <class G> struct EndedFillsMask<G, std::void_t<decltype(G::ENDED_FILLS_MASK)>> { static constexpr bool value = G::ENDED_FILLS_MASK; };

// Frontier edge / new root: the state of a node that is not in the tree yet is in sm.st.  Creates the node and its record,
// runs the terminal test and the valid-move scan (MCTS.py:127-142).  Returns the record offset (AZG_NONE on overflow);
// *terminal tells whether the new node ended the game (then es[] holds Es).  For a non-terminal leaf the canonical state and
// valid mask are written to the net's leaf batch.
// ASYNC (azg_async.hip.h, the asynchronous tree pipeline): `leaf_states` is the pipeline's leaf-record array (AsyncLeaf<G>: padded state +
// valid bit mask, one record per tree) and the record is written WRITE-THROUGH (agent-scope stores), because the net workgroup that reads it
// runs on another CU, possibly another XCD, inside the same launch; the byte mask in leaf_valid is still written for this CU's own expansion.
template <class G>
struct AsyncLeaf {
    static constexpr int MASK_OFF = G::SP;                                   // state int8[SP] (zero tail), then the valid mask u64[AW]
    static constexpr int STRIDE = (G::SP + G::AW * 8 + 15) / 16 * 16;
};
template <class G, class HS, bool ASYNC = false>
__device__ __forceinline__ uint32_t create_leaf(const ForestDev& F, int t, HS& H, typename Forest<G>::Smem& sm,
                                             uint64_t h, uint32_t free_slot, int8_t* leaf_states, uint8_t* leaf_valid,
                                             bool* terminal, float* es, const typename Forest<G>::LeafPf* pf = nullptr) {
    using FR = Forest<G>;
    const int l = lane_id();
    const uint32_t id = FR::create_node(F, t, H, sm.st, h, free_slot, pf);
    if (id == AZG_NONE) return AZG_NONE;
    const bool ended = G::game_ended(sm.st, 0, es, sm.mask);                                     // MCTS.py:131
    int nv = 0;
    if (!ended) {
        if (!EndedFillsMask<G>::value) G::valid_mask(sm.st, 0, sm.mask);                         // MCTS.py:142
        wave_sync();
#pragma unroll
        for (int k = 0; k < G::AW; k++) nv += __popcll(sm.mask[k]);
    }
    H.leaf_nv = (uint32_t)nv; H.leaf_node = id;
    int alloc_cls = 0;
    const uint32_t rec_off = FR::alloc_record(F, t, H, nv, id, &alloc_cls, pf);
    if (rec_off == AZG_NONE) return AZG_NONE;
    uint8_t* rec = FR::rec_ptr(F, t, rec_off);
    const uint8_t round = (uint8_t)G::gc_age(sm.st);          // the node's age tag for the clean-up (NodeHdr / RecHdr .round)
    if (l == 0) {
        RecHdr rh;
        rh.Ns = 0; rh.Qs = 0.f; rh.node_id = id; rh.nv = (uint16_t)nv; rh.flags = ended ? NF_TERMINAL : 0; rh.round = round;
#pragma unroll
        for (int p = 0; p < AZG_MAX_PLAYERS_DEV; p++) rh.Es[p] = (ended && p < G::P) ? es[p] : 0.f;
        if (G::P > AZG_MAX_PLAYERS_DEV && ended) rh.Qs = es[G::P > AZG_MAX_PLAYERS_DEV ? AZG_MAX_PLAYERS_DEV : 0];
        if (!ended) { rh.sq[0] = 0.0; rh.sq[1] = sqrt(0.0 + AZG_EPS); }
        *(RecHdr*)rec = rh;
        NodeHdr* nh = FR::nhdr(F, t, id);
        nh->rec_off = rec_off; nh->nv = (uint16_t)alloc_cls; nh->round = round; nh->flags = rh.flags;
    }
    *terminal = ended;
    if (!ended) {
        // the entries themselves (action id included) are written by the expansion, in one pass, when the policy arrives
        for (int a = l; a < G::A; a += 64) leaf_valid[(size_t)t * G::A + a] = (uint8_t)((sm.mask[a >> 6] >> (a & 63)) & 1);
        if constexpr (ASYNC) {
            using AL = AsyncLeaf<G>;
            static_assert(G::SP % 8 == 0, "8-byte granules");
            unsigned long long* dst = (unsigned long long*)(leaf_states + (size_t)t * AL::STRIDE);
            const unsigned long long* src = (const unsigned long long*)sm.st;
            for (int i = l; i < G::SP / 8 + G::AW; i += 64)
                __hip_atomic_store(dst + i, i < G::SP / 8 ? src[i] : (unsigned long long)sm.mask[i - G::SP / 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
            FR::store_state_unpadded(leaf_states + (size_t)t * G::S, sm.st);
    }
    return rec_off;
}

// Resolve the child of (parent record, entry j, universe): replay the env step from the parent's state
// (get_next_best_action_and_canonical_state, MCTS.py:233-248), look the state up, create it if new.
// Returns child slot value (record offset | next_player << AZG_CHILD_NP_SHIFT) or AZG_NONE on overflow.
template <class G, class HS, bool ASYNC = false>
__device__ __forceinline__ uint32_t resolve_edge(const ForestDev& F, int t, HS& H, typename Forest<G>::Smem& sm,
                                              uint32_t parent_node, int a, long long seed, int8_t* leaf_states,
                                              uint8_t* leaf_valid, bool* is_new, bool* terminal, float* es, Rng& rng,
                                              bool have_state = false, uint32_t st0 = 0, uint32_t st1 = 0, uint32_t st2 = 0) {
    using FR = Forest<G>;
#define AZG_SEG(k, d) H.cyc_seg[k] += (uint32_t)(d)
    long long c0 = AZG_CLK();
    typename FR::LeafPf pf = FR::leaf_pf_begin(F, t, H);     // allocator state for a possible new leaf: lands during the env step
    if (have_state) {                       // the level already fetched the parent's state with its entries (one-class forests)
        uint32_t* dst = (uint32_t*)sm.st;
        dst[lane_id()] = st0;
        if (lane_id() + 64 < FR::SPW) dst[lane_id() + 64] = st1;
        if (FR::SPW > 128 && lane_id() + 128 < FR::SPW) dst[lane_id() + 128] = st2;
        wave_sync();
    } else
        FR::load_state(sm.st, FR::nstate(F, t, parent_node));
    long long c1 = AZG_CLK(); AZG_SEG(0, c1 - c0);
    const int np = G::wave_make_move(sm.st, a, 0, seed, rng);      // (rng is only drawn from by STOCHASTIC games)
    c0 = AZG_CLK(); AZG_SEG(1, c0 - c1);
    FR::leaf_pf_links(F, t, pf);
    if (np != 0) G::swap_players(sm.st, sm.tmp, np);
    const uint64_t h = FR::hash_state(sm.st);
    c1 = AZG_CLK(); AZG_SEG(2, c1 - c0);
    uint32_t free_slot;
    uint32_t found_rec = AZG_NONE;
    // (Round 4, measured and dropped: requesting the table's first round trip ahead and computing the new node's terminal test and valid
    // mask under it, and requesting the backup's statistics before the expansion's arithmetic -- both bit-identical, both -0.5 %: with
    // four waves per SIMD the waits they remove were already covered by the other waves' work.)
    const uint32_t found = FR::probe(F, t, sm.st, h, &free_slot, &found_rec);
    c0 = AZG_CLK(); AZG_SEG(3, c0 - c1);
    uint32_t crec;
    *is_new = false;
    if (found != AZG_NONE) crec = found_rec;
    else {
        const long long t_l = AZG_CLK();
        crec = create_leaf<G, HS, ASYNC>(F, t, H, sm, h, free_slot, leaf_states, leaf_valid, terminal, es, &pf);
        H.cyc_leaf += (uint32_t)(AZG_CLK() - t_l);
        if (crec == AZG_NONE) return AZG_NONE;
        *is_new = true;
    }
    return crec | ((uint32_t)np << AZG_CHILD_NP_SHIFT);
}

__device__ __forceinline__ void stat_add(uint64_t* p, uint64_t v) {
    __hip_atomic_fetch_add((unsigned long long*)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef AZG_CYC_COUNTERS
// debug: ordered clock stamps through the prologue (asm volatile + memory clobber: memory operations cannot move across)
static __device__ unsigned long long g_prolog[16];
#define AZG_STAMP(k) do { long long c_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_)::"memory"); azg_stamp[k] = c_; } while (0)
#else
#define AZG_STAMP(k)
#endif

// One lock-step round, part 2: store (Ps, v) on the pending leaf and back up (MCTS.py:144-154,176-183).
// Split in a load half (everything whose address does not depend on loaded data: header words, pi, v, the valid mask the
// descent wrote for this leaf, the first 64 path entries -- ONE memory round trip) and an apply half, so that k_select can
// run it in its own prologue (self-play: the expansion of round r rides on the descent launch of round r+1).
#ifndef AZG_PATH_PF
#define AZG_PATH_PF 16       /* path entries that ride on the first round trip of a launch */
#endif
template <class G>
struct ExpandIn {
    static constexpr int NA = (G::A + 63) / 64;
    uint32_t status, pending_leaf, path_len, leaf_is_root, sim_idx, is_full, pend_nv, pend_node;
    float pv[NA];
    uint8_t va[NA];
    float v[G::P];
    PathEnt pe0;
};

// The hot words of a tree header (everything up to the statistics counters), fetched as ONE batch of wide scalar loads and
// pinned in SGPRs: left to itself the compiler turns the individual field reads into scalar loads that it issues -- and
// waits for -- group by group along the prologue (4 serialised round trips measured).
constexpr int AZG_HOT_WORDS = offsetof(TreeHdr, c_sims) / 4;
#define AZG_HW(w, field) ((w)[offsetof(TreeHdr, field) / 4])
// HOT (k_select): plain vector loads that merge into dwordx4 -- the kernel's own header stores keep the compiler from using
// the scalar path (checked in the ISA: no s_load of the header); elsewhere agent-scope atomic loads (forest.hip.h ld_agent_u32).
template <bool HOT>
__device__ __forceinline__ void load_hot_header(const TreeHdr* Hp, uint32_t (&w)[AZG_HOT_WORDS]) {
    const uint32_t* p = (const uint32_t*)Hp;
#pragma unroll
    for (int i = 0; i < AZG_HOT_WORDS; i++) w[i] = HOT ? p[i] : __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// second half, after every other load of the batch has been issued: one wait, the words move to SGPRs
__device__ __forceinline__ void pin_hot_header(uint32_t (&w)[AZG_HOT_WORDS]) {
#pragma unroll
    for (int i = 0; i < AZG_HOT_WORDS; i++) { w[i] = uni_u32(w[i]); asm volatile("" : "+s"(w[i])); }
}

// Touch every 64-byte line of the kernel's explicit arguments in ONE scalar batch at kernel entry: with ~100 SGPRs worth of
// ForestDev fields the compiler re-loads them from the segment wherever they are needed, and each first touch of a line
// would otherwise be a serialised scalar-cache miss in the middle of the prologue.
__device__ __forceinline__ void warm_kernarg_448() {
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t a, b, c, d, e, f, g;
    asm volatile("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %7, 0x40\n\ts_load_dword %2, %7, 0x80\n\t"
                 "s_load_dword %3, %7, 0xc0\n\ts_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\t"
                 "s_load_dword %6, %7, 0x180\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e), "=&s"(f), "=&s"(g)
                 : "s"(ka)
                 : "memory");
}

// ASYNC: the policy was written by a net workgroup on another CU inside this launch (write-through) -- read it past this CU's L1
template <class G, bool ASYNC = false>
__device__ __forceinline__ void expand_load(const ForestDev& F, int t, const float* pi, const float* vin,
                                            const uint8_t* leaf_valid, ExpandIn<G>& in, const uint32_t (&hw)[AZG_HOT_WORDS]) {
    const int l = lane_id();
    (void)hw;
#pragma unroll
    for (int k = 0; k < ExpandIn<G>::NA; k++) {
        const int a = l + 64 * k;
        if constexpr (ASYNC)
            in.pv[k] = a < G::A ? __uint_as_float(__hip_atomic_load((const uint32_t*)pi + (size_t)t * G::A + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.f;
        else
        in.pv[k] = a < G::A ? pi[(size_t)t * G::A + a] : 0.f;
        in.va[k] = a < G::A ? leaf_valid[(size_t)t * G::A + a] : (uint8_t)0;
    }
#pragma unroll
    for (int p = 0; p < G::P; p++)     // (the net's output of the previous launch: never through the scalar cache)
        in.v[p] = __uint_as_float(__hip_atomic_load((const uint32_t*)vin + (size_t)t * G::P + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // (the first round trip of a launch is requested by all T waves at once and is bandwidth-bound: only the path entries almost every
    // descent has ride on it -- 16 of them, a descent has 4.8 levels on average; a deeper path is completed in expand_apply)
    in.pe0 = l < AZG_PATH_PF ? (F.path + (size_t)t * AZG_MAXD)[l] : PathEnt{0u, 0, 0, 0};
}

template <class G>
__device__ __forceinline__ void expand_header(ExpandIn<G>& in, const uint32_t (&hw)[AZG_HOT_WORDS]) {
    in.status = AZG_HW(hw, status); in.pending_leaf = AZG_HW(hw, pending_leaf); in.path_len = AZG_HW(hw, path_len);
    in.leaf_is_root = AZG_HW(hw, leaf_is_root); in.sim_idx = AZG_HW(hw, sim_idx); in.is_full = AZG_HW(hw, is_full);
    in.pend_nv = AZG_HW(hw, pending_nv); in.pend_node = AZG_HW(hw, pending_node);
}

// returns true when the leaf was a fresh root that still needs its Dirichlet noise (noise_pending = 2 was set)
template <class G>
__device__ __forceinline__ bool expand_apply(const ForestDev& F, int t, const ExpandIn<G>& in, float* dense /*LDS [A]*/,
                                             PathEnt* path /*LDS [AZG_MAXD]*/, int noise_enabled, long long* azg_stamp) {
    using FR = Forest<G>;
    constexpr int NA = ExpandIn<G>::NA;
    const int l = lane_id();
    TreeHdr* Hp = &F.hdr[t];
    const int depth = (int)uni_u32(in.path_len);
    const int nv = (int)uni_u32(in.pend_nv);
    const uint32_t sim = uni_u32(in.sim_idx);
    uint8_t* rec = FR::rec_ptr(F, t, uni_u32(in.pending_leaf));
    RecHdr* rhp = (RecHdr*)rec;
    const RecGeom RG = FR::geom(F);
    const PathEnt* gp = F.path + (size_t)t * AZG_MAXD;
    if (l < AZG_PATH_PF) path[l] = in.pe0;
    for (int d = l + AZG_PATH_PF; d < depth; d += 64) path[d] = gp[d];            // (a path deeper than the prefetched entries: one more round trip)
    constexpr bool SUM_IN_REGS = G::A >= 8 && G::A <= 256;      // NumPy's pairwise order on the registers the policy was loaded into
    if (!SUM_IN_REGS) {
#pragma unroll
        for (int k = 0; k < NA; k++) if (l + 64 * k < G::A) dense[l + 64 * k] = in.pv[k];
    }
    wave_sync();
    // a root expanded by simulation 0 of a full search gets root noise (MCTS.py:147-149): keep the RAW net output in the
    // entries and let the noise step do softmax -> noise -> normalise; every other leaf is normalised here (:150,250-253)
    const bool dir_now = (noise_enabled && uni_u32(in.leaf_is_root) && sim == 0 && uni_u32(in.is_full) && F.dirichletAlpha != 0.0);
    float s = 1.f;
    if (!dir_now) {
        if constexpr (SUM_IN_REGS) s = np_sum_regs<G::A>(in.pv);
        else s = np_sum_f32_static<G::A>(dense);
    }
    AZG_STAMP(2);
    // entry j belongs to the j-th valid action (the rank of its bit in the leaf's valid mask): no read of the record needed
    int base_rank = 0;
#pragma unroll
    for (int k = 0; k < NA; k++) {                                                               // :40-41,150-152
        const uint64_t m = __ballot(in.va[k] != 0);
        if (in.va[k]) {
            const uint32_t j = (uint32_t)(base_rank + __popcll(m & ((1ull << l) - 1ull)));
            uint4 e0;
            e0.x = __float_as_uint(dir_now ? in.pv[k] : in.pv[k] / s); e0.y = 0u;                 // P, N = 0
            e0.z = (uint32_t)__double_as_longlong(AZG_NANQ); e0.w = (uint32_t)((uint64_t)__double_as_longlong(AZG_NANQ) >> 32);
            *(uint4*)(rec + RG.hot(j)) = e0;
            const int nu = (j % RG.PC) < RG.SE ? F.U : 1;                                         // (entries beyond SE share slot 0)
            for (int u = 0; u < nu; u++) *(uint32_t*)(rec + RG.child(j, (uint32_t)u)) = AZG_NONE;
            if (FR::IDB == 1) *(uint8_t*)(rec + RG.id(j)) = (uint8_t)(l + 64 * k);               // the entry's action id
            else *(uint16_t*)(rec + RG.id(j)) = (uint16_t)(l + 64 * k);
        }
        base_rank += __popcll(m);
    }
    if (l == 0) {                                                                                // :152-153
        rhp->Ns = 0; rhp->Qs = in.v[0]; rhp->flags = NF_EXPANDED;
        FR::nhdr(F, t, uni_u32(in.pend_node))->flags = NF_EXPANDED;
    }
    AZG_STAMP(3);
    FR::backup(F, t, path, depth, in.v);                                                         // leaf returns v :154
    AZG_STAMP(4);
    if (l == 0) {
        Hp->sim_idx = sim + 1;
        Hp->status = ST_SEARCHING;
        Hp->pending_leaf = AZG_NONE;
        stat_add(&Hp->c_exp, 1ull);
        stat_add(&Hp->c_depth, (uint64_t)depth);
        if (dir_now) Hp->noise_pending = 2u;
    }
    wave_sync();
    return dir_now;
}

template <class G>
__global__ __launch_bounds__(64) void k_expand_backup(ForestDev F, const float* pi, const float* vin, const uint8_t* leaf_valid,
                                                      int noise_enabled) {
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    __shared__ __attribute__((aligned(16))) PathEnt path[AZG_MAXD];
    const int t = blockIdx.x;
    ExpandIn<G> in;
    uint32_t hw[AZG_HOT_WORDS];
    load_hot_header<false>(&F.hdr[t], hw);
    expand_load<G>(F, t, pi, vin, leaf_valid, in, hw);
    pin_hot_header(hw);
    expand_header<G>(in, hw);
    if (uni_u32(in.status) != ST_WAIT_NN) return;
    long long azg_stamp[8];
    expand_apply<G>(F, t, in, dense, path, noise_enabled, azg_stamp);
    (void)azg_stamp;
}


// The fields of TreeHdr that k_select keeps live (wave-uniform => SGPRs).  Loading the whole 200-byte header into
// registers made the kernel spill to scratch; the cold fields are read-modify-written by lane 0 at the end instead.
struct SelState {
    uint32_t n_nodes, heap_top, id_top, n_free_ids, free_units, root, root_rec, sim_idx, n_sims, is_full, forced, err, leaf_is_root, mid_sim, cur_rec,
        cur_depth, cur_pre, status, pending_leaf, path_len, leaf_nv, leaf_node, cyc_leaf, cyc_seg[4];
    uint32_t rng_lo, rng_hi;        // the tree's RNG counter (STOCHASTIC games draw the env randomness of every simulated step)
};

// One lock-step round, part 1 (MCTS.search descent, MCTS.py:105-175) for tree t, run by ONE wavefront: the body of k_select (one
// single-wave workgroup per tree) and of the select phase of k_rounds_v80 (azg_fused.hip.h: sixteen trees per workgroup, each wave with
// its own Smem / dense block in the workgroup's LDS; wave_sync() is then wave-local).  pi != nullptr: the expansion + backup of the
// previous round's leaf first (its loads ride on the header's round trip).
// Returns 1 when the tree handed a leaf to the net (status ST_WAIT_NN), 2 when the level / work budget parked its descent (it goes on at the
// next call), 0 otherwise (search finished, idle, waiting for its root noise).  ASYNC: the asynchronous pipeline's hand-over forms
// (create_leaf, expand_load).
template <class G, bool ASYNC = false>
__device__ __forceinline__ int select_tree(const ForestDev& F, const int t, typename Forest<G>::Smem& sm, float* dense /* LDS [A]: fused expansion only */,
                                            int8_t* leaf_states, uint8_t* leaf_valid, uint8_t* needs_eval, int wait_noise, const float* pi,
                                            const float* vin, int noise_enabled) {
    using FR = Forest<G>;
    long long azg_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    AZG_STAMP(0);
    const long long t_first = AZG_CLK();
    const int l = lane_id();
    TreeHdr* Hp = &F.hdr[t];
#ifdef AZG_WALL_CAL
    const uint32_t w_first = (uint32_t)wall_clock64();
#endif
    // every hot header field is requested in one go (status included): one memory round trip before the first level
    SelState H;
    ExpandIn<G> ein;
    uint32_t hw[AZG_HOT_WORDS];
    load_hot_header<true>(Hp, hw);
    if (pi) expand_load<G, ASYNC>(F, t, pi, vin, leaf_valid, ein, hw);  // pi != nullptr: the previous round's expansion first
#ifdef AZG_PIN_HEADER
    pin_hot_header(hw);
#endif
    AZG_STAMP(1);
    expand_header<G>(ein, hw);
    uint32_t status0 = AZG_HW(hw, status);
    const uint32_t pending0 = AZG_HW(hw, noise_pending);
    H.id_top = AZG_HW(hw, id_top); H.n_free_ids = AZG_HW(hw, n_free_ids); H.free_units = AZG_HW(hw, free_units);
    H.n_nodes = AZG_HW(hw, n_nodes); H.heap_top = AZG_HW(hw, heap_top); H.root = AZG_HW(hw, root); H.root_rec = AZG_HW(hw, root_rec);
    H.sim_idx = AZG_HW(hw, sim_idx); H.n_sims = AZG_HW(hw, n_sims); H.is_full = AZG_HW(hw, is_full); H.forced = AZG_HW(hw, forced);
    H.err = AZG_HW(hw, err); H.leaf_is_root = AZG_HW(hw, leaf_is_root); H.mid_sim = AZG_HW(hw, mid_sim); H.cur_rec = AZG_HW(hw, cur_rec);
    H.cur_depth = AZG_HW(hw, cur_depth); H.cur_pre = AZG_HW(hw, cur_pre);
    H.rng_lo = hw[offsetof(TreeHdr, rng_counter) / 4]; H.rng_hi = hw[offsetof(TreeHdr, rng_counter) / 4 + 1];
    bool fresh_noise = false;
    // (Round 4, measured and dropped: touching the lines of the previous simulation's path records -- one load instruction, ten lanes per
    // level, issued here -- so that the descent finds them in the L2: the first round trip of a launch is requested by all T waves at
    // the same moment and is bandwidth-bound, the extra lines cost more than the hits give back: -1.6 % env-steps/s.)
    if (pi && uni_u32(status0) == ST_WAIT_NN) {
        fresh_noise = expand_apply<G>(F, t, ein, dense, sm.path, noise_enabled, azg_stamp);
#ifdef AZG_STAMP_DRAIN
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // debug: charge the drain of the expansion's stores to stamp 4 -> 5
#endif
        AZG_STAMP(5);
        status0 = ST_SEARCHING;
    }
    const bool expanded_here = pi && uni_u32(ein.status) == ST_WAIT_NN;
    // wait_noise: the root noise is applied by the periodic k_selfplay_advance launch; until then the tree sits out
    if (uni_u32(status0) != ST_SEARCHING || (wait_noise && (uni_u32(pending0) || fresh_noise))) {
        if (l == 0) needs_eval[t] = 0;
        return 0;
    }
    AZG_STAMP(7);
#ifdef AZG_PAD_CODE
    {   // experiment: AZG_PAD_CODE straight-line scalar instructions (4 bytes each) executed once per launch -- what does cold code cost?
        uint32_t pad_x = 0;
        asm volatile(".rept %1\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(pad_x) : "n"(AZG_PAD_CODE) : "scc");
        if (pad_x == 0xFFFFFFFFu) needs_eval[t] = 3;
    }
#endif
#ifdef AZG_PIN_HEADER
    H.id_top = uni_u32(H.id_top); H.n_free_ids = uni_u32(H.n_free_ids); H.free_units = uni_u32(H.free_units);
    H.n_nodes = uni_u32(H.n_nodes); H.heap_top = uni_u32(H.heap_top); H.root = uni_u32(H.root);
    H.err = uni_u32(H.err); H.leaf_is_root = uni_u32(H.leaf_is_root); H.cur_rec = uni_u32(H.cur_rec);
    H.cur_depth = uni_u32(H.cur_depth); H.cur_pre = uni_u32(H.cur_pre);
#endif
    // the allocator state, the error flags and the parked-descent state are touched once per simulation at most: they stay in VGPRs
    // (every lane holds the same value) instead of taking scalar registers from the descent loop (k_select spilled 547 SGPRs)
    H.root_rec = uni_u32(H.root_rec); H.sim_idx = uni_u32(H.sim_idx); H.n_sims = uni_u32(H.n_sims);
    H.is_full = uni_u32(H.is_full); H.forced = uni_u32(H.forced); H.mid_sim = uni_u32(H.mid_sim); H.err = uni_u32(H.err);
    Rng srng{F.rng_seed, F.stream0 + (uint64_t)t, ((uint64_t)uni_u32(H.rng_hi) << 32) | uni_u32(H.rng_lo)};
    if (expanded_here) H.sim_idx = uni_u32(ein.sim_idx) + 1u;          // the header words were requested before the expansion
    H.status = ST_SEARCHING; H.pending_leaf = AZG_NONE; H.path_len = 0; H.cyc_leaf = 0; H.leaf_nv = 0; H.leaf_node = 0;
    H.cyc_seg[0] = H.cyc_seg[1] = H.cyc_seg[2] = H.cyc_seg[3] = 0;
    uint8_t* hp = FR::heap(F, t);
    // this lane's byte offsets inside a record (entry l of the first 64): constant over the launch.  One-class forests have a single
    // page of A entries, the others pages of AZG_CLS_Q_MULTI = 32 entries (forest.hip.h RecGeom)
    const RecGeom RG = FR::geom(F);
    constexpr bool one_page = FR::ONE_CLASS;
    auto hot_off = [&](uint32_t j) { return one_page ? AZG_REC_HDR + j * 16u : AZG_REC_HDR + (j >> 5) * RG.PAGE + (j & 31u) * 16u; };
    auto child_off = [&](uint32_t j, uint32_t u) {                                      // == RG.child(j, u) with the page split spelled out
        const uint32_t pg = one_page ? AZG_REC_HDR : AZG_REC_HDR + (j >> 5) * RG.PAGE, jj = one_page ? j : (j & 31u);
        return (u == 0u || jj >= RG.SE) ? pg + RG.PC * 16u + jj * 4u : pg + RG.PC * 20u + ((u - 1u) * RG.SE + jj) * 4u;
    };
    auto id_off = [&](uint32_t j) {
        const uint32_t pg = one_page ? AZG_REC_HDR : AZG_REC_HDR + (j >> 5) * RG.PAGE, jj = one_page ? j : (j & 31u);
        return pg + RG.PC * 20u + (RG.U - 1u) * RG.SE * 4u + jj * RG.IDB;
    };
    const uint32_t l_hot = hot_off((uint32_t)l), l_id = id_off((uint32_t)l);
    // this lane's child slot of entry l: universe 0's, and -- for the entries that have per-universe slots -- where universe u's is
    const uint32_t l_child0 = child_off((uint32_t)l, 0u);
    const bool l_seeded = (one_page ? (uint32_t)l : ((uint32_t)l & 31u)) < RG.SE;
    const uint32_t l_childx = child_off((uint32_t)l, 1u), childx_ustride = RG.SE * 4u;      // (u >= 1: l_childx + (u - 1) * stride)
    // F.spec_state: 0 = never, 1 = at every level, N >= 2 = only at nodes reached over an edge with fewer than N visits (a node
    // with few visits is where the descent meets the frontier; the much-visited top of the tree almost never is, and its 3 x 128-B
    // state lines per level were a quarter of the kernel's fetches)
    const bool spec_state = F.spec_state && FR::ONE_CLASS && FR::SPW <= 192;
    const uint32_t spec_below = F.spec_state >= 2 ? (uint32_t)F.spec_state : 0xFFFFFFFFu;
    const float inv_units1 = 1.0f / (float)FR::cls_units(F, 1);
    bool need_nn = false;
    uint32_t c_sims = 0, c_levels = 0, c_sumvalid = 0, c_term = 0, levels_this_launch = 0, edges_this_launch = 0, work_units = 0;
    AZG_STAMP(6);
    const long long t_start = AZG_CLK();
#ifdef AZG_WALL_CAL
    const long long w_start = wall_clock64();
#endif
    long long cyc_levels = 0, cyc_edge = 0;
    while (true) {
        if (H.sim_idx >= H.n_sims || uni_u32(H.err)) { H.status = ST_DONE; break; }
        const int uidx = F.universes > 0 ? (int)(H.sim_idx % (uint32_t)F.universes) : 0;
        const long long seed = F.universes > 0 ? AZG_MAGIC_SEEDS[uidx] : -1ll;                 // MCTS.py:63
        int depth = 0, pre = 0;
        bool have_leaf = false, leaf_terminal = false;
        float es[G::P];
        uint32_t rec;
        uint32_t n_in = H.sim_idx;                           // visits of the edge the descent came over (root: simulations so far)
        if (H.mid_sim) {
            // resume a descent that the level budget paused in an earlier launch
            H.mid_sim = 0;
            rec = uni_u32(H.cur_rec); depth = (int)uni_u32(H.cur_depth); pre = (int)uni_u32(H.cur_pre);
            const PathEnt* gp0 = F.path + (size_t)t * AZG_MAXD;
            for (int d = l; d < depth; d += 64) sm.path[d] = gp0[d];
            wave_sync();
        } else {
            c_sims++;
            H.leaf_is_root = 0;
            rec = H.root_rec;
            if (rec == AZG_NONE) {
                // the root itself is not a node yet: it is the leaf of this simulation (MCTS.py:140-154)
                FR::load_state(sm.st, F.root_state + (size_t)t * G::SP);
                const uint64_t h = FR::hash_state(sm.st);
                uint32_t free_slot;
                uint32_t found_rec = AZG_NONE;
                const uint32_t found = FR::probe(F, t, sm.st, h, &free_slot, &found_rec);
                if (found == AZG_NONE) {
                    rec = uni_u32(create_leaf<G, SelState, ASYNC>(F, t, H, sm, h, free_slot, leaf_states, leaf_valid, &leaf_terminal, es));
                    if (rec == AZG_NONE) continue;
                    H.root = uni_u32(((const RecHdr*)(hp + (size_t)rec * 16u))->node_id);
                    H.root_rec = rec;
                    H.leaf_is_root = 1;
                    have_leaf = true;
                } else {
                    H.root = found;
                    rec = H.root_rec = found_rec;
                }
            }
        }
        bool paused = false, park_request = false;
        while (!have_leaf) {
            if ((F.level_budget > 0 && levels_this_launch >= (uint32_t)F.level_budget) ||
                (F.work_budget > 0 && (work_units >= (uint32_t)F.work_budget || park_request))) {
                // time slice: park the descent, resume next launch (the tree contributes no leaf this round)
                PathEnt* gp0 = F.path + (size_t)t * AZG_MAXD;
                for (int d = l; d < depth; d += 64) gp0[d] = sm.path[d];
                H.mid_sim = 1; H.cur_rec = rec; H.cur_depth = (uint32_t)depth; H.cur_pre = (uint32_t)pre;
                paused = true;
                break;
            }
            levels_this_launch++;
            work_units++;
            const long long t_lvl = AZG_CLK();
            // ---- one level: header + this lane's entry requested together (entry position is independent of nv) ----
            const uint8_t* rp = hp + (size_t)rec * 16u;
            const uint4 hot0 = *(const uint4*)(rp + l_hot);                                     // { P, N, Q } of entry l
            const uint32_t ch0 = *(const uint32_t*)(rp + ((uidx > 0 && l_seeded) ? l_childx + childx_ustride * (uint32_t)(uidx - 1) : l_child0));
            const uint32_t id0 = FR::IDB == 1 ? (uint32_t)*(const uint8_t*)(rp + l_id) : (uint32_t)*(const uint16_t*)(rp + l_id);
            const uint2 pn = make_uint2(hot0.x, hot0.y);
            const double q0 = __longlong_as_double((long long)(((uint64_t)hot0.w << 32) | hot0.z));
            // one-class forests: record slot == node id, so the node's state is addressable before its header arrives --
            // fetch it with the entries (this level is the frontier of ~20 % of the descents; saves that round trip)
            uint32_t ps0 = 0, ps1 = 0, ps2 = 0;
            const bool spec_now = spec_state && n_in < spec_below;
            if (spec_now) {
                const uint32_t nid = (uint32_t)((float)rec * inv_units1 + 0.5f);
                const uint32_t* nsp = (const uint32_t*)FR::nstate(F, t, nid);
                ps0 = nsp[l];
                if (l + 64 < FR::SPW) ps1 = nsp[l + 64];
                if (FR::SPW > 128 && l + 128 < FR::SPW) ps2 = nsp[l + 128];
            }
            const RecHdr rh = load_uniform_hot((const RecHdr*)rp);
            if (rh.flags & NF_TERMINAL) {                                                       // MCTS.py:136-138
                c_term++;
                float v[G::P];
#pragma unroll
                for (int p = 0; p < G::P; p++) v[p] = rec_es(rh, p);
                FR::backup(F, t, sm.path, depth, v);
                H.sim_idx++;
                break;
            }
            const int nv = rh.nv;
            // ---- pick_highest_UCB (MCTS.py:210-230) ----
            const bool forced = depth == 0 && H.forced;
            const double sqrtNs = rh.sq[0], sqrtNsEps = rh.sq[1];        // == sqrt(Ns), sqrt(Ns + EPS): see RecHdr
            const double fpu_init = F.fpu > 0 ? (double)rh.Qs - F.fpu : F.fpu;
            int j = -1, a_sel = 0;
            uint32_t child = AZG_NONE;
            {
                float p = __uint_as_float(pn.x);
                uint32_t n = pn.y;
                double q = q0;
                const bool act = l < nv;
                double best_u = -INFINITY;
                int best_j = 0x7FFFFFFF;
                uint32_t best_ch = AZG_NONE, best_id = 0, best_n = 0;
                for (int base = 0; base < nv; base += 64) {
                    uint32_t chv = ch0, idv = id0;
                    bool a_ok = act;
                    if (base > 0) {
                        const int jj = base + l;
                        a_ok = jj < nv;
                        const uint32_t j2 = (uint32_t)(a_ok ? jj : 0);
                        const uint4 h2 = *(const uint4*)(rp + hot_off(j2));
                        p = __uint_as_float(h2.x);
                        n = h2.y;
                        q = __longlong_as_double((long long)(((uint64_t)h2.w << 32) | h2.z));
                        chv = *(const uint32_t*)(rp + child_off(j2, (uint32_t)uidx));
                        idv = FR::IDB == 1 ? (uint32_t)*(const uint8_t*)(rp + id_off(j2)) : (uint32_t)*(const uint16_t*)(rp + id_off(j2));
                    }
                    if (forced) {                                                 // :218-220 first deficient action wins
                        // The reference's test is  Nsa < int(sqrt(0.5 * P * n_iter))  -- an f64 square root per lane and simulation (a ~35-
                        // instruction software sequence, v_rsq_f64 at a quarter of the rate).  It is decided without one: for an integer
                        // n >= 0,  n < trunc(RN(sqrt(x)))  <=>  m = n + 1 <= RN(sqrt(x))  <=>  m * m <= x.  (<=) sqrt(x) >= m and rounding is
                        // monotone.  (=>) otherwise sqrt(x) < m rounds UP to m, i.e. m - sqrt(x) <= m 2^-53 and m^2 - x <= m^2 2^-52; but
                        // x = 0.5 * p * k is exact (24 + 24 bits) and a multiple of q = 2^(e - 24) for p in [2^e, 2^(e + 1)), so is m^2
                        // (e <= 0), hence m^2 - x >= q, while m^2 < 2 x = p k < 2^(e + 1) k gives m^2 2^-52 < 2^(e - 51) k <= q for
                        // k < 2^24 (azg_forest_create checks numMCTSSims): a contradiction.  m * m is exact (m <= 2^24).
#ifndef AZG_FORCED_SQRT
                        const double mm = (double)(n + 1u);
                        const uint64_t def = __ballot(a_ok && (mm * mm <= 0.5 * (double)p * (double)H.sim_idx));
#else
                        const double thr = sqrt(0.5 * (double)p * (double)H.sim_idx);
                        const uint64_t def = __ballot(a_ok && ((long long)n < (long long)thr));
#endif
                        if (def) {
                            const int src = first_lane(def);
                            j = base + src;
                            child = (uint32_t)__builtin_amdgcn_readlane((int)chv, src);
                            a_sel = __builtin_amdgcn_readlane((int)idv, src);
                            n_in = (uint32_t)__builtin_amdgcn_readlane((int)n, src);
                            break;
                        }
                    }
                    double u = ucb_score(p, n, q, F.cpuct, sqrtNs, sqrtNsEps, fpu_init);
                    if (!a_ok) u = -INFINITY;
                    const int jj = a_ok ? base + l : 0x7FFFFFFF;
                    const bool take = (u > best_u) || (u == best_u && jj < best_j);
                    best_u = take ? u : best_u;
                    best_j = take ? jj : best_j;
                    best_ch = take ? chv : best_ch;
                    best_id = take ? idv : best_id;
                    best_n = take ? n : best_n;
                }
                if (j < 0) {
                    // wave arg-max, lowest index on ties (== ascending scan with strict '>', MCTS.py:216-228): reduce the
                    // f64 maximum, then the FIRST lane (lowest entry index of its chunk) holding it wins; across chunks the
                    // per-lane running best already prefers the earlier chunk on ties.
                    const double mx = wave_max_f64(best_u);
                    const uint64_t hit = __ballot(best_u == mx && best_j != 0x7FFFFFFF);
                    // several lanes can tie with different chunks' indices only when nv > 64: pick the lowest index
                    int src = first_lane(hit);
                    if (nv > 64) {
                        int cand = (best_u == mx) ? best_j : 0x7FFFFFFF;
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(cand, m, 64); cand = o < cand ? o : cand; }
                        src = uni_i32(cand) & 63;
                    }
                    j = __builtin_amdgcn_readlane(best_j, src);
                    child = (uint32_t)__builtin_amdgcn_readlane((int)best_ch, src);
                    a_sel = __builtin_amdgcn_readlane((int)best_id, src);
                    n_in = (uint32_t)__builtin_amdgcn_readlane((int)best_n, src);
                }
            }
            c_levels++;
            c_sumvalid += (uint32_t)nv;
            cyc_levels += AZG_CLK() - t_lvl;
            if (depth >= AZG_MAXD - 1) { H.err |= ERR_DEPTH_OVERFLOW; H.sim_idx = H.n_sims; break; }
            if (child == AZG_NONE) {
                if (F.work_budget > 0 && edges_this_launch > 0 && work_units + AZG_EDGE_UNITS > (uint32_t)F.work_budget) {
                    park_request = true;       // re-evaluated (same choice) when the descent resumes; the first edge of a
                    continue;                  // launch is always resolved, so every launch makes progress
                }
                edges_this_launch++;
                work_units += AZG_EDGE_UNITS;
                const int a = a_sel;
                bool is_new = false;
                const long long t_e = AZG_CLK();
#ifndef AZG_SEARCH_OWN_RNG
#define AZG_SEARCH_OWN_RNG 0
#endif
                if constexpr (G::STOCHASTIC || !AZG_SEARCH_OWN_RNG)
                    child = uni_u32(resolve_edge<G, SelState, ASYNC>(F, t, H, sm, rh.node_id, a, seed, leaf_states, leaf_valid, &is_new, &leaf_terminal, es,
                                                              srng, spec_now, ps0, ps1, ps2));
                else {
                    // (AZG_SEARCH_OWN_RNG, measured and off: a search's env step never draws from the tree's stream unless the game says so --
                    // random_seed is a magic seed or -1, MCTS.py:63, never the 0 that means "true random" -- so a stream of its own here would
                    // keep the tree's 64-bit counter out of the descent's registers; but at the 128-register cap the allocation of
                    // k_async_select<SplendorDev<2>> came out WORSE with it -- a register that carries spilled scalars was itself spilled:
                    // scratch instructions 17 -> 93, a descent 20.3 -> 21.6 us)
                    Rng none{0ull, 0ull, 0ull};
                    child = uni_u32(resolve_edge<G, SelState, ASYNC>(F, t, H, sm, rh.node_id, a, seed, leaf_states, leaf_valid, &is_new, &leaf_terminal, es,
                                                              none, spec_now, ps0, ps1, ps2));
                }
                cyc_edge += AZG_CLK() - t_e;
                if (child == AZG_NONE) { H.sim_idx = H.n_sims; break; }
                // memoise: this universe's slot -- or every slot when the env step of `a` cannot depend on the seed (the
                // child of (state, a, seed) is then the same node for all universes; 30 % of the simulations used to
                // re-resolve such an edge once per universe only to find the node through the hash table)
                // STOCHASTIC games (true randomness inside make_move, MCTS.py:238 re-rolls it at every traversal): the child of
                // (state, action) is a random variable, so the slot stays empty and every visit replays the step
                if (G::STOCHASTIC) {
                } else if (G::move_uses_seed(a)) {
                    if (l == 0) *(uint32_t*)((uint8_t*)rp + child_off((uint32_t)j, (uint32_t)uidx)) = child;
                } else if (l < F.U) *(uint32_t*)((uint8_t*)rp + child_off((uint32_t)j, (uint32_t)l)) = child;
                have_leaf = is_new;
            }
            const int np = (int)(child >> AZG_CHILD_NP_SHIFT);
            if (l == 0) {
                PathEnt e; e.rec = rec; e.j = (uint16_t)j; e.np = (uint8_t)np; e.pre = (uint8_t)pre;
                sm.path[depth] = e;
            }
            wave_sync();
            pre = (pre + np) % G::P;
            depth++;
            rec = child & AZG_CHILD_IDX_MASK;
        }
        if (paused) break;
        if (!have_leaf) continue;
        if (leaf_terminal) {                                                                      // MCTS.py:132-135
            c_term++;
            FR::backup(F, t, sm.path, depth, es);
            H.sim_idx++;
            continue;
        }
        PathEnt* gp = F.path + (size_t)t * AZG_MAXD;
        for (int d = l; d < depth; d += 64) gp[d] = sm.path[d];
        H.pending_leaf = rec;
        H.path_len = (uint32_t)depth;
        H.status = ST_WAIT_NN;
        need_nn = true;
        break;
    }
    if (l == 0) {
        Hp->n_nodes = H.n_nodes; Hp->heap_top = H.heap_top; Hp->root = H.root; Hp->root_rec = H.root_rec;
        Hp->id_top = H.id_top; Hp->n_free_ids = H.n_free_ids; Hp->free_units = H.free_units;
        Hp->sim_idx = H.sim_idx; Hp->err = H.err; Hp->leaf_is_root = H.leaf_is_root; Hp->mid_sim = H.mid_sim;
        Hp->cur_rec = H.cur_rec; Hp->cur_depth = H.cur_depth; Hp->cur_pre = H.cur_pre; Hp->status = H.status;
        Hp->pending_leaf = H.pending_leaf; Hp->path_len = H.path_len; Hp->pending_nv = H.leaf_nv; Hp->pending_node = H.leaf_node;
        if (G::STOCHASTIC) Hp->rng_counter = srng.counter;
        // statistics: no-return atomics, so the wave does not wait for a read-modify-write round trip before it retires
        atomicMax(&Hp->max_nodes_seen, H.n_nodes);
        stat_add(&Hp->c_sims, c_sims); stat_add(&Hp->c_levels, c_levels); stat_add(&Hp->c_sumvalid, c_sumvalid);
        stat_add(&Hp->c_term, c_term);
#ifdef AZG_CYC_COUNTERS
#ifndef AZG_WALL_CAL
        Hp->pad0_ += edges_this_launch;      // debug: frontier-edge resolutions of this tree (tools/archive/dbg_tail.py)
#endif
#ifdef AZG_WALL_CAL
        Hp->pad0_ = w_first; Hp->pad1_ = (uint32_t)wall_clock64(); (void)w_start;   // absolute 100 MHz stamps: wave start / end
#endif
        stat_add(&Hp->cyc_select, (uint64_t)(AZG_CLK() - t_start)); stat_add(&Hp->cyc_levels, (uint64_t)cyc_levels);
        stat_add(&Hp->cyc_edge, (uint64_t)cyc_edge); stat_add(&Hp->cyc_leaf, (uint64_t)H.cyc_leaf);
        H.cyc_seg[0] += (uint32_t)(t_start - t_first);          // prologue: header round trip + fused expansion + backup
        if (azg_stamp[5] && (t & 63) == 0) for (int k = 1; k < 7; k++) atomicAdd(&g_prolog[k], (unsigned long long)(azg_stamp[k] - azg_stamp[k - 1]));
        if (azg_stamp[5] && (t & 63) == 0) atomicAdd(&g_prolog[7], (unsigned long long)(azg_stamp[7] - azg_stamp[5]));   // end of expansion -> status checks done
        if (azg_stamp[5] && (t & 63) == 0) atomicAdd(&g_prolog[0], 1ull);     // (a sample of the trees: 4096 waves on one word serialise)
        for (int k = 0; k < 4; k++) stat_add(&Hp->cyc_seg[k], (uint64_t)H.cyc_seg[k]);
#else
        (void)t_start; (void)t_first; (void)cyc_levels; (void)cyc_edge; (void)azg_stamp;
#endif
        needs_eval[t] = need_nn ? 1 : 0;
    }
    return need_nn ? 1 : (uni_u32(H.mid_sim) ? 2 : 0);
}

template <class G>
__global__ __launch_bounds__(64, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_select(ForestDev F, int8_t* leaf_states, uint8_t* leaf_valid,
                                                  uint8_t* needs_eval, int wait_noise, const float* pi, const float* vin,
                                                  int noise_enabled) {
    __shared__ typename Forest<G>::Smem sm;
    __shared__ __attribute__((aligned(16))) float dense[G::A];          // fused expansion only
    static_assert(sizeof(ForestDev) + 52 <= 448 && sizeof(ForestDev) + 52 > 0x180 + 4,
                  "warm_kernarg_448 touches exactly the seven 64-byte lines of this kernel's explicit arguments");
    warm_kernarg_448();
    select_tree<G>(F, (int)blockIdx.x, sm, dense, leaf_states, leaf_valid, needs_eval, wait_noise, pi, vin, noise_enabled);
}

// MCTS.getActionProb epilogue (MCTS.py:67-103) into dense LDS buffers; returns false if the root has no policy.
// counts (after policy-target pruning) land in `cnt` (int), all lanes must call.
template <class G>
__device__ bool root_counts(const ForestDev& F, int t, const TreeHdr& H, int* cnt /*LDS [A]*/, float* q /*[P] regs*/) {
    using FR = Forest<G>;
    const int l = lane_id();
    for (int i = l; i < G::A; i += 64) cnt[i] = 0;
    wave_sync();
    if (H.root_rec == AZG_NONE) return false;
    const uint8_t* rec = FR::rec_ptr(F, t, H.root_rec);
    const RecHdr rh = load_uniform((const RecHdr*)rec);
    const float q0 = rh.Qs;                                                                      // :71-72
#pragma unroll
    for (int p = 0; p < G::P; p++) q[p] = p == 0 ? q0 : -q0 / (float)(G::P - 1);
    if (!(rh.flags & NF_EXPANDED)) return false;
    const RecGeom RG = FR::geom(F);
    const RecIds ids(rec, RG);
    int best = 0;
    for (int j = l; j < rh.nv; j += 64) {
        int n = (int)*(const uint32_t*)(rec + RG.hot((uint32_t)j) + AZG_H_N);
        best = n > best ? n : best;
    }
    best = wave_max_i32(best);
    for (int j = l; j < rh.nv; j += 64) {
        const uint8_t* ent = rec + RG.hot((uint32_t)j);
        int c = (int)*(const uint32_t*)(ent + AZG_H_N);
        if (H.forced) {                                                                          // :75-80
            if (c != best) {
                float tq = (0.5f * *(const float*)(ent + AZG_H_P)) * (float)H.n_sims;   // plain-Python NumPy scalar typing
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
    const TreeHdr H = load_uniform(&F.hdr[t]);
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
        // one-hot on a maximum of the (pruned) counts; among several maxima the reference draws np.random.choice(bestAs):
        // here the k-th of nb tied actions, k = floor(u * nb), with u the next uniform of this tree's counter RNG stream
        // (include/azg.h "RNG contract"); a unique maximum consumes no uniform
        int best = -1, nb = 0;
        for (int a = 0; a < G::A; a++) if (cnt[a] > best) best = cnt[a];
        for (int a = 0; a < G::A; a++) nb += cnt[a] == best;
        int k = 0;
        if (nb > 1) {
            Rng rng{forest_seed(F), F.stream0 + (uint64_t)t, H.rng_counter};
            k = (int)(rng.u01() * (double)nb);
            k = k >= nb ? nb - 1 : k;
            if (l == 0) F.hdr[t].rng_counter = rng.counter;
        }
        int ba = 0;
        for (int a = 0; a < G::A; a++) if (cnt[a] == best) { if (k-- == 0) { ba = a; break; } }
        for (int a = l; a < G::A; a += 64) pr[a] = a == ba ? 1.0 : 0.0;
        return;
    }
    double s = 0.0;                                                                              // :100-103 sequential sum
    const double e = 1.0 / temp;
    // (in index order over the actions with a count: a zero count contributes 0.0 -- or 0.0 ** e = 0.0 -- and leaves the sum as it is;
    // every lane steps through the same actions, one ballot per 64 of them, instead of evaluating all A powers itself)
    for (int base = 0; base < G::A; base += 64) {
        const int a = base + l;
        uint64_t m = __ballot(a < G::A && cnt[a < G::A ? a : 0] != 0);
        while (m) {
            const int b = base + __builtin_ctzll(m);
            m &= m - 1;
            s += (temp == 1.0) ? (double)cnt[b] : pow((double)cnt[b], e);
        }
    }
    for (int a = l; a < G::A; a += 64) pr[a] = ((temp == 1.0) ? (double)cnt[a] : pow((double)cnt[a], e)) / s;
}

template <class G>
__global__ __launch_bounds__(64) void k_root_stats(ForestDev F, int32_t* Ns, float* Qs, int32_t* Nsa, double* Qsa,
                                                   float* Ps, int32_t* n_nodes) {
    using FR = Forest<G>;
    const int t = blockIdx.x;
    const int l = lane_id();
    const TreeHdr H = load_uniform(&F.hdr[t]);
    if (n_nodes && l == 0) n_nodes[t] = (int32_t)H.n_nodes;
    for (int a = l; a < G::A; a += 64) {
        if (Nsa) Nsa[(size_t)t * G::A + a] = 0;
        if (Qsa) Qsa[(size_t)t * G::A + a] = AZG_NANQ;
        if (Ps) Ps[(size_t)t * G::A + a] = 0.f;
    }
    if (H.root_rec == AZG_NONE) { if (l == 0) { if (Ns) Ns[t] = -1; if (Qs) Qs[t] = 0.f; } return; }
    const uint8_t* rec = FR::rec_ptr(F, t, H.root_rec);
    const RecHdr rh = load_uniform((const RecHdr*)rec);
    if (l == 0) { if (Ns) Ns[t] = (int32_t)rh.Ns; if (Qs) Qs[t] = rh.Qs; }
    if (!(rh.flags & NF_EXPANDED)) return;
    __syncthreads();
    const RecGeom RG = FR::geom(F);
    const RecIds ids(rec, RG);
    for (int j = l; j < rh.nv; j += 64) {
        const uint8_t* ent = rec + RG.hot((uint32_t)j);
        const int a = ids[j];
        if (Nsa) Nsa[(size_t)t * G::A + a] = (int32_t)*(const uint32_t*)(ent + AZG_H_N);
        if (Qsa) Qsa[(size_t)t * G::A + a] = *(const double*)(ent + AZG_H_Q);
        if (Ps) Ps[(size_t)t * G::A + a] = *(const float*)(ent + AZG_H_P);
    }
}

}  // namespace azg
