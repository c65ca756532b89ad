// selfplay.hip.h -- Coach.executeEpisode (Coach.py:37-84) as a per-tree device state machine.
// Each tree plays its own game: when its search finishes it samples the move, records the example, plays the move
// (true-random chance via the RNG contract), detects the end of the game, emits the finished game's examples to the
// on-device ring, restarts, canonicalises the new root and begins the next search -- no host involvement, so trees
// run out of phase and the leaf batch for the net stays full.
#pragma once
#include "kernels.hip.h"

namespace azg {

// temp_for_selfplay (Coach.py:266-271)
__device__ __forceinline__ double temp_for_selfplay(const ForestDev& F, int n) {
    const double tb = F.temp_begin, te = F.temp_end, hl = F.tempThreshold;
    if (hl < 0) return (n > -hl) ? te : tb;
    return te + (tb - te) * pow(0.5, (double)n / hl);
}

// Drop every node that can no longer be reached: round < root_round (the move counter is part of the state, so such
// states cannot recur).  Equivalent to -- and stricter in memory than -- the reference's lazy clean-up MCTS.py:86-91,
// which removes nodes with round < r-5 every >20 rounds.
// Records and node ids NEVER MOVE, so no child slot has to be rewritten: a live node (round >= r) can only point to nodes
// with a round >= its own, i.e. to live ones.  One lane-parallel scan over the node headers
//   * links every dead node's record into the free list of its size (n_valid) -- list heads are staged in LDS, so the 64
//     lanes push concurrently with LDS atomics and write one 4-byte link per record --,
//   * pushes its id on the free-id stack (ballot + rank), marks the header NF_FREE,
//   * re-inserts every live node into the freshly cleared hash table (atomicCAS, linear probing).
// ~n/64 pipelined header loads instead of the node-by-node compaction this replaces (which cost 5-20 ms per tree and
// throttled sustained self-play to a tenth of its fresh-start rate).
// `n_waves` wavefronts of one workgroup share the scan (wave w takes the 64-id groups w, w + n_waves, ...); the list heads
// and the three running counters live in LDS.  With n_waves == 1 the barriers compile away.
// WAVE_ONLY: the caller is ONE wave of a workgroup whose other waves do something else (the asynchronous pipeline, azg_async.hip.h): the
// hand-overs between the lanes are wavefront fences, not workgroup barriers; the caller drops its CU's L1 afterwards (the table is
// re-filled with L2 atomics, which the L1 does not see).
template <class G, bool WAVE_ONLY = false>
__device__ __forceinline__ void gc_scan(const ForestDev& F, int t, TreeHdr& H, int min_round, uint32_t* lds_head /*[A + 1]*/,
                                        uint32_t* lds_ctr /*[4]*/, int wave, int n_waves) {
#define AZG_GC_SYNC() do { if (WAVE_ONLY) wave_sync(); else __syncthreads(); } while (0)
#define AZG_GC_FENCE() do { if (!WAVE_ONLY) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
    using FR = Forest<G>;
    const int l = lane_id(), tid = wave * 64 + l, nthr = n_waves * 64;
    uint8_t* hp = FR::heap(F, t);
    uint32_t* tab = FR::htab(F, t);
    uint32_t* gfree = FR::rec_free(F, t);
    uint32_t* ids = F.free_ids + (size_t)t * F.s_free;
    const uint32_t n = H.id_top;
    const uint32_t maskHT = (uint32_t)F.HT - 1u;
    const int nc = FR::n_classes(F);
    for (int i = tid; i < nc; i += nthr) lds_head[i] = gfree[i];
    for (int i = tid; i < F.HT; i += nthr) tab[i] = AZG_NONE;
    if (tid == 0) { lds_ctr[0] = H.n_free_ids; lds_ctr[1] = 0u; lds_ctr[2] = 0u; }
    AZG_GC_FENCE();
    AZG_GC_SYNC();
    for (uint32_t base = (uint32_t)wave * 64u; base < n; base += (uint32_t)nthr) {
        const uint32_t i = base + (uint32_t)l;
        NodeHdr nh;
        nh.hash = 0; nh.rec_off = 0; nh.nv = 0; nh.round = 0; nh.flags = NF_FREE;
        if (i < n) nh = *FR::nhdr(F, t, i);
        const bool in_use = !(nh.flags & NF_FREE);
        const bool dead = in_use && (int)nh.round < min_round && i != H.root;
        const bool alive = in_use && !dead;
        if (dead) {
            if (!FR::ONE_CLASS) {                                                       // (one-class forests: slot == node id)
                const uint32_t old = atomicExch(&lds_head[nh.nv], nh.rec_off);           // LDS: lanes of one size chain up
                *(uint32_t*)(hp + (size_t)nh.rec_off * 16u) = old;
            }
            FR::nhdr(F, t, i)->flags = NF_FREE;
        }
        const uint64_t bd = __ballot(dead), ba = __ballot(alive);
        const int units = !FR::ONE_CLASS ? wave_sum_i32(dead ? (int)FR::cls_units(F, nh.nv) : 0) : 0;
        uint32_t pos = 0;
        if (l == 0) {
            if (bd) { pos = atomicAdd(&lds_ctr[0], (uint32_t)__popcll(bd)); atomicAdd(&lds_ctr[2], (uint32_t)units); }
            if (ba) atomicAdd(&lds_ctr[1], (uint32_t)__popcll(ba));
        }
        pos = uni_u32(pos);
        if (dead) ids[pos + (uint32_t)__popcll(bd & ((1ull << l) - 1ull))] = i;
        if (alive) {
            const uint32_t entry = (FR::tag_of(nh.hash) << AZG_IDX_BITS) | i;
            uint32_t s = (uint32_t)nh.hash & maskHT;
            while (atomicCAS(&tab[s], AZG_NONE, entry) != AZG_NONE) s = (s + 1u) & maskHT;
        }
    }
    AZG_GC_SYNC();
    for (int i = tid; i < nc; i += nthr) gfree[i] = lds_head[i];
    AZG_GC_FENCE();
    H.n_free_ids = uni_u32(lds_ctr[0]);
    H.n_nodes = uni_u32(lds_ctr[1]);
    if (H.n_nodes > H.max_live) H.max_live = H.n_nodes;
    H.free_units += uni_u32(lds_ctr[2]);
    H.gc_runs++;
    AZG_GC_SYNC();
#undef AZG_GC_SYNC
#undef AZG_GC_FENCE
}

// single-wave form (host-driven searches)
template <class G>
__device__ __noinline__ void gc_tree(const ForestDev& F, int t, TreeHdr& H, int min_round, uint32_t* lds_head /*[A + 5]*/) {
    gc_scan<G>(F, t, H, min_round, lds_head, lds_head + G::A + 1, 0, 1);
}

template <class G>
__device__ void reset_tree(const ForestDev& F, int t, TreeHdr& H) {
    uint32_t* tab = Forest<G>::htab(F, t);
    for (int i = lane_id(); i < F.HT; i += 64) tab[i] = AZG_NONE;
    for (int i = lane_id(); i < Forest<G>::n_classes(F); i += 64) Forest<G>::rec_free(F, t)[i] = AZG_NONE;
    H.n_nodes = 0; H.heap_top = 0; H.root = AZG_NONE; H.root_rec = AZG_NONE;
    H.id_top = 0; H.n_free_ids = 0; H.free_units = 0;
    wave_sync();
}

// new game on this tree: Board.init_game (or a supplied board), player 0 to move
template <class G>
__device__ void start_game(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm, Rng& rng,
                           const int8_t* init_board) {
    using FR = Forest<G>;
    reset_tree<G>(F, t, H);
    if (init_board) FR::load_state_unpadded(sm.st, init_board);
    else {
        for (int i = lane_id(); i < G::SP; i += 64) sm.st[i] = 0;
        wave_sync();
        if (lane_id() == 0) G::init_board(sm.st, rng);
        rng.counter = bcast_u64(rng.counter, 0);
        wave_sync();
    }
    FR::store_state(F.board + (size_t)t * G::SP, sm.st);
    H.cur_player = 0; H.ply = 0; H.step = 0; H.n_rec = 0;
}

// Episode quota (Coach.executeEpisodes plays exactly numEps episodes, every one to its end, Coach.py:86-148): with
// ForestDev.episode_quota = numEps != 0, tree t plays numEps / T (+1 for t < numEps % T) games and then goes idle instead of
// restarting, so every started game finishes and is kept -- no bias towards short games.  0 = restart forever.
__device__ __forceinline__ uint32_t tree_quota(const ForestDev& F, int t) {
    const uint32_t q = F.episode_quota;
    if (q == 0xFFFFFFFFu) return 0u;                  // (azg_selfplay_start_ex with episode_quota -1: no game on this forest)
    return q / (uint32_t)F.T + ((uint32_t)t < q % (uint32_t)F.T ? 1u : 0u);
}

template <class G>
__global__ __launch_bounds__(64) void k_selfplay_start(ForestDev F, const int8_t* init_boards) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    const int t = blockIdx.x;
    TreeHdr H = load_uniform(&F.hdr[t]);
    Rng rng{forest_seed(F), F.stream0 + (uint64_t)t, 0ull};
    H.err = 0; H.games_done = 0; H.gc_runs = 0; H.max_nodes_seen = 0; H.max_live = 0;
    H.c_sims = H.c_levels = H.c_exp = H.c_sumvalid = H.c_term = H.c_depth = H.c_plies = H.c_examples = 0;
    if (F.episode_quota != 0u && tree_quota(F, t) == 0u) {       // fewer episodes than trees: this tree plays none
        reset_tree<G>(F, t, H);
        H.status = ST_IDLE; H.noise_pending = 0; H.pending_leaf = AZG_NONE; H.mid_sim = 0; H.n_rec = 0; H.rng_counter = 0;
        if (lane_id() == 0) F.hdr[t] = H;
        return;
    }
    start_game<G>(F, t, H, sm, rng, init_boards ? init_boards + (size_t)t * G::S : nullptr);
    // board is canonical for player 0 (Coach.py:61 with curPlayer == 0)
    const double u_full = rng.u01();
    begin_search_from_lds<G>(F, t, H, sm, u_full < F.prob_fullMCTS);
    H.rng_counter = rng.counter;
    if (lane_id() == 0) F.hdr[t] = H;
}

// Memory reclamation before a search from the canonical state in sm.st: when the arena or the record heap could not take
// another numMCTSSims nodes, drop everything older than the new root's round (the reference's clean-up deletes nodes with
// round < r-5 every >20 rounds, MCTS.py:86-91; unreachable nodes never influence a search, so dropping them earlier does
// not change any result).
template <class G>
__device__ __forceinline__ bool arena_is_short(const ForestDev& F, const TreeHdr& H) {
    using FR = Forest<G>;
    const uint32_t need_nodes = (uint32_t)F.numMCTSSims + 8u;
    const uint32_t need_units = need_nodes * FR::cls_units(F, FR::cls_of(F, G::A < 96 ? G::A : 96)) + 256u;
    // ids in use (dead ones included until the next clean-up) / record space left on the bump pointer and the free lists
    if ((H.id_top - H.n_free_ids) + need_nodes > (uint32_t)F.cap) return true;
    // high-water mark (cfg.gc_high_water_pct): clean up early enough that the arena never fills beyond it + one search
    if (F.gc_high_water && (H.id_top - H.n_free_ids) > F.gc_high_water) return true;
    return !FR::ONE_CLASS && (F.heap_units - H.heap_top) + H.free_units < need_units;   // one-class forests: slot == node id
}

// locate the new root (canonical state in sm.st) so that the clean-up keeps it
template <class G>
__device__ __forceinline__ void locate_root(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm) {
    using FR = Forest<G>;
    const uint64_t h = FR::hash_state(sm.st);
    uint32_t free_slot;
    uint32_t found_rec = AZG_NONE;
    H.root = FR::probe(F, t, sm.st, h, &free_slot, &found_rec);
}

template <class G>
__device__ __forceinline__ void reclaim_if_short(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm,
                                                 uint32_t* lds_head /*[A + 5]*/) {
    if (arena_is_short<G>(F, H)) {
        locate_root<G>(F, t, H, sm);
        gc_tree<G>(F, t, H, G::gc_age(sm.st), lds_head);
    }
}

// the tail of a self-play ply: playout-cap draw, search set-up, root noise (MCTS.py:58-64) -- canonical root in sm.st
template <class G>
__device__ __forceinline__ void begin_next_search(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm, Rng& rng,
                                                  float* dense /*LDS [A]*/) {
    const double u_full = rng.u01();                                                           // MCTS.py:58
    begin_search_from_lds<G>(F, t, H, sm, u_full < F.prob_fullMCTS);
    H.rng_counter = rng.counter;
    if (H.n_nodes > H.max_nodes_seen) H.max_nodes_seen = H.n_nodes;
    // an EXISTING root gets its noise before simulation 0 (MCTS.py:64,156-160): begin_search_from_lds left noise_pending set, the
    // k_root_noise launch that follows the advance kernels (azg_selfplay_advance) applies it
    (void)dense;
}

// Self-play clean-up, one workgroup of 16 wavefronts per tree that asked for it (status ST_GC): the header scan is a chain
// of dependent memory round trips per 64 ids, so 16 waves bring a 13 k-node tree from ~0.5 ms to ~40 us.
template <class G>
__global__ __launch_bounds__(1024) void k_gc(ForestDev F) {
    __shared__ uint32_t head[G::A + 1];
    __shared__ uint32_t ctr[4];
    const int t = blockIdx.x;
    if (ld_agent_u32(&F.hdr[t].status) != ST_GC) return;
    TreeHdr H = load_uniform(&F.hdr[t]);
    const int wave = (int)(threadIdx.x >> 6);
    gc_scan<G>(F, t, H, (int)H.cur_pre, head, ctr, wave, 16);
    if (threadIdx.x == 0) {
        TreeHdr* Hp = &F.hdr[t];
        Hp->n_free_ids = H.n_free_ids; Hp->n_nodes = H.n_nodes; Hp->max_live = H.max_live; Hp->free_units = H.free_units;
        Hp->gc_runs = H.gc_runs; Hp->status = ST_GC_DONE;
    }
}

// the search of a cleaned-up tree (status ST_GC_DONE) begins: one wave, the caller supplies the LDS blocks
template <class G>
__device__ __forceinline__ void after_gc_tree(const ForestDev& F, int t, typename Forest<G>::Smem& sm, float* dense /*LDS [A]*/) {
    using FR = Forest<G>;
    TreeHdr H = load_uniform(&F.hdr[t]);
    Rng rng{forest_seed(F), F.stream0 + (uint64_t)t, H.rng_counter};
    FR::load_state(sm.st, F.root_state + (size_t)t * G::SP);
    begin_next_search<G>(F, t, H, sm, rng, dense);
    if (lane_id() == 0) F.hdr[t] = H;
}

template <class G>
__global__ __launch_bounds__(64) void k_after_gc(ForestDev F) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    const int t = blockIdx.x;
    if (ld_agent_u32(&F.hdr[t].status) != ST_GC_DONE) return;
    after_gc_tree<G>(F, t, sm, dense);
}

// MCTS.getActionProb prologue for host-driven searches (azg_forest_begin_search)
template <class G>
__global__ __launch_bounds__(64) void k_begin_search(ForestDev F, const int8_t* roots, const uint8_t* full) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    __shared__ uint32_t gc_head[G::A + 5];
    const int t = blockIdx.x;
    if (full && full[t] == 2) {                    // tree sits this search out (e.g. the other player's turn): keep its contents
        if (lane_id() == 0) F.hdr[t].status = ST_IDLE;
        return;
    }
    TreeHdr H = load_uniform(&F.hdr[t]);
    FR::load_state_unpadded(sm.st, roots + (size_t)t * G::S);
    if (H.n_nodes) reclaim_if_short<G>(F, t, H, sm, gc_head);
    begin_search_from_lds<G>(F, t, H, sm, full ? full[t] != 0 : true);
    if (H.n_nodes > H.max_nodes_seen) H.max_nodes_seen = H.n_nodes;
    if (lane_id() == 0) F.hdr[t] = H;
}

// The per-ply work of Coach.executeEpisode for ONE tree whose search is finished (status ST_DONE), run by one wave; the caller supplies the
// LDS blocks (k_selfplay_advance: one single-wave workgroup per tree; the asynchronous pipeline: the descent wave that found the search
// finished, azg_async.hip.h).  Leaves the tree searching again, waiting for the clean-up (ST_GC), idle (episode quota) or parked with an
// error flag.
template <class G>
__device__ __forceinline__ void advance_tree(const ForestDev& F, const int t, typename Forest<G>::Smem& sm, int* cnt /*LDS [A]*/,
                                             double* w /*LDS [A]*/, float* dense /*LDS [A]*/) {
    using FR = Forest<G>;
    const int l = lane_id();
    TreeHdr H = load_uniform(&F.hdr[t]);
    if (H.err) return;                         // tree is parked; the host reads the error flag
    Rng rng{forest_seed(F), F.stream0 + (uint64_t)t, H.rng_counter};
    float q[G::P];
#pragma unroll
    for (int p = 0; p < G::P; p++) q[p] = 0.f;
    root_counts<G>(F, t, H, cnt, q);
    // ---- pi with temp = 1 (MCTS.py:100-103), then random_pick with the self-play temperature (Coach.py:62-63) ----
    long long tot = 0;                          // (integer: any order; every lane ends with the total)
    for (int a = l; a < G::A; a += 64) tot += cnt[a];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) tot += (long long)shfl_xor_u64((uint64_t)tot, m);
    if (tot == 0) {
        // policy-target pruning left no count above 1 (too few simulations for the number of valid actions): the reference computes 0 / 0 and
        // raises at MCTS.py:100-102.  Park the tree with an error flag instead of playing an arbitrary move.
        if (l == 0) atomicOr(&F.hdr[t].err, (uint32_t)ERR_EMPTY_POLICY);
        return;
    }
    const double T = temp_for_selfplay(F, (int)H.step + 1);
    const double u_pick = rng.u01();
    // temperature == 0 (Coach.py:278-292): np.random.choice(bests) takes u_pick (the k-th of the nb maxima, k = floor(u * nb)), and
    // random_pick then still calls np.random.choice(len(p), p = the one-hot result) -- a second uniform is consumed, whatever its value
    if (T == 0.0) (void)rng.u01();
    for (int a = l; a < G::A; a += 64) {
        double p = tot > 0 ? (double)cnt[a] / (double)tot : 0.0;
        w[a] = (T == 0.0) ? p : pow(p, 1.0 / T);
    }
    wave_sync();
    // The reference's sums run over all A entries in index order; an entry without visits is 0.0 and leaves every running f64 sum as it
    // is, so only the actions with visits take part: per 64 actions one ballot says which have any, and the wave -- every lane the same
    // arithmetic on broadcast reads -- steps through those in ascending order: a few dozen steps instead of three passes of one lane over
    // 1782 / 3402 / 4056 entries with two f64 divisions each.
    auto visited = [&](auto&& fn) {            // fn(a) for every a with w[a] != 0, ascending, until it returns false
        for (int base = 0; base < G::A; base += 64) {
            const int a = base + l;
            uint64_t m = __ballot(a < G::A && w[a < G::A ? a : 0] != 0.0);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                if (!fn(base + b)) return;
            }
        }
    };
    int action = 0;
    if (T == 0.0) {                                                        // uniform among the maxima (Coach.py:279-282)
        double mx = -1.0; int nb = 0;
        visited([&](int a) { mx = w[a] > mx ? w[a] : mx; return true; });
        visited([&](int a) { nb += w[a] == mx; return true; });
        int k = (int)(u_pick * nb); k = k >= nb ? nb - 1 : k;
        visited([&](int a) { if (w[a] == mx) { if (k-- == 0) { action = a; return false; } } return true; });
    } else {
        double s = 0.0;
        visited([&](int a) { s += w[a]; return true; });
        double tot2 = 0.0;
        visited([&](int a) { tot2 += w[a] / s; return true; });
        double cdf = 0.0; int pick = -1, last = 0;
        visited([&](int a) {
            const double pa = w[a] / s;
            cdf += pa;
            if (pa > 0) last = a;
            if (cdf / tot2 > u_pick) { pick = a; return false; }
            return true;
        });
        action = pick < 0 ? last : pick;
    }
    action = __shfl(action, 0, 64);
    // ---- record the example on full searches (Coach.py:65-69; symmetries are applied by the consumer) ----
    if (H.is_full) {
        if (H.n_rec < (uint32_t)F.max_rec) {
            const size_t r = (size_t)t * F.max_rec + H.n_rec;
            const int8_t* rs = F.root_state + (size_t)t * G::SP;
            for (int i = l; i < G::S; i += 64) F.rec_board[r * G::S + i] = rs[i];
            for (int a = l; a < G::A; a += 64) {
                F.rec_pi[r * G::A + a] = (float)(tot > 0 ? (double)cnt[a] / (double)tot : 0.0);
                F.rec_valid[r * G::A + a] = 0;
            }
            wave_sync();
            if (H.root_rec != AZG_NONE) {
                const uint8_t* rrec = FR::rec_ptr(F, t, H.root_rec);
                const RecHdr rh = load_uniform((const RecHdr*)rrec);
                if (rh.flags & NF_EXPANDED) {
                    const RecIds ids(rrec, FR::geom(F));
                    for (int j = l; j < rh.nv; j += 64) F.rec_valid[r * G::A + ids[j]] = 1;
                }
            }
            if (l == 0) {
                for (int p = 0; p < G::P; p++) F.rec_q[r * G::P + p] = q[p];
                F.rec_player[r] = (uint8_t)H.cur_player;
                F.rec_ply[r] = (uint16_t)H.ply;
            }
            H.n_rec++;
        } else H.err |= ERR_REC_OVERFLOW;
    }
    // ---- play the move for real: random_seed = 0 (Coach.py:71) ----
    FR::load_state(sm.st, F.board + (size_t)t * G::SP);
    const int np = G::wave_make_move(sm.st, action, (int)H.cur_player, 0ll, rng);
    H.c_plies++;
    float es[G::P];
    const bool ended = G::game_ended(sm.st, np, es, sm.mask);                                  // Coach.py:73
    if (ended) {
        // z = np.roll(r, -player) (Coach.py:76-82): z[i] = r[(i + player) mod P]
        // reserve the ring slots of the whole game or of none of it (a finished game is never truncated); a game that
        // does not fit is dropped and counted in ex_count[1] (azg_selfplay_stats.examples_dropped, error bit 16)
        unsigned long long base = 0;
        if (l == 0) {
            const unsigned long long n = (unsigned long long)H.n_rec, cap = (unsigned long long)F.max_examples;
            unsigned long long seen = *(volatile unsigned long long*)F.ex_count;
            while (true) {
                if (seen + n > cap) { base = ~0ull; atomicAdd(F.ex_count + 1, n); break; }
                const unsigned long long prev = atomicCAS(F.ex_count, seen, seen + n);
                if (prev == seen) { base = seen; break; }
                seen = prev;
            }
        }
        base = bcast_u64(base, 0);
        for (uint32_t k = 0; k < H.n_rec && base != ~0ull; k++) {
            const size_t r = (size_t)t * F.max_rec + k;
            const unsigned long long dst = base + k;
            for (int i = l; i < G::S; i += 64) F.ex_board[dst * G::S + i] = F.rec_board[r * G::S + i];
            for (int a = l; a < G::A; a += 64) {
                F.ex_pi[dst * G::A + a] = F.rec_pi[r * G::A + a];
                F.ex_valid[dst * G::A + a] = F.rec_valid[r * G::A + a];
            }
            if (l == 0) {
                const int pl = F.rec_player[r];
                for (int p = 0; p < G::P; p++) {
                    F.ex_z[dst * G::P + p] = es[(p + pl) % G::P];
                    F.ex_q[dst * G::P + p] = F.rec_q[r * G::P + p];
                }
                F.ex_meta[dst * 4 + 0] = (int32_t)(F.stream0 + (uint64_t)t);
                F.ex_meta[dst * 4 + 1] = (int32_t)H.games_done;
                F.ex_meta[dst * 4 + 2] = (int32_t)F.rec_ply[r];
                F.ex_meta[dst * 4 + 3] = pl;
            }
        }
        if (base != ~0ull) H.c_examples += H.n_rec;
        H.games_done++;
        if (F.episode_quota != 0u && H.games_done >= tree_quota(F, t)) {      // episode quota reached: no restart
            H.status = ST_IDLE; H.n_rec = 0; H.rng_counter = rng.counter;
            if (l == 0) F.hdr[t] = H;
            return;
        }
        start_game<G>(F, t, H, sm, rng, nullptr);
    } else {
        FR::store_state(F.board + (size_t)t * G::SP, sm.st);
        H.cur_player = (uint32_t)np;
        H.ply++;
        H.step++;
        if (np != 0) G::swap_players(sm.st, sm.tmp, np);                                      // Coach.py:61
    }
    // ---- memory reclamation, then the next search ----
    if (!ended && arena_is_short<G>(F, H)) {
        // hand the tree to the clean-up kernel (k_gc, 16 waves) and let k_after_gc begin the search
        locate_root<G>(F, t, H, sm);
        FR::store_state(F.root_state + (size_t)t * G::SP, sm.st);
        H.cur_pre = (uint32_t)G::gc_age(sm.st);
        H.rng_counter = rng.counter;
        H.status = ST_GC;
        if (l == 0) F.hdr[t] = H;
        return;
    }
    begin_next_search<G>(F, t, H, sm, rng, dense);
    if (l == 0) F.hdr[t] = H;
}

template <class G>
__global__ __launch_bounds__(64) void k_selfplay_advance(ForestDev F) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    __shared__ int cnt[G::A];
    __shared__ double w[G::A];
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    const int t = blockIdx.x;
    // nothing to advance unless the search is finished; a root expanded by simulation 0 may still be waiting for its Dirichlet noise
    // (MCTS.py:147-149): k_root_noise, launched right after, serves it
    if (ld_agent_u32(&F.hdr[t].status) != ST_DONE) return;
    advance_tree<G>(F, t, sm, cnt, w, dense);
}

}  // namespace azg
