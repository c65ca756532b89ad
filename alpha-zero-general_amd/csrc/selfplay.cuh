// selfplay.cuh -- Coach.executeEpisode (Coach.py:37-84) as a per-tree device state machine.
// Each tree plays its own game: when its search finishes it samples the move, records the example, plays the move
// (true-random chance via the RNG contract), detects the end of the game, emits the finished game's examples to the
// on-device ring, restarts, canonicalises the new root and begins the next search -- no host involvement, so trees
// run out of phase and the leaf batch for the net stays full.
#pragma once
#include "kernels.cuh"

namespace azg {

// temp_for_selfplay (Coach.py:266-271)
__device__ __forceinline__ double temp_for_selfplay(const ForestDev& F, int n) {
    const double tb = F.temp_begin, te = F.temp_end, hl = F.tempThreshold;
    if (hl < 0) return (n > -hl) ? te : tb;
    return te + (tb - te) * pow(0.5, (double)n / hl);
}

// Drop every node that can no longer be reached: round < root_round (the move counter is part of the state, so such
// states cannot recur).  Equivalent to -- and stricter in memory than -- the reference's lazy clean-up MCTS.py:86-91,
// which removes nodes with round < r-5 every >20 rounds.  In-place sliding compaction by the tree's own wave:
//   pass 1  per node: keep?, new node id, new record offset (wave prefix sums); both maps live in the hash-table memory
//   pass 2  rewrite the child slots of every kept record (old record offset -> child's node id -> new record offset)
//   pass 3  slide records / states / cold headers down (dst <= src, forward copies), fix RecHdr.node_id
//   pass 4  clear + re-insert the hash table
// Every control value is made explicitly wave-uniform (readfirstlane) so the barriers sit in uniform control flow.
template <class G>
__device__ __noinline__ void gc_tree(const ForestDev& F, int t, TreeHdr& H, int min_round) {
    using FR = Forest<G>;
    const int l = lane_id();
    uint32_t* map_id = FR::htab(F, t);               // [cap]   (HT >= 2*cap)
    uint32_t* map_rec = map_id + F.cap;              // [cap]
    uint8_t* hp = FR::heap(F, t);
    const uint32_t n = H.n_nodes;
    const uint32_t ES = entry_stride(F.U);
    // pass 1
    uint32_t kept = 0, top = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + l;
        bool keep = false;
        uint32_t units = 0;
        if (i < n) {
            const NodeHdr* nh = FR::nhdr(F, t, i);
            keep = (int)nh->round >= min_round || i == H.root;
            units = RecLayout(nh->nv, F.U).total / 16u;
        }
        const uint64_t b = __ballot(keep);
        const uint32_t rank = (uint32_t)__popcll(b & ((1ull << l) - 1ull));
        uint32_t incl = keep ? units : 0u;           // inclusive prefix sum of kept record sizes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (l >= d) incl += o;
        }
        if (i < n) {
            map_id[i] = keep ? kept + rank : AZG_NONE;
            map_rec[i] = keep ? top + incl - units : AZG_NONE;
        }
        kept += (uint32_t)__popcll(b);
        top += uni_u32(__shfl(incl, 63, 64));
    }
    wave_sync();
    // pass 2: child slots (records are still at their old places, so a child's node id can be read from its header)
    for (uint32_t i = 0; i < n; i++) {
        if (uni_u32(map_id[i]) == AZG_NONE) continue;
        const NodeHdr nh = *FR::nhdr(F, t, i);
        const uint32_t nv = uni_u32((uint32_t)nh.nv);
        if (!(uni_u32((uint32_t)nh.flags) & NF_EXPANDED)) continue;
        uint8_t* rec = hp + (size_t)uni_u32(nh.rec_off) * 16u;
        const uint32_t slots = nv * (uint32_t)F.U;
        for (uint32_t k = l; k < slots; k += 64) {
            uint32_t* sp = (uint32_t*)(rec + AZG_REC_HDR + (size_t)(k / (uint32_t)F.U) * ES + AZG_E_C + 4u * (k % (uint32_t)F.U));
            const uint32_t c = *sp;
            if (c != AZG_NONE) {
                const uint32_t cid = ((const RecHdr*)(hp + (size_t)(c & AZG_CHILD_IDX_MASK) * 16u))->node_id;
                const uint32_t nr = cid < n ? map_rec[cid] : AZG_NONE;
                *sp = nr == AZG_NONE ? AZG_NONE : ((c & ~AZG_CHILD_IDX_MASK) | nr);
            }
        }
    }
    wave_sync();
    // pass 3: slide (ascending node order == ascending record order)
    uint32_t new_root = AZG_NONE, new_root_rec = AZG_NONE;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t ni = uni_u32(map_id[i]);
        if (ni != AZG_NONE) {
            NodeHdr nh = *FR::nhdr(F, t, i);
            const uint32_t old_off = uni_u32(nh.rec_off);
            const uint32_t new_off = uni_u32(map_rec[i]);
            const uint32_t units = RecLayout((int)uni_u32((uint32_t)nh.nv), F.U).total / 16u;
            if (new_off != old_off) {
                const uint4* src = (const uint4*)(hp + (size_t)old_off * 16u);
                uint4* dst = (uint4*)(hp + (size_t)new_off * 16u);
                for (uint32_t base = 0; base < units; base += 64) {
                    const uint32_t k = base + (uint32_t)l;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (k < units) v = src[k];
                    wave_sync();                       // every lane has read its chunk before anyone overwrites it
                    if (k < units) dst[k] = v;
                    wave_sync();
                }
            }
            if (l == 0) ((RecHdr*)(hp + (size_t)new_off * 16u))->node_id = ni;
            if (ni != i) {
                const uint32_t* ssrc = (const uint32_t*)FR::nstate(F, t, i);
                uint32_t* sdst = (uint32_t*)FR::nstate(F, t, ni);
                for (int k = l; k < FR::SPW; k += 64) sdst[k] = ssrc[k];
            }
            nh.rec_off = new_off;
            if (l == 0) *FR::nhdr(F, t, ni) = nh;
            if (i == H.root) { new_root = ni; new_root_rec = new_off; }
        }
        wave_sync();
    }
    H.root = new_root;
    H.root_rec = new_root_rec;
    // pass 4
    uint32_t* tab = FR::htab(F, t);
    for (int i = l; i < F.HT; i += 64) tab[i] = AZG_NONE;
    wave_sync();
    const uint32_t maskHT = (uint32_t)F.HT - 1u;
    if (l == 0) {
        for (uint32_t i = 0; i < kept; i++) {
            const uint64_t h = FR::nhdr(F, t, i)->hash;
            uint32_t s = (uint32_t)h & maskHT;
            while (tab[s] != AZG_NONE) s = (s + 1u) & maskHT;
            tab[s] = (FR::tag_of(h) << AZG_IDX_BITS) | i;
        }
    }
    wave_sync();
    H.n_nodes = kept;
    H.heap_top = uni_u32(top);
    H.gc_runs++;
}

template <class G>
__device__ void reset_tree(const ForestDev& F, int t, TreeHdr& H) {
    uint32_t* tab = Forest<G>::htab(F, t);
    for (int i = lane_id(); i < F.HT; i += 64) tab[i] = AZG_NONE;
    H.n_nodes = 0; H.heap_top = 0; H.root = AZG_NONE; H.root_rec = AZG_NONE;
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

template <class G>
__global__ __launch_bounds__(64) void k_selfplay_start(ForestDev F, const int8_t* init_boards) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    const int t = blockIdx.x;
    TreeHdr H = load_uniform(&F.hdr[t]);
    Rng rng{F.rng_seed, F.stream0 + (uint64_t)t, 0ull};
    H.err = 0; H.games_done = 0; H.gc_runs = 0; H.max_nodes_seen = 0;
    H.c_sims = H.c_levels = H.c_exp = H.c_sumvalid = H.c_term = H.c_depth = H.c_plies = H.c_examples = 0;
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
__device__ __forceinline__ void reclaim_if_short(const ForestDev& F, int t, TreeHdr& H, typename Forest<G>::Smem& sm) {
    using FR = Forest<G>;
    if (H.n_nodes + (uint32_t)F.numMCTSSims + 8u > (uint32_t)F.cap ||
        H.heap_top + (uint32_t)(F.numMCTSSims + 8) * (RecLayout(G::A < 96 ? G::A : 96, F.U).total / 16u) + 256u > F.heap_units) {
        // locate the new root first so that GC can keep it
        const uint64_t h = wave_hash_state((const uint32_t*)sm.st, FR::SPW);
        uint32_t free_slot;
        uint32_t found_rec = AZG_NONE;
        H.root = FR::probe(F, t, sm.st, h, &free_slot, &found_rec);
        gc_tree<G>(F, t, H, G::get_round(sm.st));
    }
}

// MCTS.getActionProb prologue for host-driven searches (azg_forest_begin_search)
template <class G>
__global__ __launch_bounds__(64) void k_begin_search(ForestDev F, const int8_t* roots, const uint8_t* full) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    const int t = blockIdx.x;
    if (full && full[t] == 2) {                    // tree sits this search out (e.g. the other player's turn): keep its contents
        if (lane_id() == 0) F.hdr[t].status = ST_IDLE;
        return;
    }
    TreeHdr H = load_uniform(&F.hdr[t]);
    FR::load_state_unpadded(sm.st, roots + (size_t)t * G::S);
    if (H.n_nodes) reclaim_if_short<G>(F, t, H, sm);
    begin_search_from_lds<G>(F, t, H, sm, full ? full[t] != 0 : true);
    if (H.n_nodes > H.max_nodes_seen) H.max_nodes_seen = H.n_nodes;
    if (lane_id() == 0) F.hdr[t] = H;
}

template <class G>
__global__ __launch_bounds__(64, 4) void k_selfplay_advance(ForestDev F) {
    using FR = Forest<G>;
    __shared__ typename FR::Smem sm;
    __shared__ int cnt[G::A];
    __shared__ double w[G::A];
    __shared__ __attribute__((aligned(16))) float dense[G::A];
    const int t = blockIdx.x;
    const int l = lane_id();
    const uint32_t status0 = uni_u32(F.hdr[t].status);
    if (status0 != ST_DONE) {
        // nothing to advance; a root expanded by simulation 0 may still be waiting for its Dirichlet noise (MCTS.py:147-149)
        if (!uni_u32(F.hdr[t].noise_pending) || status0 != ST_SEARCHING) return;
        const uint32_t root_rec = uni_u32(F.hdr[t].root_rec);
        const uint64_t c_sims = F.hdr[t].c_sims;
        if (root_noise_tree<G>(F, t, root_rec, c_sims, nullptr, -1, dense, sm.mask) && l == 0) F.hdr[t].noise_pending = 0u;
        return;
    }
    TreeHdr H = load_uniform(&F.hdr[t]);
    if (H.err) return;                         // tree is parked; the host reads the error flag
    Rng rng{F.rng_seed, F.stream0 + (uint64_t)t, H.rng_counter};
    float q[G::P];
#pragma unroll
    for (int p = 0; p < G::P; p++) q[p] = 0.f;
    root_counts<G>(F, t, H, cnt, q);
    // ---- pi with temp = 1 (MCTS.py:100-103), then random_pick with the self-play temperature (Coach.py:62-63) ----
    long long tot = 0;
    for (int a = 0; a < G::A; a++) tot += cnt[a];
    const double T = temp_for_selfplay(F, (int)H.step + 1);
    const double u_pick = rng.u01();
    for (int a = l; a < G::A; a += 64) {
        double p = tot > 0 ? (double)cnt[a] / (double)tot : 0.0;
        w[a] = (T == 0.0) ? p : pow(p, 1.0 / T);
    }
    wave_sync();
    int action = 0;
    if (l == 0) {
        if (T == 0.0) {                                                    // uniform among the maxima (Coach.py:279-282)
            double mx = -1.0; int nb = 0;
            for (int a = 0; a < G::A; a++) mx = w[a] > mx ? w[a] : mx;
            for (int a = 0; a < G::A; a++) nb += w[a] == mx;
            int k = (int)(u_pick * nb); k = k >= nb ? nb - 1 : k;
            for (int a = 0; a < G::A; a++) if (w[a] == mx) { if (k-- == 0) { action = a; break; } }
        } else {
            double s = 0.0;
            for (int a = 0; a < G::A; a++) s += w[a];
            double tot2 = 0.0;
            for (int a = 0; a < G::A; a++) tot2 += w[a] / s;
            double cdf = 0.0; int pick = -1, last = 0;
            for (int a = 0; a < G::A; a++) {
                double pa = w[a] / s;
                cdf += pa;
                if (pa > 0) last = a;
                if (cdf / tot2 > u_pick) { pick = a; break; }
            }
            action = pick < 0 ? last : pick;
        }
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
                    const RecIds ids(rrec, F.U);
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
        unsigned long long base = 0;
        if (l == 0) base = atomicAdd(F.ex_count, (unsigned long long)H.n_rec);
        base = bcast_u64(base, 0);
        for (uint32_t k = 0; k < H.n_rec; k++) {
            const size_t r = (size_t)t * F.max_rec + k;
            const unsigned long long dst = base + k;
            if (dst >= (unsigned long long)F.max_examples) { if (l == 0) atomicAdd(F.ex_count + 1, 1ull); continue; }
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
        H.c_examples += H.n_rec;
        H.games_done++;
        start_game<G>(F, t, H, sm, rng, nullptr);
    } else {
        FR::store_state(F.board + (size_t)t * G::SP, sm.st);
        H.cur_player = (uint32_t)np;
        H.ply++;
        H.step++;
        if (np != 0) G::swap_players(sm.st, sm.tmp, np);                                      // Coach.py:61
    }
    // ---- memory reclamation, then the next search ----
    if (!ended) reclaim_if_short<G>(F, t, H, sm);
    const double u_full = rng.u01();                                                           // MCTS.py:58
    begin_search_from_lds<G>(F, t, H, sm, u_full < F.prob_fullMCTS);
    H.rng_counter = rng.counter;
    if (H.n_nodes > H.max_nodes_seen) H.max_nodes_seen = H.n_nodes;
    // an EXISTING root gets its noise before simulation 0 (MCTS.py:64,156-160)
    if (H.noise_pending && root_noise_tree<G>(F, t, H.root_rec, H.c_sims, nullptr, -1, dense, sm.mask)) H.noise_pending = 0u;
    if (l == 0) F.hdr[t] = H;
}

}  // namespace azg
