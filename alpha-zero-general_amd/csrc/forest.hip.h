// forest.hip.h -- struct-of-arrays MCTS forest in HBM, one 64-lane wavefront per tree.
//
// What the reference does (MCTS.py): nodes_data is a dict keyed by board.tobytes() holding dense per-action arrays
// (Ps f32[A], Qsa f64[A], Nsa i64[A]); every simulation re-plays the env step at every level and re-hashes the child
// state to find the next node (MCTS.py:125-126,164-175).
//
// What this engine keeps in HBM per tree (all private to the tree's wavefront, so no atomics and no cross-workgroup
// visibility protocol are needed inside a launch):
//   record heap          bytes    one contiguous, 16-B aligned RECORD per node -- everything a descent level touches:
//                                   RecHdr 32 B  { Ns u32, Qs f32, node id, n_valid, flags, round, Es f32[4] }
//                                   pages of PC = cls_q VALID-ACTION-COMPACTED entries, struct-of-arrays inside a page:
//                                     hot[PC]      16 B  { P f32, N u32, Q f64 }   -- all a PUCT evaluation reads
//                                     child[U][PC]  4 B  child slot per universe   -- a level reads only the current universe's
//                                     id[PC]        2 B  action id                  -- read when an edge is resolved
//                                 entry j of a record sits at a position that does not depend on n_valid, so a wave
//                                 requests the header, every lane's hot part and child slot in ONE round trip per level:
//                                 20 bytes per lane in two dense runs (16 + 4) instead of a 32-byte array-of-structs entry
//   NodeHdr[cap]         16 B     cold: 64-bit state hash, record offset, n_valid, round, flags (probe / GC / dumps)
//   state[cap][SP]       int8     the node's canonical state = the dict KEY of the reference (full-key verified)
//   htab[HT]             u32      open-addressing table: (10-bit tag | 22-bit node id), probed 64 slots per wave load
//   path[MAXD]           8 B      the descent of the pending simulation (record, entry index, next_player, roll prefix)
//
// child[u] of entry j caches the RECORD OFFSET (| next_player << 29) of the node reached through valid action j in universe u (u = sim index mod universes, the
// reference's seeded-chance mechanism MCTS.py:14,63).  It is pure memoisation of the reference's "replay env step +
// dict lookup": the child of (state, action, seed) is a deterministic function, node identity stays the full state
// (transpositions are found through the hash table exactly like the dict), and nodes are only ever dropped when they
// are unreachable (round < root round), so a cached id can never differ from what the lookup would return.
//
// Numerics follow the shipped (Numba-typed) reference: UCB in f64 with f32 operands widened (MCTS.py:210-230),
// Qsa running mean in f64, Qs in f32 scalar arithmetic (MCTS.py:178-181), no FMA contraction (-ffp-contract=off).
#pragma once
#include <type_traits>
#include "azg_common.hip.h"

namespace azg {

enum : uint32_t { ST_IDLE = 0, ST_SEARCHING = 1, ST_WAIT_NN = 2, ST_DONE = 3,
                  ST_GC = 4,        // self-play: the next search waits for the clean-up kernel (k_gc); cur_pre = min round
                  ST_GC_DONE = 5 }; // cleaned up: k_after_gc begins the search from root_state
enum : uint8_t { NF_TERMINAL = 1, NF_EXPANDED = 2, NF_FREE = 4 };   // NF_FREE: node id on the free stack (NodeHdr only)
enum : uint32_t {
    ERR_NODE_OVERFLOW = 1, ERR_HEAP_OVERFLOW = 2, ERR_DEPTH_OVERFLOW = 4, ERR_BAD_STATE = 8, ERR_EXAMPLE_OVERFLOW = 16,
    ERR_REC_OVERFLOW = 32,
    ERR_EMPTY_POLICY = 64      // every pruned root count is 0: the reference divides 0 / 0 here (MCTS.py:77-80,100-102)
};

// phase cycle counters of k_select (azg_selfplay_stats.cyc_*): compiled in only with -DAZG_CYC_COUNTERS (tools/dbg_cycles.py),
// each s_memtime read costs the wave an lgkmcnt(0) wait
#ifdef AZG_CYC_COUNTERS
#define AZG_CLK() clock64()
#else
#define AZG_CLK() 0ll
#endif

#define AZG_EDGE_UNITS 5
#define AZG_MAXD 256
#define AZG_IDX_BITS 22
#define AZG_IDX_MASK ((1u << AZG_IDX_BITS) - 1u)
#define AZG_CHILD_NP_SHIFT 29            /* child slot = record offset (29 bits of 16-byte units) | next_player << 29 (3 bits: up to 5 players) */
#define AZG_CHILD_IDX_MASK 0x1FFFFFFFu

struct __attribute__((aligned(16))) NodeHdr {          // cold per-node data
    uint64_t hash;
    uint32_t rec_off;      // 16-byte units into the tree's record heap
    uint16_t nv;           // size CLASS index of the record (see ForestDev::cls_q); the node's n_valid is in its RecHdr
    uint8_t round;
    uint8_t flags;
};

struct __attribute__((aligned(16))) RecHdr {           // first 32 bytes of every record
    uint32_t Ns;
    float Qs;
    uint32_t node_id;
    uint16_t nv;
    uint8_t flags;
    uint8_t round;
    union {
        float Es[AZG_MAX_PLAYERS_DEV];     // terminal node: game result (MCTS.py:131-135) of players 0..3; a fifth player's (The Little
                                           // Prince with 5 players) rides in Qs, which a terminal node does not use: rec_es()
        double sq[2];                      // otherwise: sqrt(Ns), sqrt(Ns + EPS) -- kept by the backup so that a descent
    };                                     // level does not recompute two f64 square roots (MCTS.py:213-214)
};

__host__ __device__ inline float rec_es(const RecHdr& rh, int p) { return p < AZG_MAX_PLAYERS_DEV ? rh.Es[p] : rh.Qs; }

struct __attribute__((aligned(16))) TreeHdr {
    uint32_t n_nodes, heap_top, root, status;          // n_nodes = LIVE nodes; heap_top = bump pointer of the record heap;
                                                       // root = node id of the search root (AZG_NONE: not a node yet)
    uint32_t sim_idx, n_sims, is_full, forced;
    uint32_t pending_leaf, path_len, ply, cur_player;
    uint64_t rng_counter;
    uint32_t err, root_round;
    uint32_t leaf_is_root, games_done, step, n_rec;
    uint32_t mid_sim, cur_rec, cur_depth, cur_pre;     // a descent paused by the per-launch level budget
    uint32_t pending_nv, pending_node, pad0_, pad1_;   // n_valid and node id of pending_leaf (saves k_expand_backup a round trip)
    uint32_t id_top, n_free_ids, free_units, max_live; // node ids ever handed out; size of the free-id stack; 16-B units on the
                                                       // record free lists; most nodes that survived a clean-up
    uint32_t max_nodes_seen, gc_runs, root_rec, noise_pending;   // root_rec = record offset of the root;
                                                                 // noise_pending: root Dirichlet noise still to apply
    uint64_t c_sims, c_levels, c_exp, c_sumvalid, c_term, c_depth, c_plies, c_examples;
    uint64_t cyc_select, cyc_levels, cyc_edge, cyc_leaf, cyc_seg[4];   // shader-clock cycles spent in k_select and its phases
};

struct PathEnt {
    uint32_t rec;      // record offset (16-byte units)
    uint16_t j;
    uint8_t np;        // next_player of this edge
    uint8_t pre;       // sum of np over the entries above this one (mod P)
};

struct ForestDev {
    int T, cap, HT, U;                 // trees, nodes per tree, hash slots per tree (pow2), child slots per action
    uint32_t heap_units;               // 16-byte units per tree heap
    // per-tree strides: the sizes rounded up to an ODD multiple of 256 B, so that "the same offset in every tree" (root
    // record, node 0, ...) walks over all HBM channels instead of hitting one (power-of-two strides were ~8 % slower)
    size_t s_heap, s_nstate;           // bytes
    size_t s_nhdr, s_htab;             // elements (NodeHdr, u32)
    size_t s_free, s_recfree;          // elements (u32): free-id stack [cap], record free-list heads [A + 1]
    int universes, numMCTSSims, ratio_fullMCTS, forced_playouts;
    int seed_entries, id_bytes;        // record geometry (RecGeom SE, IDB): set by the host from the game's traits, like cls_q
    int cls_q;                         // record size classes: class 0 = no entries (terminal nodes), class c >= 1 = room for
                                       // min(A, c * cls_q) entries; cls_q == A (small action spaces) makes every expanded
                                       // node's record the same size, so a freed record fits any later node
    int level_budget;                  // max descent levels per tree per k_select launch (0 = unlimited)
    int work_budget;                   // max work units (level = 1, edge resolution = AZG_EDGE_UNITS) per tree per launch
    int spec_state;                    // one-class forests: fetch the node's state with its entries (the frontier edge then finds its parent
                                       // state in registers): 1 = at every level, N >= 2 = at nodes reached over an edge with < N visits
                                       // (default 8), 0 = never, i.e. when an edge is resolved (AZG_SPEC_STATE, A/B runs)
    uint32_t gc_high_water;            // self-play: clean a tree up before its next search once its arena holds more nodes than this (0 = only
                                       // when the arena could not take another search)
    uint32_t episode_quota;            // self-play: total games this forest plays (azg_selfplay_start_ex), 0 = restart forever
    double cpuct, fpu, prob_fullMCTS, dirichletAlpha, temp_begin, temp_end, temp_root, tempThreshold;
    uint64_t rng_seed, stream0;
    int max_examples, max_rec;
    TreeHdr* hdr;
    NodeHdr* node_hdr;
    int8_t* node_state;
    uint8_t* heap;
    uint32_t* htab;
    uint32_t* free_ids;                // [T][cap] stack of reusable node ids
    uint32_t* rec_free;                // [T][A + 1] head of the free list of records with n_valid == index (AZG_NONE = empty);
                                       // a free record's first dword links to the next one
    PathEnt* path;
    int8_t* root_state;
    int8_t* board;
    // per-game record buffers (self-play): [T][max_rec]
    int8_t* rec_board; float* rec_pi; uint8_t* rec_valid; float* rec_q; uint8_t* rec_player; uint16_t* rec_ply;
    // example ring
    int8_t* ex_board; float* ex_pi; float* ex_z; uint8_t* ex_valid; float* ex_q;
    int32_t* ex_meta;                  // [max_examples][4] = (global game stream, game index on that stream, ply, player)
    unsigned long long* ex_count;      // [0] = records written, [1] = dropped (games that did not fit)
};
// the forest's effective RNG seed: cfg.rng_seed, re-keyed by the epoch of azg_selfplay_start_ex (a kernel argument: HIP graphs
// captured before a change of epoch or episode quota must be captured again -- SelfPlayEngine.start does)
__device__ __forceinline__ uint64_t forest_seed(const ForestDev& F) { return F.rng_seed; }

// Wave-uniform reads of MUTABLE global memory (tree / node / record headers, counters, free lists).
// They must never become scalar loads: the compiler turns a uniform-address load of global memory into an s_load whenever it
// sees no earlier store in the kernel, the scalar data cache is not coherent with the vector stores / L2 atomics of earlier
// launches, and HIP-graph replays do not reliably invalidate it between kernel nodes (measured in round 2: a scalarised read of
// TreeHdr.c_sims made self-play results depend on the launch cadence).  A relaxed agent-scope atomic load is never scalarised
// and is served by the L2 (`global_load ... sc1`); `load_uniform_hot` keeps plain vector loads for the descent loop of k_select,
// where the surrounding stores already rule the scalar path out and the loads may merge into dwordx4.
__device__ __forceinline__ uint32_t ld_agent_u32(const uint32_t* p) {
    return uni_u32(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ uint64_t ld_agent_u64(const uint64_t* p) {
    const uint64_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((uint64_t)uni_u32((uint32_t)(v >> 32)) << 32) | uni_u32((uint32_t)v);
}
__device__ __forceinline__ int ld_agent_i32(const int32_t* p) { return (int)ld_agent_u32((const uint32_t*)p); }
template <class T, bool HOT = false>
__device__ __forceinline__ T load_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "word-sized struct");
    const uint32_t* s = (const uint32_t*)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = HOT ? uni_u32(s[k]) : ld_agent_u32(s + k);
    T out;
    __builtin_memcpy(&out, w, sizeof(T));          // (not a cast of &out: writes through uint32_t* may not alias T's fields)
    return out;
}
template <class T>
__device__ __forceinline__ T load_uniform_hot(const T* p) { return load_uniform<T, true>(p); }

static __device__ __constant__ long long AZG_MAGIC_SEEDS[8] = {31416, 1, 14142, 42, 27183, 2, 16180, 7};   // MCTS.py:14

__host__ __device__ __forceinline__ uint32_t align16u(uint32_t x) { return (x + 15u) & ~15u; }

// record geometry: RecHdr | page[pages(nv)], page = hot[PC] (16 B: P f32, N u32, Q f64) | child0[PC] (u32) | childX[U - 1][SE] (u32) | id[PC]
// (IDB bytes each), PC = cls_q entries per page (one-class forests: PC = A, a single page).
// Child slots: universe 0's for every entry, the other universes' only for the first SE entries of a page (round 4).  An entry needs one
// slot per universe only if its env step can read random_seed; a game whose seed-dependent actions have the LOWEST action ids (Splendor:
// the 27 buy / reserve actions that draw a replacement card, of 81) names their count as SEED_ACTIONS -- entries are valid-action-compacted
// in ascending id order, so such an entry always sits at an index below that count.  Entries at or beyond SE use slot 0 in every universe.
// IDB: one byte per action id when A <= 256.  Splendor record: 2464 -> 1952 B.
#define AZG_REC_HDR 32u
#define AZG_CLS_Q_MULTI 32      /* entries per page / size class of multi-class forests (one-class forests: A) */
#define AZG_H_P 0u
#define AZG_H_N 4u
#define AZG_H_Q 8u
struct RecGeom {
    uint32_t PC, U, SE, IDB, PAGE;        // entries per page, universes, entries with per-universe slots, bytes per action id, bytes per page
    __host__ __device__ RecGeom(int pc, int u, int se, int idb)
        : PC((uint32_t)pc), U((uint32_t)u), SE((uint32_t)(se < pc ? se : pc)), IDB((uint32_t)idb),
          PAGE(align16u((uint32_t)pc * (20u + (uint32_t)idb) + ((uint32_t)u - 1u) * (uint32_t)(se < pc ? se : pc) * 4u)) {}
    __host__ __device__ __forceinline__ uint32_t pages(int nv) const { return nv == 0 ? 0u : 1u + ((uint32_t)nv - 1u) / PC; }
    __host__ __device__ __forceinline__ uint32_t total(int nv) const { return AZG_REC_HDR + pages(nv) * PAGE; }      // bytes
    __host__ __device__ __forceinline__ uint32_t page_of(uint32_t j) const { return AZG_REC_HDR + (j / PC) * PAGE; }
    __host__ __device__ __forceinline__ uint32_t hot(uint32_t j) const { return page_of(j) + (j % PC) * 16u; }
    __host__ __device__ __forceinline__ uint32_t child(uint32_t j, uint32_t u) const {
        const uint32_t jj = j % PC;
        return (u == 0u || jj >= SE) ? page_of(j) + PC * 16u + jj * 4u : page_of(j) + PC * 20u + ((u - 1u) * SE + jj) * 4u;
    }
    __host__ __device__ __forceinline__ uint32_t id(uint32_t j) const { return page_of(j) + PC * 20u + (U - 1u) * SE * 4u + (j % PC) * IDB; }
};
// read-only view of a record's action ids (ids[j] = action of valid-action-compacted entry j)
struct RecIds {
    const uint8_t* rec; RecGeom G;
    __host__ __device__ RecIds(const uint8_t* r, const RecGeom& g) : rec(r), G(g) {}
    __host__ __device__ __forceinline__ uint16_t operator[](int j) const {
        const uint8_t* p = rec + G.id((uint32_t)j);
        return G.IDB == 1u ? (uint16_t)*p : *(const uint16_t*)p;
    }
};
// per-game layout parameters (device: Forest<G>; host: the same rule in azg.hip)
template <class G, class = void> struct SeedActions { static constexpr int value = 1 << 30; };
template <class G> struct SeedActions<G, std::void_t<decltype(G::SEED_ACTIONS)>> { static constexpr int value = G::SEED_ACTIONS; };

// NumPy's pairwise float32 summation order (np.sum called by `normalise`, MCTS.py:250-253) for n <= 128 elements,
// executed by lanes 0..7 over an LDS array; every lane returns the sum.
__device__ __forceinline__ float np_sum_block_f32(const float* a, int n) {
    int l = lane_id();
    float res;
    if (n < 8) {
        res = 0.f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    int lim = n - (n % 8);
    float r = 0.f;
    if (l < 8) {
        r = a[l];
        for (int i = 8 + l; i < lim; i += 8) r += a[i];
    }
    float r1 = __shfl_xor(r, 1, 64);
    float s2 = (l & 1) ? (r1 + r) : (r + r1);          // lanes 0,1 -> r0+r1 ; 2,3 -> r2+r3 ...
    float s2o = __shfl_xor(s2, 2, 64);
    float s4 = (l & 2) ? (s2o + s2) : (s2 + s2o);      // (r0+r1)+(r2+r3)
    float s4o = __shfl_xor(s4, 4, 64);
    float s8 = (l & 4) ? (s4o + s4) : (s4 + s4o);
    res = __shfl(s8, 0, 64);
    for (int i = lim; i < n; i++) res += a[i];
    return res;
}

// full pairwise recursion (n > 128 splits in halves rounded down to a multiple of 8)
__device__ inline float np_sum_f32(const float* a, int n) {
    if (n <= 128) return np_sum_block_f32(a, n);
    if (n <= 256) {                             // one split: both halves are leaf blocks (Azul's 180 actions: 88 + 92)
        int n2 = n / 2; n2 -= n2 % 8;
        const float lo = np_sum_block_f32(a, n2);
        return lo + np_sum_block_f32(a + n2, n - n2);
    }
    // iterative traversal of the recursion tree with an explicit stack (depth <= 8 for n <= 32768)
    int st_off[12], st_n[12], st_state[12];
    float st_acc[12];
    int sp = 0;
    st_off[0] = 0; st_n[0] = n; st_state[0] = 0; st_acc[0] = 0.f;
    float ret = 0.f;
    while (sp >= 0) {
        int off = st_off[sp], len = st_n[sp];
        if (len <= 128) { ret = np_sum_block_f32(a + off, len); sp--; continue; }
        int n2 = len / 2; n2 -= n2 % 8;
        if (st_state[sp] == 0) {            // descend left
            st_state[sp] = 1;
            sp++; st_off[sp] = off; st_n[sp] = n2; st_state[sp] = 0;
        } else if (st_state[sp] == 1) {     // left done -> descend right
            st_acc[sp] = ret; st_state[sp] = 2;
            sp++; st_off[sp] = off + n2; st_n[sp] = len - n2; st_state[sp] = 0;
        } else {                            // both done
            ret = st_acc[sp] + ret; sp--;
        }
    }
    return ret;
}

// The same recursion for a length known at compile time (G::A), laid out for the wave: the recursion tree is static, so its leaf blocks
// (<= 128 elements each: 16 for Santorini's 1782 actions, 32 for Abalone / Akropolis) are summed EIGHT AT A TIME -- one per group of eight
// lanes, the eight strided accumulators of a block in the group's lanes, the fixed ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) tree
// by xor-shuffles inside the group, then the n % 8 tail -- and combined left + right along the static tree.  The run-time form above walks
// the blocks one after the other with its stack in private arrays (scratch memory): 81 k cycles of a Santorini-with-gods expansion.
constexpr int np_half(int len) { return len / 2 - (len / 2) % 8; }
template <int LEN> constexpr int np_leaves() {
    if constexpr (LEN <= 128) return 1;
    else return np_leaves<np_half(LEN)>() + np_leaves<LEN - np_half(LEN)>();
}
template <int OFF, int LEN, int IDX>
__device__ __forceinline__ void np_leaf_of(int want, int& off, int& len) {           // (offset, length) of leaf block `want`
    if constexpr (LEN <= 128) { if (want == IDX) { off = OFF; len = LEN; } }
    else {
        np_leaf_of<OFF, np_half(LEN), IDX>(want, off, len);
        np_leaf_of<OFF + np_half(LEN), LEN - np_half(LEN), IDX + np_leaves<np_half(LEN)>()>(want, off, len);
    }
}
template <int OFF, int LEN, int IDX, int R>
__device__ __forceinline__ float np_combine(const float (&leaf)[R]) {
    if constexpr (LEN <= 128) return __shfl(leaf[IDX >> 3], 8 * (IDX & 7), 64);
    else {
        const float lo = np_combine<OFF, np_half(LEN), IDX, R>(leaf);
        const float hi = np_combine<OFF + np_half(LEN), LEN - np_half(LEN), IDX + np_leaves<np_half(LEN)>(), R>(leaf);
        return lo + hi;
    }
}
template <int N>
__device__ __forceinline__ float np_sum_f32_static(const float* a) {
    if constexpr (N <= 256) return np_sum_f32(a, N);
    else {
        constexpr int NL = np_leaves<N>(), R = (NL + 7) / 8;
        const int l = lane_id(), g = l >> 3, j = l & 7;
        float leaf[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            int off = 0, len = 0;
            np_leaf_of<0, N, 0>(8 * r + g, off, len);
            const int lim = len - (len % 8);                                       // (a leaf of the recursion holds at least 8 elements)
            float acc = 0.f;
            if (len > 0) {
                acc = a[off + j];
                for (int i = 8 + j; i < lim; i += 8) acc += a[off + i];
            }
            const float r1 = __shfl_xor(acc, 1, 64);
            const float s2 = (l & 1) ? (r1 + acc) : (acc + r1);
            const float s2o = __shfl_xor(s2, 2, 64);
            const float s4 = (l & 2) ? (s2o + s2) : (s2 + s2o);
            const float s4o = __shfl_xor(s4, 4, 64);
            float res = (l & 4) ? (s4o + s4) : (s4 + s4o);
            res = __shfl(res, 8 * g, 64);                                           // the group's lane 0 holds the tree's value
            for (int i = lim; i < len; i++) res += a[off + i];
            leaf[r] = res;
        }
        return np_combine<0, N, 0, R>(leaf);
    }
}

// The same sums on a policy held in REGISTERS (element i in lane i & 63 of pv[i >> 6], the layout the expansion loads it in): NumPy's
// order -- 8 strided accumulators r[j] = a[j] + a[8 + j] + ..., the fixed tree ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the
// n % 8 tail -- with the strided terms fetched by ds_bpermute (all of them in flight together) instead of eight lanes walking an LDS
// array term by term (2.5 k cycles of the expansion's 11 k for Splendor's 81 actions).  Every lane returns the sum.
template <int OFF, int LEN, int NA>
__device__ __forceinline__ float np_sum_block_regs(const float (&pv)[NA]) {
    static_assert(LEN >= 8 && LEN <= 128 && OFF % 8 == 0 && OFF + LEN <= 64 * NA, "one pairwise leaf block");
    constexpr int LIM = LEN - LEN % 8, K = LIM / 8;
    const int l = lane_id(), j = l & 7;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const float v = __shfl(pv[(OFF + 8 * k) >> 6], ((OFF + 8 * k) & 63) + j, 64);      // a[OFF + 8k + j]: every lane gets its j's term
        r = k == 0 ? v : r + v;
    }
    const float r1 = __shfl_xor(r, 1, 64);
    const float s2 = (l & 1) ? (r1 + r) : (r + r1);
    const float s2o = __shfl_xor(s2, 2, 64);
    const float s4 = (l & 2) ? (s2o + s2) : (s2 + s2o);
    const float s4o = __shfl_xor(s4, 4, 64);
    float res = (l & 4) ? (s4o + s4) : (s4 + s4o);          // every group of 8 lanes holds the same r[0..7]: all lanes end with the full sum
#pragma unroll
    for (int i = LIM; i < LEN; i++)
        res += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[(OFF + i) >> 6]), (OFF + i) & 63));
    return res;
}
template <int N, int NA>
__device__ __forceinline__ float np_sum_regs(const float (&pv)[NA]) {
    static_assert(N >= 8 && N <= 256, "np_sum_regs: one or two leaf blocks");
    if constexpr (N <= 128) return np_sum_block_regs<0, N, NA>(pv);
    else {
        constexpr int N2 = N / 2 - (N / 2) % 8;
        const float lo = np_sum_block_regs<0, N2, NA>(pv);
        return lo + np_sum_block_regs<N2, N - N2, NA>(pv);
    }
}

// Gamma(alpha, 1) variate from a counter-based uniform stream (Marsaglia-Tsang squeeze; alpha < 1 via the
// U^(1/alpha) boost).  Deterministic in (key, ctr): the same call returns the same variate, so callers may recompute
// instead of storing.  Used for the root Dirichlet noise (rng.dirichlet([alpha]*n_valid) == normalised Gammas,
// MCTS.py:187-192) when the caller supplies no noise tensor.
__device__ inline double gamma_variate(double alpha, uint64_t key, uint64_t ctr) {
    auto u01 = [&](uint64_t k) { return ((double)(mix64(key + k) >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
    const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (int it = 0; it < 8; it++) {
        const double u1 = u01(ctr + 4 * it), u2 = u01(ctr + 4 * it + 1), u3 = u01(ctr + 4 * it + 2);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(u3) < 0.5 * x * x + d - d * v + d * log(v)) { g = d * v; break; }
    }
    if (alpha < 1.0) g *= pow(u01(ctr + 3), 1.0 / alpha);
    return g;
}

template <class G>
struct Forest {
    static constexpr int S = G::S, SP = G::SP, A = G::A, P = G::P, AW = G::AW;
    static constexpr int SPW = SP / 4;
    // record size classes are a property of the game (the host sets ForestDev::cls_q by the same rule, azg.hip): entries per page = A for a
    // small action space (one class, one page), 32 otherwise.  As a compile-time constant every j / PC, j % PC of the record geometry is a
    // shift or a multiply-shift instead of a dozen instructions of integer-division emulation (expansion: per valid entry; backup: per level).
    static constexpr int CLS_Q = A <= 96 ? A : AZG_CLS_Q_MULTI;
    static constexpr bool ONE_CLASS = CLS_Q == A;

    struct Smem {
        __attribute__((aligned(16))) int8_t st[SP];
        __attribute__((aligned(16))) int8_t tmp[SP];           // (directly behind st: also the scratch of a game's make_move, MoveScratch<G>)
        static_assert(MoveScratch<G>::value <= SP, "make_move scratch lives in tmp");
        __attribute__((aligned(16))) uint64_t mask[AW];
        __attribute__((aligned(16))) PathEnt path[AZG_MAXD];
    };

    // ---- addressing ----
    __device__ static __forceinline__ NodeHdr* nhdr(const ForestDev& F, int t, uint32_t id) {
        return F.node_hdr + (size_t)t * F.s_nhdr + id;
    }
    __device__ static __forceinline__ int8_t* nstate(const ForestDev& F, int t, uint32_t id) {
        return F.node_state + (size_t)t * F.s_nstate + (size_t)id * SP;
    }
    __device__ static __forceinline__ uint8_t* heap(const ForestDev& F, int t) {
        return F.heap + (size_t)t * F.s_heap;
    }
    __device__ static __forceinline__ uint32_t* htab(const ForestDev& F, int t) { return F.htab + (size_t)t * F.s_htab; }
    __device__ static __forceinline__ uint32_t* rec_free(const ForestDev& F, int t) { return F.rec_free + (size_t)t * F.s_recfree; }
    __device__ static __forceinline__ uint8_t* rec_ptr(const ForestDev& F, int t, uint32_t rec_off) {
        return heap(F, t) + (size_t)rec_off * 16u;
    }

    // States of more than 768 bytes (Botanik 2310, Akropolis 1352, The Little Prince 825) move as 16-byte vectors, every request of a copy
    // issued before the first use: 3 wide requests instead of 10 dependent dword round trips for a Botanik state.
    static constexpr bool BIG_STATE = SPW > 192;
    static constexpr int SPV = SP / 16, NV4 = (SPV + 63) / 64;
    __device__ static __forceinline__ void load_state(int8_t* lds, const int8_t* g_padded) {
        if constexpr (BIG_STATE) {
            const uint4* src = (const uint4*)g_padded;
            uint4* dst = (uint4*)lds;
            uint4 r[NV4];
#pragma unroll
            for (int k = 0; k < NV4; k++) r[k] = lane_id() + 64 * k < SPV ? src[lane_id() + 64 * k] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NV4; k++)
                if (lane_id() + 64 * k < SPV) dst[lane_id() + 64 * k] = r[k];
        } else {
        const uint32_t* src = (const uint32_t*)g_padded;
        uint32_t* dst = (uint32_t*)lds;
        for (int i = lane_id(); i < SPW; i += 64) dst[i] = src[i];
        }
        wave_sync();
    }
    __device__ static __forceinline__ void store_state(int8_t* g_padded, const int8_t* lds) {
        if constexpr (BIG_STATE) {
            const uint4* src = (const uint4*)lds;
            uint4* dst = (uint4*)g_padded;
#pragma unroll
            for (int k = 0; k < NV4; k++)
                if (lane_id() + 64 * k < SPV) dst[lane_id() + 64 * k] = src[lane_id() + 64 * k];
        } else {
        const uint32_t* src = (const uint32_t*)lds;
        uint32_t* dst = (uint32_t*)g_padded;
        for (int i = lane_id(); i < SPW; i += 64) dst[i] = src[i];
        }
    }
    // wave_hash_state (azg_common.hip.h) of the state in LDS: the same sum, read as 16-byte vectors for the big states
    __device__ static __forceinline__ uint64_t hash_state(const int8_t* lds) {
        if constexpr (BIG_STATE) {
            const uint4* src = (const uint4*)lds;
            uint64_t acc = 0;
#pragma unroll
            for (int k = 0; k < NV4; k++) {
                const int i = lane_id() + 64 * k;
                if (i < SPV) {
                    const uint4 x = src[i];
                    acc += (uint64_t)x.x * (uint64_t)(((uint32_t)(4 * i + 1) * 0x9E3779B1u) | 1u) + (uint64_t)x.y * (uint64_t)(((uint32_t)(4 * i + 2) * 0x9E3779B1u) | 1u) +
                           (uint64_t)x.z * (uint64_t)(((uint32_t)(4 * i + 3) * 0x9E3779B1u) | 1u) + (uint64_t)x.w * (uint64_t)(((uint32_t)(4 * i + 4) * 0x9E3779B1u) | 1u);
                }
            }
            return mix64(wave_sum_u64(acc) ^ 0xD6E8FEB86659FD93ULL);
        } else
            return wave_hash_state((const uint32_t*)lds, SPW);
    }
    // unpadded S-byte state (API buffers) -> LDS with zero tail
    __device__ static __forceinline__ void load_state_unpadded(int8_t* lds, const int8_t* g) {
        for (int i = lane_id(); i < SP; i += 64) lds[i] = i < S ? g[i] : (int8_t)0;
        wave_sync();
    }
    __device__ static __forceinline__ void store_state_unpadded(int8_t* g, const int8_t* lds) {
        for (int i = lane_id(); i < S; i += 64) g[i] = lds[i];
    }

    __device__ static __forceinline__ uint32_t tag_of(uint64_t h) { return (uint32_t)(h >> 54); }   // 10 bits

    // Find the node whose key equals the state in LDS (*found_rec = its record offset).  Also returns the first free slot
    // met (for insertion).
    __device__ static uint32_t probe(const ForestDev& F, int t, const int8_t* st_lds, uint64_t h, uint32_t* free_slot,
                                     uint32_t* found_rec) {
        const uint32_t* tab = htab(F, t);
        const uint32_t maskHT = (uint32_t)F.HT - 1u;
        uint32_t slot0 = (uint32_t)h & maskHT;
        const uint32_t tg = tag_of(h);
        const uint32_t* my = (const uint32_t*)st_lds;
        for (int round = 0; round < F.HT / 64 + 1; round++) {
            uint32_t slot = (slot0 + (uint32_t)round * 64u + (uint32_t)lane_id()) & maskHT;
            uint32_t e = tab[slot];
            uint64_t emp = __ballot(e == AZG_NONE);
            uint64_t mat = __ballot(e != AZG_NONE && (e >> AZG_IDX_BITS) == tg);
            int fe = emp ? first_lane(emp) : 64;
            if (fe < 64) mat &= (fe == 0) ? 0ull : (~0ull >> (64 - fe));
            while (mat) {
                int src = first_lane(mat);
                mat &= mat - 1;
                const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)e, src) & AZG_IDX_MASK;
                // hash check and full-key compare requested together (one round trip; tag collisions are rare)
                const uint32_t* other = (const uint32_t*)nstate(F, t, id);
                bool eq = true;
                if constexpr (BIG_STATE) {
#pragma unroll
                    for (int k = 0; k < NV4; k++)
                        if (lane_id() + 64 * k < SPV) {
                            const uint4 a = ((const uint4*)other)[lane_id() + 64 * k], b = ((const uint4*)my)[lane_id() + 64 * k];
                            eq = eq && a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
                        }
                } else
                for (int i = lane_id(); i < SPW; i += 64) eq = eq && (other[i] == my[i]);
                const NodeHdr nh = load_uniform(nhdr(F, t, id));
                if (nh.hash != h) continue;
                if (__all(eq)) { *found_rec = nh.rec_off; return id; }
            }
            if (fe < 64) {
                *free_slot = (slot0 + (uint32_t)round * 64u + (uint32_t)fe) & maskHT;
                return AZG_NONE;
            }
        }
        *free_slot = AZG_NONE;
        return AZG_NONE;
    }

    // What creating a leaf reads from the tree's allocator state -- the top of the free-id stack, the heads of the record free
    // lists and the link word of each head -- requested when the env step of a frontier edge starts, so that the three dependent
    // round trips have landed when create_node / alloc_record need them.  Only this wave changes that state, and a prefetch is
    // consumed by the create_leaf of the same edge, so the values are never stale.
    struct LeafPf {
        uint32_t id_v, head_v, next_v;      // per-lane loaded values: id (same in every lane), head / link of class = lane
        bool lists;                         // the record free lists are covered (several classes, at most 64 of them)
    };
    template <class HS>
    __device__ static __forceinline__ LeafPf leaf_pf_begin(const ForestDev& F, int t, const HS& H) {
        LeafPf pf;
        pf.id_v = H.n_free_ids > 0 ? (F.free_ids + (size_t)t * F.s_free)[H.n_free_ids - 1] : 0u;
        pf.lists = !ONE_CLASS && n_classes(F) <= 64;
        pf.head_v = (pf.lists && lane_id() < n_classes(F)) ? rec_free(F, t)[lane_id()] : AZG_NONE;
        pf.next_v = AZG_NONE;
        return pf;
    }
    __device__ static __forceinline__ void leaf_pf_links(const ForestDev& F, int t, LeafPf& pf) {
        if (pf.lists && pf.head_v != AZG_NONE) pf.next_v = *(const uint32_t*)(heap(F, t) + (size_t)pf.head_v * 16u);
    }

    // Allocate a node for the state in LDS, write key + hash, insert in the table.  Returns AZG_NONE on overflow.
    template <class HS>
    __device__ static uint32_t create_node(const ForestDev& F, int t, HS& H, const int8_t* st_lds, uint64_t h,
                                           uint32_t free_slot, const LeafPf* pf = nullptr) {
        if (free_slot == AZG_NONE) { H.err |= ERR_NODE_OVERFLOW; return AZG_NONE; }
        uint32_t id;
        if (H.n_free_ids > 0) {                                      // reuse the id of a node the clean-up dropped
            H.n_free_ids--;
            id = uni_u32(pf ? pf->id_v : (F.free_ids + (size_t)t * F.s_free)[H.n_free_ids]);     // (k_select only: plain vector load)
        } else {
            if (H.id_top >= (uint32_t)F.cap) { H.err |= ERR_NODE_OVERFLOW; return AZG_NONE; }
            id = H.id_top++;
        }
        H.n_nodes++;
        store_state(nstate(F, t, id), st_lds);
        if (lane_id() == 0) {
            nhdr(F, t, id)->hash = h;
            htab(F, t)[free_slot] = (tag_of(h) << AZG_IDX_BITS) | id;
        }
        return id;
    }

    // ---- record size classes ----
    __device__ static __forceinline__ int cls_of(const ForestDev& F, int nv) { (void)F; return nv == 0 ? 0 : 1 + (nv - 1) / CLS_Q; }
    static constexpr int SEED_E = SeedActions<G>::value < CLS_Q ? SeedActions<G>::value : CLS_Q;      // entries of a page with per-universe slots
    static constexpr int IDB = A <= 256 ? 1 : 2;
    __device__ static __forceinline__ RecGeom geom(const ForestDev& F) { return RecGeom(CLS_Q, F.U, SEED_E, IDB); }
    __device__ static __forceinline__ uint32_t cls_units(const ForestDev& F, int c) {      // class c = c pages
        return (AZG_REC_HDR + (uint32_t)c * geom(F).PAGE) / 16u;
    }
    __device__ static __forceinline__ int n_classes(const ForestDev& F) { (void)F; return 2 + (A - 1) / CLS_Q; }

    // Record for a node with nv valid actions.  Records never move; a dropped node's record sits on the free list of its
    // size class.  The wave reads the heads of classes c .. c+63 in one request and takes the first non-empty one (closest
    // fit; *cls_out = its class, kept in NodeHdr.nv so that the record returns to the right list), else the bump pointer.
    // Entry-less records (terminal nodes) only recycle their own class.  AZG_NONE on overflow.
    template <class HS>
    __device__ static __forceinline__ uint32_t alloc_record(const ForestDev& F, int t, HS& H, int nv, uint32_t node_id,
                                                            int* cls_out, const LeafPf* pf = nullptr) {
        if (ONE_CLASS) {
            // one class for every expanded node: the record slot IS the node id (no list, no load); heap = cap slots
            *cls_out = cls_of(F, nv);
            return node_id * cls_units(F, 1);
        }
        uint32_t* heads = rec_free(F, t);
        const int c = cls_of(F, nv), nc = n_classes(F);
        if (pf && pf->lists) {              // heads and links are in registers (lane = class): the same closest-fit choice
            const int l = lane_id();
            const uint64_t m = __ballot(pf->head_v != AZG_NONE && l < nc && (c > 0 ? l >= c : l == 0));
            if (m) {
                const int src = first_lane(m);
                const uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)pf->head_v, src);
                const uint32_t next = (uint32_t)__builtin_amdgcn_readlane((int)pf->next_v, src);
                if (l == 0) heads[src] = next;
                H.free_units -= cls_units(F, src);
                *cls_out = src;
                return head;
            }
            const uint32_t units = cls_units(F, c);
            if (H.heap_top + units + 256u > F.heap_units) { H.err |= ERR_HEAP_OVERFLOW; return AZG_NONE; }
            const uint32_t off = H.heap_top;
            H.heap_top += units;
            *cls_out = c;
            return off;
        }
        const int idx = c + lane_id();
        const uint32_t v = (idx < nc && (c > 0 || lane_id() == 0)) ? heads[idx] : AZG_NONE;
        const uint64_t m = __ballot(v != AZG_NONE);
        if (m) {
            const int src = first_lane(m);
            const uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
            const uint32_t next = uni_u32(*(const uint32_t*)(heap(F, t) + (size_t)head * 16u));
            if (lane_id() == 0) heads[c + src] = next;
            H.free_units -= cls_units(F, c + src);
            *cls_out = c + src;
            return head;
        }
        const uint32_t units = cls_units(F, c);
        // 256 units (4 KB) of slack: a level's speculative loads may reach two pages past the header of a short record
        if (H.heap_top + units + 256u > F.heap_units) { H.err |= ERR_HEAP_OVERFLOW; return AZG_NONE; }
        const uint32_t off = H.heap_top;
        H.heap_top += units;
        *cls_out = c;
        return off;
    }

    // Lane-parallel value backup along the recorded path (MCTS.py:176-183 unwound): level d belongs to lane d.
    // (Round 4, measured and dropped: carrying the four statistics the descent read at each level (Nsa, Qsa, Ns, Qs) along with the
    // path, so that the backup only stores -- the extra bytes ride on the bandwidth-bound first round trip of every launch and the descent
    // pays two more readlanes per level: -2 % env-steps/s.)
    __device__ static __forceinline__ void backup(const ForestDev& F, int t, const PathEnt* path, int depth, const float* v) {
        if (depth == 0) return;
        int tot = 0;
        {
            const PathEnt last = path[depth - 1];
            tot = (last.pre + last.np) % P;
        }
        uint8_t* hp = heap(F, t);
        const RecGeom RG = geom(F);
        for (int base = 0; base < depth; base += 64) {
            int d = base + lane_id();
            if (d < depth) {
                PathEnt e = path[d];
                int roll = ((tot - e.pre) % P + P) % P;            // sum of next_player over levels >= d
                const int vi = ((0 - roll) % P + P) % P;            // np.roll(v, n)[0] = v[(-n) mod P]
                // select chain over opaque register copies: a dynamically indexed v[] (or the load-of-select the optimiser
                // makes of a plain select chain) would force the caller's whole value block into scratch memory
                float v0 = v[0];
                asm volatile("" : "+v"(v0));
#pragma unroll
                for (int p = 1; p < P; p++) {
                    float vp = v[p];
                    asm volatile("" : "+v"(vp));
                    v0 = vi == p ? vp : v0;
                }
                uint8_t* rec = hp + (size_t)e.rec * 16u;
                RecHdr* rh = (RecHdr*)rec;
                uint8_t* ent = rec + RG.hot(e.j);
                const uint32_t n = *(uint32_t*)(ent + AZG_H_N), ns = rh->Ns;
                const double q = *(double*)(ent + AZG_H_Q);
                const float qs = rh->Qs;
                *(double*)(ent + AZG_H_Q) = ((double)n * q + (double)v0) / (double)(n + 1u);
                float tq = (float)(ns + 1u) * qs;
                tq = tq + v0;
                rh->Qs = tq / (float)(ns + 2u);
                *(uint32_t*)(ent + AZG_H_N) = n + 1u;
                rh->Ns = ns + 1u;
                rh->sq[0] = sqrt((double)(ns + 1u));
                rh->sq[1] = sqrt((double)(ns + 1u) + AZG_EPS);
            }
        }
    }

    // softmax(Ps, T) + applyDirNoise + normalise on a DENSE policy in LDS (MCTS.py:147-150,156-160,187-197,255-261).
    // `noise` holds, for this tree, either a Dirichlet sample over the valid actions (normalised == true; parity tests
    // inject the reference's sample) or iid Gamma(alpha, 1) variates that are normalised over the first n_valid
    // entries here (Dirichlet(alpha) == normalised Gammas; rng.dirichlet MCTS.py:189).
    __device__ static void root_noise_dense(float* dense, const uint64_t* mask, double temp_root, const double* noise,
                                            bool normalised, double alpha = 0.0, uint64_t gkey = 0, uint64_t gctr = 0) {
        int l = lane_id();
        if (temp_root != 1.0) {
            // Numba typing: float32 array ** float64 -> float64 array, normalised in f64, cast to f32
            double s = 0.0;
            for (int i = l; i < A; i += 64) s += pow((double)dense[i], 1.0 / temp_root);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) s += shfl_xor_f64(s, m);
            for (int i = l; i < A; i += 64) dense[i] = (float)(pow((double)dense[i], 1.0 / temp_root) / s);
            wave_sync();
        }
        int nv = 0;
#pragma unroll
        for (int k = 0; k < AW; k++) nv += __popcll(mask[k]);
        if (noise == nullptr && alpha < 0.0) alpha = 10.0 / (double)nv;            // automatic value MCTS.py:190-192
        double gsum = 1.0;
        if (!normalised) {
            gsum = 0.0;
            for (int r = l; r < nv; r += 64) gsum += noise ? noise[r] : gamma_variate(alpha, gkey, gctr + ((uint64_t)r << 6));
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) gsum += shfl_xor_f64(gsum, m);
            if (!(gsum > 0.0)) gsum = 1.0;
        }
        // dir_values are indexed by the rank of the valid action
        for (int i = l; i < A; i += 64) {
            uint64_t w = mask[i >> 6];
            if ((w >> (i & 63)) & 1) {
                int rank = __popcll(w & ((1ull << (i & 63)) - 1ull));
                for (int k = 0; k < (i >> 6); k++) rank += __popcll(mask[k]);
                float a = 0.75f * dense[i];
                double d = normalised ? noise[rank]
                                      : (noise ? noise[rank] : gamma_variate(alpha, gkey, gctr + ((uint64_t)rank << 6))) / gsum;
                dense[i] = (float)((double)a + 0.25 * d);
            }
        }
        wave_sync();
        float s = np_sum_f32_static<A>(dense);
        for (int i = l; i < A; i += 64) dense[i] = dense[i] / s;
        wave_sync();
    }
};

}  // namespace azg
