// azg_async.hip.h -- (included at the end of azg_nn.hip) the ASYNCHRONOUS TREE PIPELINE: games search while other games sit in the net.
//
// What the reference does (Coach.py:117-144, GenericNNetWrapper.py:122-157): N game threads share one inference server through a lock ring --
// a game that has reached a leaf waits for the batch, the others keep searching.  The two-kernel round of this engine (k_select ->
// k_v80_net_h2) has a LAUNCH-WIDE boundary instead: all T trees descend, the grid drains (the median descent wave is done
// after 19 us, the launch after 32-36), all T leaves go through the net, the grid drains.  Round 4 measured every barriered form of overlap
// (stream groups, XCD pinning, the per-CU round kernel) and none won.  Here there is no boundary wider than one tree:
//
//   * k_async_select: `n_sel` persistent 16-wave workgroups.  Workgroup g OWNS trees g, g + n_sel, g + 2 n_sel, ... for the whole launch
//     (so everything a tree keeps in HBM is only ever touched from ONE CU: plain loads / stores stay coherent through that CU's L1 and its
//     XCD's L2 -- no fence per hand-over, which would cost a whole-L1 invalidate / a whole-L2 write-back each, MI355X_MICROARCH.md).  A free
//     wave claims one of the workgroup's READY trees (ready word per tree in HBM, claim bit in LDS), runs select_tree -- the expansion +
//     backup of the tree's evaluated leaf, then the next descent: the same device function as k_select -- and pushes the new leaf on the
//     LEAF RING.
//   * k_async_net: `n_net` persistent 12-wave workgroups, each the V80 forward of k_v80_net_h2<12> (168 VGPRs, 156 KB LDS: one per CU).
//     A workgroup claims up to 16 leaves from the ring (16 when there are 16; what is there after `batch_wait` of waiting), evaluates
//     them, writes pi / v at the TREES' rows and marks the 16 trees ready.
//   Both kernels are resident at the same time on two streams; a select workgroup (16 waves x <= 128 VGPRs = every VGPR of its CU) and a
//   net workgroup (504 of a SIMD's 512 VGPRs) can never share a CU, so n_sel + n_net <= CUs makes every workgroup resident whatever the
//   order of dispatch -- the CU partition round 4 could not get from CU masks falls out of occupancy.
//   Every tree runs `rounds` (descent, forward) pairs per launch, then waits for the launch to end (k_selfplay_advance runs between
//   launches, at the cadence the two-kernel rounds have).  Per tree the sequence of select_tree calls and the pi / v each one sees are
//   those of the two-kernel rounds, so results are identical bit for bit (tests/test_gpu_selfplay.py); nothing depends on which leaves
//   share a forward (rows of the MFMA tiles are independent).
//
// Hand-over protocol (MI355X: 8 XCDs with private L2s, per-CU L1s that other CUs' stores never refresh; cdna_hip_programming.md G16):
//   payload that crosses CUs -- the leaf record (state + valid bit mask), pi, v -- is stored WRITE-THROUGH (relaxed agent-scope atomic
//   stores = `global_store ... sc1`) by the producer, every storing wave drains (`s_waitcnt vmcnt(0)`), then ONE lane publishes the
//   queue word (agent-scope atomic); the consumer polls that word relaxed and reads the payload with agent-scope loads (past its L1).
//   No fence anywhere.  Queue words: ring entries carry a lap tag (never reset), ready words are owned by one select workgroup.
//   Every spin loop is bounded: a wave that finds nothing to do for `timeout_ticks` of the 100 MHz wall clock sets ctl->abort (everyone
//   leaves) and error bit ERR_ASYNC_TIMEOUT on tree 0 -- a pipeline that cannot make progress (a workgroup not resident) fails loudly
//   instead of hanging.
//
// Recovery (round 6).  The two kernels depend on every workgroup running.  On the MI355X boxes of this project a launch is, about once per 45 s
// of pipeline time, hit by an event that freezes the whole chip for ~1 ms and after which a few workgroups (seen: three net workgroups) do
// not run again until other workgroups leave (tools/dbg_async_stall.py: their batch counters stop, they still hold the ticket ranges they
// had claimed, they resume the moment the launch winds down) -- a preemption by the platform, not something the kernels do.  The leaves queued
// into those ranges are never evaluated, their trees never come back, the launch cannot end: rounds 5 and 6 saw it as the 20 s "time-out"
// (error bit 128).  Such a launch now ENDS EARLY AND SAFELY instead: a wave that finds
// nothing to do for `timeout_ticks` (50 ms) raises the abort flag, everyone leaves, and nothing is lost -- every tree records whether the leaf
// it queued last has been evaluated (`evald`: cleared by the descent wave in front of the ticket, set by the net workgroup with the hand-back),
// and the next launch begins with a small kernel (k_async_requeue) that puts the leaf of every tree still owed its evaluation back on the ring (the
// leaf record is still in the pipeline's leaf array) and starts that tree as "in the net" instead of expanding it with a policy that never arrived.  Eight launches in a row that end this way set the error bit after all (a pipeline that
// really cannot run -- a workgroup that never becomes resident -- still fails loudly).  With the work-sharing budget that is all: an early end
// is a launch that did fewer calls.  With per-tree budgets (shared_budget == 0: "exactly `rounds` calls per tree") every tree also keeps
// the calls it has not run (`carry`: rounds + carry at the start of a launch, the calls left at every hand-over, 0 when the tree retires), and the
// entry point launches the pipeline twice -- the second launch grants nothing, it runs what the first left over -- so the contract holds at
// every return.
#pragma once
#include "azg_fused.hip.h"
#include "selfplay.hip.h"

namespace azg {

constexpr uint32_t ERR_ASYNC_TIMEOUT = 128;
constexpr int ASYNC_RING_LAPS = 8;            // ring slots = pow2 >= ASYNC_RING_LAPS x T: see the ring entry in k_async_select
#ifndef AZG_ASYNC_SEL_WAVES
#define AZG_ASYNC_SEL_WAVES 16                /* waves of a descent workgroup: 16 (<= 128 VGPRs each) fills a CU; 12 (<= 168 VGPRs) measured in round 5 */
#endif
constexpr int ASYNC_SEL_WAVES = AZG_ASYNC_SEL_WAVES;
#ifndef AZG_ASYNC_SCOUTS
#define AZG_ASYNC_SCOUTS 1                  /* waves of a descent workgroup that may poll the ready words at the same time */
#endif
constexpr uint32_t ASYNC_IDLE_STEP_CAP = 100000u;          // 1 ms in 10-ns ticks: what one pass of a waiting loop can add to its idle time
#ifndef AZG_IDLE_SLEEP
#define AZG_IDLE_SLEEP 16                   /* s_sleep of a wave that found nothing ready and is not the scout (units of 64 cycles) */
#endif
#ifndef AZG_SCOUT_SLEEP
#define AZG_SCOUT_SLEEP 8                   /* s_sleep of the scout between two polls of the ready words that found nothing */
#endif
#ifndef AZG_V80_NET_SHARE_256
#define AZG_V80_NET_SHARE_256 136           /* default net share of the CUs for Splendor 2 players + V80, in 256ths */
#endif
constexpr int ASYNC_RS = 128;                 // ready words per select workgroup (trees per workgroup <= 128: two ballots)
constexpr int ASYNC_NPROF = 96;

struct AsyncCtl {                              // zeroed by the host before every launch; hot words on lines of their own
    uint32_t leaf_tail; uint32_t pad0[31];     // leaf tickets issued (producers: descent waves)
    uint32_t leaf_head; uint32_t pad1[31];     // leaf tickets claimed (consumers: net workgroups)
    uint32_t retired; uint32_t abort; uint32_t pad2[30];   // trees that are done with this launch; != 0: leave now (1 select, 2 net, 3 ring)
    unsigned long long calls; uint32_t stop; uint32_t pad3[29];   // work-sharing budget: select_tree calls so far (flushed per workgroup in
};                                                                 // blocks of 32); 1 once they reach AsyncArgs.total_calls: trees retire

// profile counters (accumulated over launches, wall-clock ticks of 10 ns):
//  0 select_tree calls   1 ticks inside them        2 ticks a descent wave spent looking for a ready tree
//  3 net batches         4 leaves in them           5 ticks inside the forward    6 ticks a net workgroup spent waiting for leaves
//  7 sum of (leaf claimed - leaf pushed)            8 sum of (tree claimed - tree marked ready)
//  9 launches           10 select workgroup-ticks resident   11 net workgroup-ticks resident
// 12 n_sel  13 n_net (filled by the host)   14 shader-clock cycles inside the forwards   15 inside the descents   16 plies advanced in-kernel
// 17 / 18 launches that ended early because a descent wave / a net workgroup gave up (time-out); 19 such launches in a row;
// 26 ticket ranges a net workgroup abandoned because the ring had lapped it; 27 leaves re-queued at the start of a launch (owed by one that ended early)
// 32..63 histogram of the leaf wait in us (last bucket: >= 31)    64..95 histogram of the ready wait
struct AsyncArgs {
    ForestDev F;
    H2Weights W;                               // net V80 (Splendor 2 players): nn_v80_h2.hip.h
    Conv5NetW C5; float c5_descale; int c5_pad; // net V89 (Santorini no-gods): nn_conv5x5.hip.h
    Mb1dNetW MB;                               // the MobileNet-1d family (Splendor 3 / 4 players, Azul): nn_mb1d.hip.h
    int8_t* aleaf; uint8_t* leaf_valid; uint8_t* needs_eval; float* pi; float* v;
    AsyncCtl* ctl; unsigned long long* ring; uint32_t* ready; uint32_t* ts_ready;
    unsigned long long* prof;
    uint32_t* evald;                           // [T]: 1 = the leaf the tree queued last has been evaluated (its pi / v are in place); 0 = still owed (see "Recovery")
    uint8_t* pending;                          // [T]: 1 = k_async_requeue put the tree's owed leaf back on the ring: it starts this launch waiting for the net
    uint32_t* carry;                           // [T] (per-tree budgets): the calls the tree still has to run -- what a launch that ended early left over is added to the next launch's `rounds`
    unsigned long long* wginfo;                // [n_sel + n_net][4]: where the workgroup ran (XCC | cu << 8 | se << 16 | sh << 24), role, calls, busy shader cycles
    int noise, rounds, n_sel, ring_bits, batch_wait, timeout_ticks;
    unsigned long long total_calls;            // != 0: the launch ends when the trees TOGETHER have had this many calls (whichever tree is fast
};                                             // gets more of them: no tree waits for the slowest at the end of a launch); 0: `rounds` calls per tree
__device__ __forceinline__ uint32_t where_am_i() {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));              // HW_REG_XCC_ID
    asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hw));               // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    return (xcc & 0xFu) | (((hw >> 8) & 0xFu) << 8) | (((hw >> 13) & 0x7u) << 16) | (((hw >> 12) & 1u) << 24);
}
typedef const AsyncArgs __attribute__((address_space(4))) * AsyncArgsC;

__device__ __forceinline__ uint32_t aload(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t wall32() { return (uint32_t)wall_clock64(); }
// a wave-uniform 64-bit value (an LDS word every lane loaded from the same address) moved to a SCALAR register pair: the claim loop's
// tree masks then live, and are combined, on the scalar unit -- as vector pairs the allocator spilled them to scratch around the inlined
// descent (k_async_select<SplendorDev<2>>: ~12 scratch loads / stores on the claim path, each a vmcnt(0) round trip through the CU's busy
// memory queue between "a tree is ready" and "a wave works on it")
__device__ __forceinline__ unsigned long long uni_u64(unsigned long long v) {
    return ((unsigned long long)uni_u32((uint32_t)(v >> 32)) << 32) | (unsigned long long)uni_u32((uint32_t)v);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifdef AZG_ASYNC_PART_SELECT     /* the descent kernel: azg_async_sel.hip */
// The forest descriptor reaches the persistent kernel through a device buffer (constant address space), not as a kernel argument: the
// compiler then knows nothing about the pointers inside it and every access of the descent became a FLAT instruction (45 flat_load,
// 59 flat_store in k_async_select<SplendorDev<2>>: 64-bit VALU address arithmetic per access, no scalar-base addressing, and a wait on
// BOTH counters -- vmcnt and lgkmcnt -- at every use, which ties the level loop's memory round trip to its LDS traffic).  A round trip
// through the global address space tells it what a by-value kernel argument would have: the same accesses are global_load / global_store.
// Which pointers: the gain is the level loop's and the edge resolution's (heap, node states, hash table, node headers, path, free lists);
// the per-game trait picks the set -- for the multi-class forests (Santorini, Azul) a GLOBAL record heap makes the register allocator spill
// vector registers inside the level loop (k_async_select<SantoriniDev<1>>: scratch instructions 70 -> 185 with the heap alone, 203 with
// everything; measured: a descent 33 -> 47 us, Azul 38 -> 50 us); every other pointer leaves the spill count as it is but gains nothing
// (Santorini 34.1 -> 36.0 us per descent, Azul 36.8 -> 36.4): those two kernels keep their generic pointers.
template <class T>
__device__ __forceinline__ T* as_global(T* p) { return (T*)(T __attribute__((address_space(1)))*)(uintptr_t)p; }
enum : uint32_t { AG_HEAP = 1, AG_STATE = 2, AG_HTAB = 4, AG_NHDR = 8, AG_HDR = 16, AG_ALLOC = 32, AG_PATH = 64, AG_ROOT = 128, AG_IO = 256, AG_ALL = 511 };
#ifndef AZG_AG_DEFAULT
#define AZG_AG_DEFAULT AG_ALL
#endif
#ifndef AZG_AG_MULTI
#define AZG_AG_MULTI 0                      /* multi-class forests: nothing (measured, see above) */
#endif
template <class G> struct AsyncGlobalMask { static constexpr uint32_t value = Forest<G>::ONE_CLASS ? (uint32_t)(AZG_AG_DEFAULT) : (uint32_t)(AZG_AG_MULTI); };
template <uint32_t M>
__device__ __forceinline__ ForestDev forest_as_global(ForestDev F) {
    if (M & AG_HEAP) F.heap = as_global(F.heap);
    if (M & AG_STATE) F.node_state = as_global(F.node_state);
    if (M & AG_HTAB) F.htab = as_global(F.htab);
    if (M & AG_NHDR) F.node_hdr = as_global(F.node_hdr);
    if (M & AG_HDR) F.hdr = as_global(F.hdr);
    if (M & AG_ALLOC) { F.free_ids = as_global(F.free_ids); F.rec_free = as_global(F.rec_free); }
    if (M & AG_PATH) F.path = as_global(F.path);
    if (M & AG_ROOT) { F.root_state = as_global(F.root_state); F.board = as_global(F.board); }
    return F;
}
// LDS control block of a select workgroup
struct AsyncSelLds {
    unsigned long long claimed[2];             // bit i: tree i of this workgroup is being handled by one of its waves
    unsigned long long seen[2];                // bit i: the last poll of the ready words (by the scout wave) found tree i ready
    uint32_t retired;                          // trees of this workgroup that are done with the launch
    uint32_t cursor;                           // rotating start of the scan (fairness)
    uint32_t scout;                            // 1: a wave is polling the ready words in HBM (ONE poller per CU; the other idle waves watch `seen`)
    uint32_t calls;                            // select_tree calls of this workgroup (every 32nd one adds 32 to ctl->calls)
    uint32_t stop;                             // the scout's copy of ctl->stop
    uint32_t pad[3];
    unsigned long long prof[6];                // calls, busy ticks, idle ticks, ready-wait sum, shader cycles inside the descents, plies advanced
    uint32_t hist[32];
    uint32_t idle_acc[16], idle_last[16];      // per wave: time spent looking for work since it last handled a tree / its last clock read
    uint32_t rw[ASYNC_RS];                     // the scout's snapshot of the ready words (calls left + 1; 0 = not ready)
    uint32_t rts[ASYNC_RS];                    // ... and of the time stamps the net wrote beside them (profile: how long a ready tree waits)
    uint32_t last[ASYNC_RS];                   // the ready word this workgroup consumed last for tree i: a tree's words DECREASE over a launch,
};                                             // so a snapshot is current iff it is smaller -- no second look at HBM before a claim

// What the two-kernel rounds do BETWEEN rounds (azg_selfplay_advance: k_selfplay_advance -> k_gc -> k_after_gc -> k_root_noise), for one
// tree, on the descent wave that found the search finished or the fresh root waiting for its noise: the tree goes on at once instead of
// sitting out the rest of a launch.  Out of line: ~once per 800 calls of a tree, and the f64 pow / log / cos of the temperature and of the
// Gamma sampler must not weigh on the descent loop's registers.  Returns true when the tree is searching again.
template <class G>
__device__ __noinline__ bool async_between_calls(const AsyncArgs* args, int t, uint8_t* mine /* this wave's LDS block */) {
    using FR = Forest<G>;
    using RL = RoundLds<G>;
    const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
    const ForestDev F = load_const(&A->F);
    typename FR::Smem& sm = *(typename FR::Smem*)mine;
    float* const dense = (float*)(mine + RL::DENSE_OFF);
    static_assert(sizeof(sm.path) >= sizeof(double) * G::A && sizeof(sm.path) >= sizeof(uint32_t) * (G::A + 8), "the path block doubles as scratch here");
    const int l = lane_id();
    uint32_t status = ld_agent_u32(&F.hdr[t].status);
    if (status == ST_DONE && !ld_agent_u32(&F.hdr[t].err)) {
        advance_tree<G>(F, t, sm, (int*)dense, (double*)sm.path, dense);
        wave_sync();
        status = ld_agent_u32(&F.hdr[t].status);
        if (status == ST_GC) {
            // the clean-up on this one wave (k_gc gives it sixteen; here the other fifteen are descending other trees)
            TreeHdr H = load_uniform(&F.hdr[t]);
            uint32_t* head = (uint32_t*)sm.path;
            gc_scan<G, true>(F, t, H, (int)H.cur_pre, head, head + G::A + 1, 0, 1);
            if (l == 0) {
                TreeHdr* Hp = &F.hdr[t];
                Hp->n_free_ids = H.n_free_ids; Hp->n_nodes = H.n_nodes; Hp->max_live = H.max_live; Hp->free_units = H.free_units;
                Hp->gc_runs = H.gc_runs; Hp->status = ST_GC_DONE;
            }
            drain_vmem();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // the table was re-filled with L2 atomics: drop this CU's L1 copy of it
            after_gc_tree<G>(F, t, sm, dense);
            wave_sync();
        }
    }
    if (A->noise && F.dirichletAlpha != 0.0 && ld_agent_u32(&F.hdr[t].noise_pending)) {
        const uint32_t root_rec = ld_agent_u32(&F.hdr[t].root_rec);
        const uint64_t c_sims = ld_agent_u64(&F.hdr[t].c_sims);
        if (root_noise_tree<G>(F, t, root_rec, c_sims, nullptr, -1, dense, sm.mask) && l == 0) F.hdr[t].noise_pending = 0u;
        wave_sync();
    }
    drain_vmem();
    return ld_agent_u32(&F.hdr[t].status) == ST_SEARCHING && !ld_agent_u32(&F.hdr[t].err);
}

template <class G>
__global__ __launch_bounds__(ASYNC_SEL_WAVES * 64) void k_async_select(const AsyncArgs* args) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using RL = RoundLds<G>;
    AsyncSelLds* const C = (AsyncSelLds*)(lds + ASYNC_SEL_WAVES * RL::STRIDE);
    const int g = (int)blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t t_begin = wall32();
    int n_g, n_sel, timeout;
    uint32_t* my_ready;
    {
        const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
        n_sel = A->n_sel; timeout = A->timeout_ticks;
        const int T = A->F.T, rounds = A->rounds;
        n_g = g < T ? (T - g + n_sel - 1) / n_sel : 0;
        my_ready = A->ready + (size_t)g * ASYNC_RS;
        // every tree of this workgroup starts the launch ready, with `rounds` calls to go (ready word = calls left + 1)
        const int i = (int)threadIdx.x;
        // (a tree whose owed leaf k_async_requeue has put back on the ring starts the launch waiting for the net: see "Recovery")
        unsigned long long start0 = 0ull, start1 = 0ull;
        if (i < ASYNC_RS) {
            const bool here = i < n_g && !A->pending[g + i * n_sel];
            // (per-tree budgets: what an earlier launch that ended early left of the tree's calls is run now, see "Recovery")
            const uint32_t calls = (uint32_t)rounds + ((i < n_g && !A->total_calls) ? A->carry[g + i * n_sel] : 0u);
            if (here && !A->total_calls) A->carry[g + i * n_sel] = calls;
            const uint32_t w0 = here ? calls + 1u : 0u;
            astore(my_ready + i, w0);
            C->rw[i] = w0; C->last[i] = 0xFFFFFFFFu;
            const unsigned long long bal = __ballot(here);
            if (i < 64) start0 = bal; else start1 = bal;
        }
        if (i < ASYNC_RS) { astore(A->ts_ready + (size_t)g * ASYNC_RS + i, t_begin); C->rts[i] = t_begin; }
        if (i == 0) {
            C->claimed[0] = C->claimed[1] = 0ull; C->retired = 0u; C->cursor = 0u; C->scout = 0u; C->calls = 0u; C->stop = 0u;
            for (int k = 0; k < 6; k++) C->prof[k] = 0ull;
            C->seen[0] = start0;                                               // every tree starts ready (but those waiting for the net)
        }
        if (i == 64) C->seen[1] = start1;
        if (i < 32) C->hist[i] = 0u;
        drain_vmem();
    }
    __syncthreads();
    uint8_t* const mine = lds + wave * RL::STRIDE;
    typename RL::Smem& sm = *(typename RL::Smem*)mine;
    float* const dense = (float*)(mine + RL::DENSE_OFF);
    uint32_t idle_since = wall32();                 // (profile sums live in the LDS block: nothing but these words are carried around the loop)
    // The idle time-out counts the time this wave has itself SPENT looking for work: the clock advance between two passes of its loop,
    // capped at 1 ms per pass.  A wave that was not running -- the whole GPU held up for seconds by a driver operation (a 200 GB forest of
    // the previous engine being unmapped), a clock that jumped -- has not been idle, and must not abort the launch: one full-size test in
    // eight did, with the plain difference of two clock reads.
    // (The two words live in the LDS, AsyncSelLds::idle_acc / idle_last: as registers carried around the loop they cost the descent 0.8 us.)
    if (lane_id() == 0) { C->idle_acc[wave] = 0u; C->idle_last[wave] = idle_since; }
#define AZG_LDS_LD64(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define AZG_LDS_LD32(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#pragma unroll 1
    for (;;) {
        // (the lane id is taken afresh -- lane_id() is opaque in this translation unit -- wherever it is used: as ONE value for the whole
        // iteration it was live across the inlined descent, the allocator spilled it at the top of the loop and re-loaded it from scratch in
        // front of every use on the claim path: six vmcnt(0) round trips through the CU's busy memory queue between "a tree is ready"
        // and "a wave works on it")
#define AZG_L lane_id()
        if (AZG_LDS_LD32(&C->retired) >= (uint32_t)n_g) break;
        // ---- look for a ready tree of this workgroup that no wave is handling: in the LDS copy of the ready words first ----
        unsigned long long c0 = uni_u64(AZG_LDS_LD64(&C->seen[0]) & ~AZG_LDS_LD64(&C->claimed[0]));
        unsigned long long c1 = uni_u64(AZG_LDS_LD64(&C->seen[1]) & ~AZG_LDS_LD64(&C->claimed[1]));
        if (!(c0 | c1)) {
            // nothing known to be ready: ONE wave of the workgroup (the scout) polls the ready words in HBM, the others sleep on the LDS copy
            uint32_t got = 0u;
            if (AZG_L == 0) {
                // (the constant comes out of an opaque scalar: the compiler otherwise keeps ONE vector register holding 1 for the whole kernel,
                // spills it around the descent and re-loads it from scratch here, in front of every poll)
                uint32_t one;
                asm volatile("s_mov_b32 %0, 1" : "=s"(one));
                if (AZG_ASYNC_SCOUTS == 1) got = __hip_atomic_exchange(&C->scout, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u ? 1u : 0u;
                else {          // up to AZG_ASYNC_SCOUTS polls in flight: a poll's answer is ~2 us old, a second one started meanwhile halves the gap
                    got = __hip_atomic_fetch_add(&C->scout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)AZG_ASYNC_SCOUTS ? 1u : 0u;
                    if (!got) __hip_atomic_fetch_sub(&C->scout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (!uni_u32(got)) {                        // (the clock is read on passes that found nothing only: never in front of a claim)
                if (AZG_L == 0) {
                    const uint32_t now = wall32(), d = now - C->idle_last[wave];
                    C->idle_acc[wave] += d < ASYNC_IDLE_STEP_CAP ? d : ASYNC_IDLE_STEP_CAP;
                    C->idle_last[wave] = now;
                }
                __builtin_amdgcn_s_sleep(AZG_IDLE_SLEEP);
                continue;
            }
            const uint32_t v0 = AZG_L < n_g ? aload(my_ready + AZG_L) : 0u;
            const uint32_t v1 = AZG_L + 64 < n_g ? aload(my_ready + 64 + AZG_L) : 0u;
            {
                const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
                const uint32_t* my_ts = A->ts_ready + (size_t)g * ASYNC_RS;
                const uint32_t s0 = AZG_L < n_g ? aload(my_ts + AZG_L) : 0u, s1 = AZG_L + 64 < n_g ? aload(my_ts + 64 + AZG_L) : 0u;
                __hip_atomic_store(&C->rts[AZG_L], s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&C->rts[64 + AZG_L], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (A->total_calls) {
                    const uint32_t st = uni_u32(aload(&A->ctl->stop));
                    if (st && AZG_L == 0) __hip_atomic_store(&C->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // current = smaller than the word this workgroup consumed last for the tree (a tree's ready words decrease over a launch)
            const unsigned long long r0 = __ballot(v0 != 0u && v0 < AZG_LDS_LD32(&C->last[AZG_L]));
            const unsigned long long r1 = __ballot(v1 != 0u && v1 < AZG_LDS_LD32(&C->last[64 + AZG_L]));
            __hip_atomic_store(&C->rw[AZG_L], v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&C->rw[64 + AZG_L], v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wave_sync();
            if (AZG_L == 0) {
                __hip_atomic_store(&C->seen[0], r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&C->seen[1], r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            c0 = r0 & ~uni_u64(AZG_LDS_LD64(&C->claimed[0]));
            c1 = r1 & ~uni_u64(AZG_LDS_LD64(&C->claimed[1]));
            bool leave = false;
            if (!(c0 | c1)) {
                const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
                const uint32_t now = wall32();
                uint32_t acc = 0u;
                if (AZG_L == 0) {
                    const uint32_t d = now - C->idle_last[wave];
                    acc = C->idle_acc[wave] + (d < ASYNC_IDLE_STEP_CAP ? d : ASYNC_IDLE_STEP_CAP);
                    C->idle_acc[wave] = acc; C->idle_last[wave] = now;
                }
                acc = uni_u32(acc);
                if (uni_u32(aload(&A->ctl->abort))) leave = true;
                else if (acc > (uint32_t)timeout) {                        // (this wave has looked for work for that long and found none)
                    // the launch ends early and the next one carries on (see "Recovery"); eight such launches in a row: an error
                    if (AZG_L == 0) {
                        astore(&A->ctl->abort, 1u);
                        if (atomicAdd(A->prof + 19, 1ull) >= 7ull) atomicOr(&A->F.hdr[0].err, ERR_ASYNC_TIMEOUT);
                    }
                    leave = true;
                }
                if (leave && AZG_L == 0) __hip_atomic_store(&C->retired, 0x7FFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // everyone out
                if (!leave) __builtin_amdgcn_s_sleep(AZG_SCOUT_SLEEP);
            }
            if (AZG_L == 0) {
                if (AZG_ASYNC_SCOUTS == 1) __hip_atomic_store(&C->scout, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_sub(&C->scout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (leave) break;
            if (!(c0 | c1)) continue;
        }
        // the first candidate at or after the rotating cursor
        const uint32_t cur = uni_u32(AZG_LDS_LD32(&C->cursor)) & 127u;
        const unsigned long long m0 = cur < 64u ? c0 & (~0ull << cur) : 0ull, m1 = cur < 64u ? c1 : c1 & (~0ull << (cur - 64u));
        int i;
        if (m0) i = __builtin_ctzll(m0);
        else if (m1) i = 64 + __builtin_ctzll(m1);
        else if (c0) i = __builtin_ctzll(c0);
        else i = 64 + __builtin_ctzll(c1);
        const unsigned long long bit = 1ull << (i & 63);
        unsigned long long old = 0ull;
        if (AZG_L == 0) {
            old = __hip_atomic_fetch_or(&C->claimed[i >> 6], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&C->cursor, (uint32_t)i + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (uni_u32((uint32_t)((old & bit) != 0ull))) continue;            // another wave of this workgroup was faster
        // the claim is ours.  Is the snapshot current?  (A wave that handled the tree since the poll recorded the word it consumed BEFORE it
        // released the claim.)
        const uint32_t word = uni_u32(AZG_LDS_LD32(&C->rw[i]));
        if (AZG_L == 0) __hip_atomic_fetch_and(&C->seen[i >> 6], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // taken: ready again only by a later poll
        if (word == 0u || word >= uni_u32(AZG_LDS_LD32(&C->last[i]))) {
            if (AZG_L == 0) __hip_atomic_fetch_and(&C->claimed[i >> 6], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        int t;
        {
            const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
            t = g + i * A->n_sel;
            const uint32_t now = wall32();
            const uint32_t w = now - uni_u32(AZG_LDS_LD32(&C->rts[i]));
            if (AZG_L == 0) {
                atomicAdd(&C->prof[2], (unsigned long long)(now - idle_since));
                atomicAdd(&C->prof[3], (unsigned long long)w);
                atomicAdd(&C->hist[(w / 100u) < 31u ? (w / 100u) : 31u], 1u);
            }
        }
        bool need = false;
        uint32_t left = word - 1u;                                          // calls of this tree still to run in this launch
        if (AZG_LDS_LD32(&C->stop)) left = 0u;                              // the launch's shared budget is spent: the tree retires as it is
#ifdef AZG_ASYNC_EXP_ACQUIRE   /* experiment (profiles/r06_pooling.md): the L1 invalidate a wave would need after claiming a tree that another CU of its XCD may have worked on */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        while (left > 0u) {
            int r;
            {
                const AsyncArgs* a = args;
                asm volatile("" : "+s"(a));                                 // (opaque per call: nothing of a descent is kept live across the loop)
                const AsyncArgsC A = (AsyncArgsC)(uintptr_t)a;
                constexpr uint32_t AGM = AsyncGlobalMask<G>::value;
                const ForestDev F = forest_as_global<AGM>(load_const(&A->F));
                const uint32_t c0t = wall32();
                const uint32_t y0t = (uint32_t)clock64();
                if constexpr (AGM & AG_IO)
                    r = select_tree<G, true>(F, t, sm, dense, as_global(A->aleaf), as_global(A->leaf_valid), as_global(A->needs_eval), A->noise,
                                             as_global(A->pi), as_global(A->v), A->noise);
                else
                    r = select_tree<G, true>(F, t, sm, dense, A->aleaf, A->leaf_valid, A->needs_eval, A->noise, A->pi, A->v, A->noise);
                if (AZG_L == 0) {
                    atomicAdd(&C->prof[0], 1ull); atomicAdd(&C->prof[1], (unsigned long long)(wall32() - c0t));
                    atomicAdd(&C->prof[4], (unsigned long long)((uint32_t)clock64() - y0t));
                }
            }
            left--;
            need = r == 1;
            {
                const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
                const unsigned long long total = A->total_calls;
                if (total) {
                    uint32_t stop = 0u;
                    if (AZG_L == 0) {
                        const uint32_t c = __hip_atomic_fetch_add(&C->calls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
                        if ((c & 31u) == 0u) {
                            const unsigned long long g0 = atomicAdd(&A->ctl->calls, 32ull) + 32ull;
                            if (g0 >= total) { astore(&A->ctl->stop, 1u); __hip_atomic_store(&C->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                        }
                        stop = AZG_LDS_LD32(&C->stop);
                    }
                    if (uni_u32(stop) && r != 1) { left = 0u; }              // (a tree that has just queued a leaf carries on until the net hands it back)
                }
            }
            if (r == 1) break;
            // r == 2: the work budget parked the descent (what a round of the two-kernel form does with such a tree: no leaf this round, the
            // descent goes on in the next one) -- the next call follows at once, on this wave.
            // r == 0: the search is finished, or its fresh root waits for the noise: what azg_selfplay_advance does between rounds happens
            // here and now (the move, the example record, the next search, the clean-up, the noise), then the next call follows
            if (r == 0) {
                drain_vmem();
                const bool on = async_between_calls<G>(args, t, mine);
                if (AZG_L == 0) atomicAdd(&C->prof[5], 1ull);
                if (!on) { left = 0u; break; }                              // idle (episode quota), parked with an error: done with this launch
            }
        }
        // ---- hand the tree on: the leaf's ticket is taken FIRST (a returning atomic: its round trip runs under the drain), then EVERY store
        // of this wave drains (the tree's records, its leaf record), then the word consumed is recorded, the claim released, the ring entry
        // published ----
        uint32_t tk = 0u;
        // ... and the workgroup's ready words are polled HERE as well, under the same drain: a wave that has just finished a tree is the one
        // that needs a new one, and without this it found the LDS copy empty more often than not and paid a poll's round trip (1.5 us of a
        // 25 us cycle) before it could go on
        uint32_t pv0, pv1, ps0, ps1;
        {
            const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
            if (need && AZG_L == 0) { astore(A->evald + t, 0u); tk = atomicAdd(&A->ctl->leaf_tail, 1u); }     // (owed from here on: drained with the leaf record, before the ring entry)
            const uint32_t* my_ts = A->ts_ready + (size_t)g * ASYNC_RS;
            pv0 = AZG_L < n_g ? aload(my_ready + AZG_L) : 0u; pv1 = AZG_L + 64 < n_g ? aload(my_ready + 64 + AZG_L) : 0u;
            ps0 = AZG_L < n_g ? aload(my_ts + AZG_L) : 0u; ps1 = AZG_L + 64 < n_g ? aload(my_ts + 64 + AZG_L) : 0u;
        }
        if (AZG_L == 0) __hip_atomic_store(&C->last[i], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        drain_vmem();
        {
            // publish the snapshot (as a scout would; several waves may do so at once: every value is checked against `last` when it is claimed)
            wave_sync();
            const unsigned long long r0 = __ballot(pv0 != 0u && pv0 < AZG_LDS_LD32(&C->last[AZG_L]));
            const unsigned long long r1 = __ballot(pv1 != 0u && pv1 < AZG_LDS_LD32(&C->last[64 + AZG_L]));
            __hip_atomic_store(&C->rw[AZG_L], pv0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&C->rw[64 + AZG_L], pv1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&C->rts[AZG_L], ps0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&C->rts[64 + AZG_L], ps1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wave_sync();
            if (AZG_L == 0) {
                __hip_atomic_store(&C->seen[0], r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&C->seen[1], r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (AZG_L == 0) __hip_atomic_fetch_and(&C->claimed[i >> 6], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        {
            const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
            if (AZG_L == 0) {
                if (!A->total_calls) A->carry[t] = left;                    // (per-tree budgets: the calls still to run, should this launch end early)
                if (need) {
                    const uint32_t rb = (uint32_t)A->ring_bits;
                    // ring entry: tree [19:0] | time stamp (100 MHz clock >> 4, 12 bits: profile only) [31:20] | calls left [55:32] | lap tag [63:60]
                    __hip_atomic_store(A->ring + (tk & ((1u << rb) - 1u)),
                                       (unsigned long long)((uint32_t)t | (((wall32() >> 4) & 0xFFFu) << 20)) | ((unsigned long long)left << 32) |
                                           ((unsigned long long)(((tk >> rb) & 7u) + 1u) << 60),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    __hip_atomic_fetch_add(&C->retired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    atomicAdd(&A->ctl->retired, 1u);
                }
            }
        }
        idle_since = wall32();
        if (lane_id() == 0) { C->idle_acc[wave] = 0u; C->idle_last[wave] = idle_since; }
    }
#undef AZG_L
    if (lane_id() == 0) atomicAdd(&C->prof[2], (unsigned long long)(wall32() - idle_since));
    __syncthreads();
    {
        const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
        unsigned long long* prof = A->prof;
        const int i = (int)threadIdx.x;
        if (i == 0) {
            atomicAdd(prof + 0, C->prof[0]); atomicAdd(prof + 1, C->prof[1]); atomicAdd(prof + 2, C->prof[2]); atomicAdd(prof + 8, C->prof[3]);
            atomicAdd(prof + 15, C->prof[4]); atomicAdd(prof + 16, C->prof[5]);
            unsigned long long* wi = A->wginfo + (size_t)g * 4;
            wi[0] = where_am_i(); wi[1] = 1ull; wi[2] += C->prof[0]; wi[3] += C->prof[4];   // role | ticks this workgroup stayed << 8
            atomicAdd(prof + 10, (unsigned long long)(wall32() - t_begin));
            if (g == 0 && A->rounds > 0) atomicAdd(prof + 9, 1ull);            // (launches; not the catch-up launch of a per-tree budget)
            // (Launches that ended early are counted by the NET kernel's workgroup 0: a net workgroup leaves only when the launch is over --
            // a descent workgroup leaves when ITS trees are done, with per-tree budgets long before another one's wave may run into the
            // time-out.  Anything more elaborate in the time-out branch itself changes the register allocation of the whole kernel: a
            // post-mortem dump there took the scratch instructions of the Azul / Santorini / Splendor descents from 45 / 61 / 17 to
            // 175 / 194 / 75 and Azul at 1600 simulations from 35 to 25.5 k env-steps/s.)
        }
        if (i < 32 && C->hist[i]) atomicAdd(prof + 64 + i, (unsigned long long)C->hist[i]);
    }
}

#endif  // AZG_ASYNC_PART_SELECT

#ifdef AZG_ASYNC_PART_NET        /* the net kernel and the host side: azg_async.hip */
// The net side of the pipeline is the same for every net: NET names the forward (a 12-wave workgroup body that takes its samples from a batch
// descriptor in LDS), its batch size BS (samples per forward = tickets per range) and its LDS bytes; behind the forward's LDS map sit 512
// bytes: the batch descriptor (tree of sample s [16 ints], calls left [16], count, ...), the profile sums and the samples' valid bit masks.
constexpr int ASYNC_DESC_BYTES = 640;
struct NetV80 {                                // Splendor 2 players: k_v80_net_h2<12>'s body, 16 leaves per forward
    using G = SplendorDev<2>;
    static constexpr int BS = 16, LDS = H2_LDS;
    static __device__ __forceinline__ void run(uint8_t* lds, AsyncArgsC A, const int* sidx, unsigned long long* smask) {
        (void)smask;                           // (h2_net_body finds the masks H2_IND_MASK ints behind sidx)
        h2_net_body<12, true>(lds, &A->W, A->aleaf, (const uint8_t*)A->aleaf, A->F.T, G::P, A->pi, A->v, 0, sidx);
    }
};
constexpr int C5_NET_LDS = C5_LDS_LEAD + 2 * 202 * 128 + 65536 + (2 * 25 * 162 + 25 * 64 + 64 * 2 + 64) * 4;      // (azg_nn.hip conv5_launch)
struct NetC5 {                                 // Santorini no-gods: k_conv5_net<5, 162, 2, 2>'s body, 8 leaves per forward
    using G = SantoriniDev<1>;
    static constexpr int BS = 8, LDS = (C5_NET_LDS + 255) / 256 * 256;
    static __device__ __forceinline__ void run(uint8_t* lds, AsyncArgsC A, const int* sidx, unsigned long long* smask) {
        conv5_net_body<5, 162, 2, 2, true>((float*)lds, &A->C5, A->aleaf, (const uint8_t*)A->aleaf, A->F.T, A->pi, A->v, A->c5_descale, 0, sidx, smask);
    }
};
template <class CF, class GAME>
struct NetMb1d {                               // Splendor 3 / 4 players (8 leaves per forward), Azul (16): k_mb1d_net<CF, true>'s body
    using G = GAME;
    static constexpr int BS = CF::NS, LDS = (CF::LDS_FLOATS * 4 + 255) / 256 * 256;
    static __device__ __forceinline__ void run(uint8_t* lds, AsyncArgsC A, const int* sidx, unsigned long long* smask) {
        mb1d_net_body<CF, true, true, AsyncLeaf<G>::STRIDE, AsyncLeaf<G>::MASK_OFF>((float*)lds, &A->MB, A->aleaf, (const uint8_t*)A->aleaf, A->F.T, A->pi, A->v, 0,
                                                                                    sidx, smask);
    }
};
// The integer hash-net of SURVEY.md Appendix C.3 (tests/hashnet.py; k_eval_hashnet in azg.hip is the same function as a stand-alone kernel)
// as the pipeline's evaluator: the deterministic stand-in for NeuralNet.predict that the parity tests run on both sides.  With it the
// PIPELINE ITSELF -- not only its two-kernel twin -- plays the oracle's episodes and the episodes the reference's Coach.executeEpisode
// played (tests/test_gpu_selfplay.py [async] cases).  A test aid (include/azg_testaids.h), one wave per leaf, for every game that has a
// descent kernel here.
template <class GAME>
struct NetHash {
    using G = GAME;
    static constexpr int BS = 16, LDS = 256;
    static __device__ __forceinline__ void run(uint8_t* lds, AsyncArgsC A, const int* sidx, unsigned long long* smask) {
        (void)lds; (void)smask;
        using AL = AsyncLeaf<G>;
        const int tid = (int)threadIdx.x, wave = tid >> 6, l = tid & 63;
        float* const pi = A->pi;
        float* const v = A->v;
        for (int s = wave; s < BS; s += 12) {
            const int t = sidx[s];
            if (t < 0) continue;                                   // (wave-uniform)
            const uint8_t* rec = (const uint8_t*)A->aleaf + (size_t)t * AL::STRIDE;       // written write-through by a descent wave on another CU
            long long acc = 0;
            for (int i = l; i < G::SP / 4; i += 64) {              // the zero tail beyond S adds nothing
                const uint32_t w = __hip_atomic_load((const uint32_t*)rec + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 4; k++) acc += (long long)(int8_t)(w >> (8 * k)) * (long long)(4 * i + k + 1);
            }
            const uint64_t sum = wave_sum_u64((uint64_t)acc);
            const uint32_t h = (uint32_t)(sum * 2654435761ull);
            unsigned long long m[G::AW];
#pragma unroll
            for (int k = 0; k < G::AW; k++)
                m[k] = __hip_atomic_load((const unsigned long long*)(rec + AL::MASK_OFF) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int wsum = 0;
            for (int a = l; a < G::A; a += 64) {
                unsigned long long mk = m[0];
#pragma unroll
                for (int k = 1; k < G::AW; k++) mk = (a >> 6) == k ? m[k] : mk;
                wsum += ((mk >> (a & 63)) & 1ull) ? 1 + (int)(((h >> 8) + 2654435761u * (uint32_t)a) % 13u) : 0;
            }
            wsum = wave_sum_i32(wsum);
            for (int a = l; a < G::A; a += 64) {
                unsigned long long mk = m[0];
#pragma unroll
                for (int k = 1; k < G::AW; k++) mk = (a >> 6) == k ? m[k] : mk;
                const int w = ((mk >> (a & 63)) & 1ull) ? 1 + (int)(((h >> 8) + 2654435761u * (uint32_t)a) % 13u) : 0;
                __hip_atomic_store((uint32_t*)pi + (size_t)t * G::A + a, __float_as_uint((float)((double)w / (double)wsum)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (l < G::P) {
                const float v0 = (float)((double)h / 2147483648.0 - 1.0);
                __hip_atomic_store((uint32_t*)v + (size_t)t * G::P + l, __float_as_uint(l == 0 ? v0 : (float)(-(double)v0 / (double)(G::P - 1))), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};
using NetSpl3 = NetMb1d<CfgSplendor3, SplendorDev<3>>;
using NetSpl4 = NetMb1d<CfgSplendor4, SplendorDev<4>>;
using NetAzul = NetMb1d<CfgAzul, AzulDev>;
static_assert(NetV80::LDS + ASYNC_DESC_BYTES <= 160 * 1024 && NetC5::LDS + ASYNC_DESC_BYTES <= 160 * 1024 && NetSpl4::LDS + ASYNC_DESC_BYTES <= 160 * 1024 &&
              NetSpl3::LDS + ASYNC_DESC_BYTES <= 160 * 1024 && NetAzul::LDS + ASYNC_DESC_BYTES <= 160 * 1024, "net LDS + batch descriptor");
static_assert(16 * 3 * 8 + H2_IND_MASK * 4 <= ASYNC_DESC_BYTES, "sixteen samples' three mask words behind the descriptor");

template <class NET>
__global__ __launch_bounds__(768) void k_async_net(const AsyncArgs* args) {
    extern __shared__ __attribute__((aligned(256))) uint8_t lds[];
    constexpr int BS = NET::BS;
    constexpr uint32_t FULL = (1u << BS) - 1u;
    int* const sidx = (int*)(lds + NET::LDS);
    const uint32_t t_begin = wall32();
    unsigned long long* const P = (unsigned long long*)(lds + NET::LDS + 160);   // batches, leaves, busy, idle, leaf wait (LDS: nothing live across the forward)
    unsigned long long* const smask = (unsigned long long*)(sidx + H2_IND_MASK);
    if ((int)threadIdx.x < 6) P[threadIdx.x] = 0ull;
    // the workgroup's TICKET RANGE: sixteen consecutive leaf tickets taken with one returning atomic add (a claim that has to look at
    // head and tail and compare-and-swap costs several memory round trips and serialises the 150 workgroups: measured 285 us of leaf
    // wait).  The range is consumed in one batch when its sixteen leaves are there, in several when the producers are slow; sidx[34] =
    // first ticket, sidx[35] = bit mask of the tickets already consumed, sidx[36] = 1 when the range is valid.
    if (threadIdx.x == 0) sidx[36] = 0;
    __syncthreads();
#pragma unroll 1
    for (;;) {
        // (the thread index is opaque per batch: what the claim and the hand-back derive from it -- LDS addresses of the batch descriptor, lane
        // masks -- was otherwise computed at kernel entry, kept across the forward that fills the register file, i.e. SPILLED, and
        // reloaded from scratch with a vmcnt(0) wait in front of the claim's and the hand-back's first use: on the hand-over path)
        int tid_ = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        if (wave == 0) {
            const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
            AsyncCtl* const ctl = A->ctl;
            const int T = A->F.T, wait_ticks = A->batch_wait, timeout = A->timeout_ticks;
            const uint32_t rb = (uint32_t)A->ring_bits, rmask = (1u << rb) - 1u;
            const uint32_t idle0 = wall32();
            uint32_t idle_acc = 0u, idle_last = idle0;                           // (time spent waiting, 1 ms per pass at most: see k_async_select)
            uint32_t base = (uint32_t)sidx[34], taken = (uint32_t)sidx[35];
            if (!sidx[36]) {
                uint32_t b = 0u;
                if (lane == 0) b = atomicAdd(&ctl->leaf_head, (uint32_t)BS);
                base = uni_u32(b); taken = 0u;
            }
            uint32_t tk = base + (uint32_t)(lane % BS);
            uint32_t tag = ((tk >> rb) & 7u) + 1u;
            uint32_t first_seen = 0u;
            unsigned long long e = 0ull;
            bool seen = false;
            int n = 0;
            uint32_t take = 0u;
            unsigned spins = 0u;
            bool dumped = false; (void)dumped;
            for (;;) {
                e = lane < BS ? __hip_atomic_load(A->ring + (tk & rmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const uint32_t filled = (uint32_t)__ballot(lane < BS && (uint32_t)(e >> 60) == tag) & ~taken;
                // a slot of this range that already carries the NEXT lap's tag before its ticket was taken here: the producers lapped this
                // workgroup -- it did not run for a whole lap of the ring (ASYNC_RING_LAPS x T tickets: the platform froze it, see "Recovery") and
                // what was queued into the rest of its range has been overwritten.  It ABANDONS the range and takes a fresh one; the trees whose
                // leaves were lost are still marked as owed their evaluation (`evald`) and are re-queued by the next launch, which this
                // launch's end -- those trees never come back, a wave runs into the time-out -- brings about.
                if ((uint32_t)__ballot(lane < BS && (uint32_t)(e >> 60) == ((((tk >> rb) + 1u) & 7u) + 1u)) & ~taken) {
                    uint32_t b = 0u;
                    if (lane == 0) { b = atomicAdd(&ctl->leaf_head, (uint32_t)BS); atomicAdd(A->prof + 26, 1ull); }
                    base = uni_u32(b); taken = 0u;
                    tk = base + (uint32_t)(lane % BS); tag = ((tk >> rb) & 7u) + 1u;
                    seen = false;
                    continue;
                }
                const uint32_t now = wall32();
                { const uint32_t d = now - idle_last; idle_acc += d < ASYNC_IDLE_STEP_CAP ? d : ASYNC_IDLE_STEP_CAP; idle_last = now; }
#ifdef AZG_ASYNC_POSTMORTEM   /* debug: this range has shown nothing for 1 ms although every ticket of it has been issued -- what do the slots hold? */
                if (!filled && !dumped && (now - idle0) > 100000u) {
                    const uint32_t tl = uni_u32(__hip_atomic_fetch_add(&ctl->leaf_tail, uni_u32(0u) * (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if ((int32_t)(tl - (base + (uint32_t)BS)) > 0) {
                        dumped = true;
                        unsigned long long* dd = A->wginfo + (size_t)4 * 1024 + 280 + 512 + 40 * (size_t)(blockIdx.x < 104 ? blockIdx.x : 103);
                        if (lane < BS) { dd[8 + lane] = e; dd[24 + lane] = __hip_atomic_fetch_add(A->ring + (tk & rmask), (unsigned long long)(uni_u32(0u) * (uint32_t)lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        if (lane == 0) { dd[0] = (unsigned long long)base | ((unsigned long long)taken << 32); dd[1] = (unsigned long long)tl | ((unsigned long long)aload(&ctl->leaf_head) << 32);
                                         dd[2] = (unsigned long long)(now - t_begin); dd[3] = (unsigned long long)tag | ((unsigned long long)rb << 8) | ((unsigned long long)P[0] << 16); }
                    }
                }
#endif
                if (filled && !seen) { seen = true; first_seen = now; }
                if (filled && ((filled | taken) == FULL || (int)(now - first_seen) >= wait_ticks)) { take = filled; n = __popc(filled); break; }
                if (!filled && (++spins & 7u) == 0u) {
                    if (uni_u32(aload(&ctl->retired)) >= (uint32_t)T || uni_u32(aload(&ctl->abort))) { n = -1; break; }
                    if (idle_acc > (uint32_t)timeout) {                              // (no leaf for that long)
                        if (lane == 0) {
                            astore(&ctl->abort, 2u);
                            if (atomicAdd(A->prof + 19, 1ull) >= 7ull) atomicOr(&A->F.hdr[0].err, ERR_ASYNC_TIMEOUT);
                        }
                        n = -1;
                        break;
                    }
                }
                if (filled) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4);
            }
            if (n > 0) {
                const uint32_t now = wall32();
                taken |= take;
                if (lane < 16) { sidx[lane] = -1; sidx[16 + lane] = 0; }
                const bool mine = lane < BS && ((take >> lane) & 1u);
                uint32_t w = 0u;
                if (mine) {
                    const int slot = __popc(take & ((1u << lane) - 1u));           // the batch is the taken tickets, compacted
                    const int t = (int)((uint32_t)e & 0xFFFFFu);
                    sidx[slot] = t;
                    sidx[16 + slot] = (int)((e >> 32) & 0xFFFFFFu);
                    w = ((((now >> 4) & 0xFFFu) - (((uint32_t)e >> 20) & 0xFFFu)) & 0xFFFu) << 4;       // (ticks; wraps at 655 us)
                    atomicAdd(A->prof + 32 + ((w / 100u) < 31u ? (w / 100u) : 31u), 1ull);
                }
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) w += __shfl_xor(w, m, 16);
                if (lane == 0) {
                    P[4] += (unsigned long long)w; P[3] += (unsigned long long)(now - idle0);
                    sidx[34] = (int)base; sidx[35] = (int)taken; sidx[36] = taken != FULL ? 1 : 0;
                }
            }
            if (lane == 0) sidx[32] = n;
        }
        __syncthreads();
        if (sidx[32] < 0) break;
        if (tid == 0) { sidx[33] = (int)wall32(); sidx[37] = (int)(uint32_t)clock64(); }
        {
            const AsyncArgs* a = args;
            asm volatile("" : "+s"(a));
            const AsyncArgsC A = (AsyncArgsC)(uintptr_t)a;
            NET::run(lds, A, sidx, smask);
        }
        // the next ticket range, when this one is used up: the returning atomic is issued HERE, so that its round trip runs under the drain
        // and the hand-back below instead of in front of the next claim (measured: 4.6 us per batch outside the forward, 12 % of a
        // net-bound pipeline's time)
        uint32_t next_base = 0u;
        const bool need_range = wave == 0 && !sidx[36];
        if (need_range && lane == 0) next_base = atomicAdd(&((AsyncArgsC)(uintptr_t)args)->ctl->leaf_head, (uint32_t)BS);
#ifdef AZG_ASYNC_POSTMORTEM   /* debug: is the range this workgroup was just given plausible?  (head can be at most ~T + ranges behind / ahead of the tail) */
        if (need_range && lane == 0) {
            const AsyncArgsC A_ = (AsyncArgsC)(uintptr_t)args;
            const uint32_t tl = __hip_atomic_fetch_add(&A_->ctl->leaf_tail, (uint32_t)(clock64() >> 62), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int32_t)(tl - next_base) > 16384 || (int32_t)(next_base - tl) > 16384) {
                unsigned long long* dd = A_->wginfo + (size_t)4 * 1024 + 280 + 512 + 40 * (size_t)(blockIdx.x < 104 ? blockIdx.x : 103);
                if (dd[4] == 0ull) { dd[4] = (unsigned long long)next_base | ((unsigned long long)tl << 32); dd[5] = (unsigned long long)(uint32_t)sidx[34] | ((unsigned long long)(wall32() - t_begin) << 32);
                                     dd[6] = (unsigned long long)__hip_atomic_fetch_add(&A_->ctl->leaf_head, (uint32_t)(clock64() >> 62), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | ((unsigned long long)P[0] << 32); }
            }
        }
#endif
        drain_vmem();                                   // EVERY wave: its write-through pi / v rows have left
        __syncthreads();
        {
            const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
            if (need_range && lane == 0) { sidx[34] = (int)next_base; sidx[35] = 0; sidx[36] = 1; }
            if (tid < 16 && sidx[tid] >= 0) {
                const int t = sidx[tid], ns = A->n_sel;
                const int gi = t % ns, ii = t / ns;
                astore(A->evald + t, 1u);                                       // (read by the NEXT launch only: see "Recovery")
                astore(A->ts_ready + (size_t)gi * ASYNC_RS + ii, wall32());     // (profile only: not ordered against the ready word)
                astore(A->ready + (size_t)gi * ASYNC_RS + ii, (uint32_t)sidx[16 + tid] + 1u);
            }
        }
        if (tid == 0) {
            P[0] += 1ull; P[1] += (unsigned long long)sidx[32]; P[2] += (unsigned long long)(wall32() - (uint32_t)sidx[33]);
            P[5] += (unsigned long long)((uint32_t)clock64() - (uint32_t)sidx[37]);
        }
        // (no barrier here: between the barrier above and the next claim only wave 0 touches the batch descriptor, in program order)
    }
    if (threadIdx.x == 0) {
        const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
        unsigned long long* prof = A->prof;
        atomicAdd(prof + 3, P[0]); atomicAdd(prof + 4, P[1]); atomicAdd(prof + 5, P[2]); atomicAdd(prof + 6, P[3]);
        atomicAdd(prof + 7, P[4]); atomicAdd(prof + 11, (unsigned long long)(wall32() - t_begin)); atomicAdd(prof + 14, P[5]);
        unsigned long long* wi = A->wginfo + (size_t)(A->n_sel + (int)blockIdx.x) * 4;
        wi[0] = where_am_i(); wi[1] = 2ull | ((unsigned long long)(wall32() - t_begin) << 8); wi[2] += P[0]; wi[3] += P[5];
        // sticky: launches that ended because a descent wave (17) / a net workgroup (18) gave up; 19 = such launches in a row (eight: an error)
        if (blockIdx.x == 0) { const uint32_t ab = aload(&A->ctl->abort); if (ab == 1u || ab == 2u) atomicAdd(prof + 16 + ab, 1ull); else if (ab == 0u) prof[19] = 0ull; }
        // (post-mortems: the ticket range this workgroup held when it left -- first ticket | tickets taken << 32, valid | last batch size << 32)
        unsigned long long* dn = A->wginfo + (size_t)4 * 1024 + 280 + 2 * (size_t)blockIdx.x;
        dn[0] = (unsigned long long)(uint32_t)sidx[34] | ((unsigned long long)(uint32_t)sidx[35] << 32);
        dn[1] = (unsigned long long)(uint32_t)sidx[36] | ((unsigned long long)(uint32_t)sidx[32] << 32);
    }
}

#endif  // AZG_ASYNC_PART_NET
}  // namespace azg

#ifdef AZG_ASYNC_PART_SELECT
// the descent kernel's launcher, called by azg_forest_async_rounds_v80_h2 (azg_async.hip): the two kernels live in two translation units
// because they want different code generation (build.py)
template <class G>
static int async_launch_select(const azg::AsyncArgs* devbuf, int n_sel, hipStream_t s) {
    static bool attr[64];                       // (per device: a function attribute belongs to the device that was current when it was set)
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    if (device >= 0 && device < 64 && !attr[device]) {
        HIPCHK(hipFuncSetAttribute((const void*)azg::k_async_select<G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr[device] = true;
    }
    azg::k_async_select<G><<<dim3(n_sel), dim3(azg::ASYNC_SEL_WAVES * 64), azg::ASYNC_SEL_WAVES * azg::RoundLds<G>::STRIDE + (int)sizeof(azg::AsyncSelLds), s>>>(devbuf);
    HIPCHK(hipGetLastError());
    return 0;
}
// net_kind: 0 = Splendor 2 players (V80), 1 = Santorini no-gods (V89), 2 / 3 = Splendor 3 / 4 players, 4 = Azul (MobileNet-1d), 5 = Santorini with gods
int azg_async_launch_select(int net_kind, const azg::AsyncArgs* devbuf, int n_sel, hipStream_t s) {
#ifdef AZG_ASYNC_ONLY_KIND      /* code-generation experiments: one game's kernel only */
    if (net_kind != AZG_ASYNC_ONLY_KIND) return -1;
    return async_launch_select<AZG_ASYNC_ONLY_GAME>(devbuf, n_sel, s);
#else
    switch (net_kind) {
        case 0: return async_launch_select<azg::SplendorDev<2>>(devbuf, n_sel, s);
        case 1: return async_launch_select<azg::SantoriniDev<1>>(devbuf, n_sel, s);
        case 2: return async_launch_select<azg::SplendorDev<3>>(devbuf, n_sel, s);
        case 3: return async_launch_select<azg::SplendorDev<4>>(devbuf, n_sel, s);
        case 4: return async_launch_select<azg::AzulDev>(devbuf, n_sel, s);
#ifdef AZG_ASYNC_SANTORINI11
        case 5: return async_launch_select<azg::SantoriniDev<11>>(devbuf, n_sel, s);
#endif
        default: return -1;
    }
#endif
}
#endif  // AZG_ASYNC_PART_SELECT

#ifdef AZG_ASYNC_PART_NET
int azg_async_launch_select(int net_kind, const azg::AsyncArgs* devbuf, int n_sel, hipStream_t s);
// ---- host side ----
struct AsyncSlot {
    AsyncArgs host; AsyncArgs* devbuf;
    int8_t* aleaf; AsyncCtl* ctl; unsigned long long* ring; uint32_t* ready; uint32_t* ts; uint32_t* evald; uint8_t* pending; uint32_t* carry; unsigned long long* prof; unsigned long long* wginfo;
    hipEvent_t fork, join, join_net;
    int n_sel, n_net, ring_bits;
    int device, n_cu, leaf_stride, T;          // what the buffers were sized for (checked on every reuse)
};
static void async_slot_free(void* p) {
    AsyncSlot* s = (AsyncSlot*)p;
    (void)hipFree(s->devbuf); (void)hipFree(s->aleaf); (void)hipFree(s->ctl); (void)hipFree(s->ring); (void)hipFree(s->ready);
    (void)hipFree(s->ts); (void)hipFree(s->evald); (void)hipFree(s->pending); (void)hipFree(s->carry); (void)hipFree(s->prof); (void)hipFree(s->wginfo);
    if (s->fork) (void)hipEventDestroy(s->fork);
    if (s->join) (void)hipEventDestroy(s->join);
    if (s->join_net) (void)hipEventDestroy(s->join_net);
    delete s;
}

// Per-DEVICE state of the pipeline: the CU count and the two private streams its kernels run on.
// The two kernels MUST run side by side, so their streams must not share a hardware queue (HIP multiplexes streams onto a few queues
// -- 4 per priority level by default -- and kernels of one queue run one after the other: the net kernel would wait for leaves that
// the descent kernel, queued behind it, can never deliver; seen as the 2 s time-out in a process that had created many streams
// before) and must not synchronise implicitly with the legacy default stream (a BLOCKING stream's kernel waits for the default
// stream's earlier work: the same dead end when the caller is on the default stream).  Both kernels therefore run on two private
// NON-BLOCKING streams of the HIGH priority level, created once per device on the first launch there -- the first streams of that
// level own a hardware queue each --, and the caller's stream only forks into them and joins them.
// (Round 5 kept these as function-local statics made on whichever device was current at the first call: a process that moved to
// another GPU afterwards launched on the first device's streams with the first device's CU count.)
// The pipeline assumes that it has the GPU's CUs to itself while a launch runs: every workgroup must be resident (n_sel + n_net <= CUs)
// because they wait for each other.  Another process or stream that occupies CUs for seconds makes the launch end in the time-out
// (error bit 128) -- loud, never a hang; see include/azg.h.
struct AsyncDevice { int n_cu; hipStream_t net_stream, sel_stream; bool attr_done; };
constexpr int ASYNC_MAX_DEVICES = 64;
static AsyncDevice g_async_dev[ASYNC_MAX_DEVICES];

// include/azg.h: profile counters of the asynchronous pipeline since the last reset
extern "C" int azg_forest_async_profile(azg_forest* f, double* out /* [ASYNC_NPROF] */, int reset) {
    if (!f || !out) return fail("azg_forest_async_profile: null argument");
    for (int i = 0; i < ASYNC_NPROF; i++) out[i] = 0.0;
    AsyncSlot* sl = (AsyncSlot*)azg_forest_attached(f, "async_v80");
    if (!sl) return 0;
    HIPCHK(hipDeviceSynchronize());
    unsigned long long h[ASYNC_NPROF];
    HIPCHK(hipMemcpy(h, sl->prof, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < ASYNC_NPROF; i++) out[i] = (double)h[i];
    out[12] = (double)sl->n_sel; out[13] = (double)sl->n_net;
    {   // the control block as the last launch left it: tickets issued / claimed, trees retired, abort code, shared-budget state
        AsyncCtl c;
        HIPCHK(hipMemcpy(&c, sl->ctl, sizeof(c), hipMemcpyDeviceToHost));
        out[20] = (double)c.leaf_tail; out[21] = (double)c.leaf_head; out[22] = (double)c.retired; out[23] = (double)c.abort;
        out[24] = (double)c.calls; out[25] = (double)c.stop;
    }
    if (reset) HIPCHK(hipMemset(sl->prof, 0, sizeof(h)));
    return 0;
}

// include/azg.h: per-workgroup placement and load of the pipeline's kernels (debugging / placement studies)
extern "C" int azg_forest_async_wginfo(azg_forest* f, unsigned long long* out /* host [max_wg][4] */, int max_wg, int reset) {
    if (!f || !out) return fail("azg_forest_async_wginfo: null argument");
    AsyncSlot* sl = (AsyncSlot*)azg_forest_attached(f, "async_v80");
    if (!sl) return 0;
    const int n = sl->n_sel + sl->n_net < max_wg ? sl->n_sel + sl->n_net : max_wg;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, sl->wginfo, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(sl->wginfo, 0, sizeof(unsigned long long) * 4 * n));
    return n;
}

// include/azg_testaids.h: post-mortem of a pipeline time-out: the debug area behind the workgroup table, the ready words and the leaf ring as they
// are in MEMORY after the launch
extern "C" int azg_forest_async_debug(azg_forest* f, unsigned long long* out /* host [792 + 104 * 40] */, uint32_t* ready_out /* host [128 * max_wg] or null */, int max_wg,
                                      unsigned long long* ring_out /* host [ring slots] or null */, int max_ring) {
    if (!f || !out) return fail("azg_forest_async_debug: null argument");
    AsyncSlot* sl = (AsyncSlot*)azg_forest_attached(f, "async_v80");
    if (!sl) return 0;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, sl->wginfo + (size_t)4 * 1024, sizeof(unsigned long long) * (280 + 512 + 104 * 40), hipMemcpyDeviceToHost));
    if (ready_out) HIPCHK(hipMemcpy(ready_out, sl->ready, sizeof(uint32_t) * ASYNC_RS * (size_t)(max_wg < sl->n_cu ? max_wg : sl->n_cu), hipMemcpyDeviceToHost));
    if (ring_out) HIPCHK(hipMemcpy(ring_out, sl->ring, sizeof(unsigned long long) * (size_t)(max_ring < (1 << sl->ring_bits) ? max_ring : (1 << sl->ring_bits)), hipMemcpyDeviceToHost));
    return sl->ring_bits;
}

// One launch of the pipeline.  kind = which game's descent kernel (and, hash == 0, which net): 0 Splendor 2 players + V80 (w = 43 pointers,
// descale = 16 host floats); 1 Santorini no-gods + V89 (w = 14 pointers, descale = 1 host float); 2 / 3 Splendor 3 / 4 players, 4 Azul
// (MobileNet-1d: 43 pointers + 16 factors); 5 Santorini with gods (hash-net only so far).  hash != 0: the integer hash-net as the evaluator.
// Recovery, first half: in front of the two persistent kernels, every tree that is still owed the evaluation of the leaf it queued in an
// earlier launch (status ST_WAIT_NN, `evald` clear: that launch ended early) gets its leaf record -- still in the pipeline's leaf array --
// back on the ring and is flagged `pending`: its descent workgroup starts it as "in the net".  Costs one tiny launch; in the common case
// (the previous launch ended normally) it finds nothing.
__global__ __launch_bounds__(256) void k_async_requeue(const AsyncArgs* args) {
    const AsyncArgsC A = (AsyncArgsC)(uintptr_t)args;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= A->F.T) return;
    const bool owed = aload(A->evald + t) == 0u && aload(&A->F.hdr[t].status) == ST_WAIT_NN && !aload(&A->F.hdr[t].err);
    A->pending[t] = owed ? (uint8_t)1 : (uint8_t)0;
    if (owed) {
        const uint32_t rb = (uint32_t)A->ring_bits, tk = atomicAdd(&A->ctl->leaf_tail, 1u);
        const uint32_t calls = (uint32_t)A->rounds + (A->total_calls ? 0u : A->carry[t]);   // (per-tree budgets: + what the tree was left owing)
        if (!A->total_calls) A->carry[t] = calls;
        // (calls left = the launch's `rounds`: the hand-back word rounds + 1 is below the 0xFFFFFFFF the workgroup starts its bookkeeping from)
        A->ring[tk & ((1u << rb) - 1u)] = (unsigned long long)((uint32_t)t | (((wall32() >> 4) & 0xFFFu) << 20)) | ((unsigned long long)calls << 32) |
                                           ((unsigned long long)(((tk >> rb) & 7u) + 1u) << 60);
        atomicAdd(A->prof + 27, 1ull);
    }
}

template <class NET>
static int async_launch_net(const AsyncArgs* devbuf, int n_net, hipStream_t s) {
    k_async_net<NET><<<dim3(n_net), dim3(768), NET::LDS + ASYNC_DESC_BYTES, s>>>(devbuf);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class NET>
static int async_net_attr() {
    HIPCHK(hipFuncSetAttribute((const void*)k_async_net<NET>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}
static int async_rounds_impl(const char* who, int kind, int hash, azg_forest* f, uint8_t* leaf_valid, uint8_t* needs_eval, float* pi, float* v,
                             int noise_stride, const void* const* w, const float* descale, int rounds, int n_net, int n_sel, int batch_wait_ticks,
                             int shared_budget, void* stream) {
    const std::string me(who);
    if (!f || !leaf_valid || !needs_eval || !pi || !v || (!hash && (!w || !descale))) return fail(me + ": null argument");
    if (rounds < 0 || (rounds == 0 && shared_budget)) return 0;       // (per-tree budgets: rounds == 0 = only what earlier launches left over)
    if (rounds >= (1 << 24)) return fail(me + ": at most 2^24 - 1 rounds per launch");
    if (noise_stride != 0 && noise_stride != -2) return fail(me + ": noise_stride must be 0 or -2");
    int game = 0, variant = 0;
    double alpha = 0.0;
    const ForestDev* dev = azg_forest_dev_internal(f, &game, &variant, &alpha);
    if (kind == 0 && (game != AZG_SPLENDOR || variant != 2)) return fail(me + ": Splendor 2 players only (the V80 geometry of nn_v80_h2.hip.h)");
    if (kind == 1 && (game != AZG_SANTORINI || variant != 1)) return fail(me + ": Santorini without gods only (the V89 geometry of nn_conv5x5.hip.h)");
    if ((kind == 2 || kind == 3) && (game != AZG_SPLENDOR || variant != kind + 1)) return fail(me + ": the geometry does not match the forest's game");
    if (kind == 4 && game != AZG_AZUL) return fail(me + ": the geometry does not match the forest's game");
    if (kind == 5 && (game != AZG_SANTORINI || variant != 11)) return fail(me + ": Santorini with gods only");
    static_assert(AsyncLeaf<SplendorDev<2>>::STRIDE == H2_AL_STRIDE && AsyncLeaf<SplendorDev<2>>::MASK_OFF == H2_AL_MASK, "leaf record layout shared with the net kernel");
    static_assert(AsyncLeaf<SantoriniDev<1>>::STRIDE == C5_AL_STRIDE && AsyncLeaf<SantoriniDev<1>>::MASK_OFF == C5_AL_MASK, "leaf record layout shared with the net kernel");
    const int leaf_strides[6] = {H2_AL_STRIDE, C5_AL_STRIDE, AsyncLeaf<SplendorDev<3>>::STRIDE, AsyncLeaf<SplendorDev<4>>::STRIDE, AsyncLeaf<AzulDev>::STRIDE,
                                 AsyncLeaf<SantoriniDev<11>>::STRIDE};
    const int batch[6] = {NetV80::BS, NetC5::BS, NetSpl3::BS, NetSpl4::BS, NetAzul::BS, 16};
    const int leaf_stride = leaf_strides[kind], bs = hash ? 16 : batch[kind];
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    if (device < 0 || device >= ASYNC_MAX_DEVICES) return fail(me + ": device index out of range");
    AsyncDevice& D = g_async_dev[device];
    if (!D.n_cu) {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        hipStream_t a = nullptr, b = nullptr;
        HIPCHK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        if (async_net_attr<NetV80>() || async_net_attr<NetC5>() || async_net_attr<NetSpl3>() || async_net_attr<NetSpl4>() || async_net_attr<NetAzul>()) return -1;
        D.net_stream = a; D.sel_stream = b; D.n_cu = prop.multiProcessorCount;
    }
    const int n_cu = D.n_cu;
    const int T = dev->T;
    if (n_net <= 0 || n_sel <= 0) {
        // default split of the CUs.  V80: measured at 4096 x 800 (round 6, descent 19.9 us: see DESIGN.md 3.6); V89: round 5 a forward of 8
        // leaves cost ~75 us of a CU, a descent ~33 us of a sixteenth of one: 13 / 16 for the net (208 + 48 -> 28.8 k env-steps/s, 216 + 40
        // 24.5 k, 204 + 52 28.7 k, 200 + 56 28.3 k; two kernels 26.5 k); round 6 with the forward at 63-70 us (nn_tid): 3 / 4 (208 + 48 -> 26.8 k,
        // 200 + 56 30.5 k, 192 + 64 32.4 k, 184 + 72 31.7 k, 176 + 80 30.7 k; with the claim path off scratch 192 + 64 33.2 k, 196 + 60 33.9 k,
        // 200 + 56 33.6 k: 49 / 64); Splendor 3 / 4 players (forward 55-59 us per 8 leaves, descent 27 us):
        // round 5 25 / 32 for the net (4 players: 200 + 56 -> 35.8 k, 208 + 48 33.1 k, 192 + 64 34.6 k), round 6 with the descent at 22 us
        // 13 / 16 (200 + 56 -> 38.8 k, 208 + 48 40.1 k, 216 + 40 36.1 k); Azul (descent-heavy, forward 29 us per 16):
        // round 5 3 / 8 (96 + 160 -> 70.4 k; 112 + 144 68.6 k, 88 + 168 66.3 k), round 6 13 / 32 (96 + 160 69.9 k, 104 + 152 72.3 k).  The hash-net costs next to nothing: a sixteenth.
        n_net = hash ? (n_cu / 16 > 0 ? n_cu / 16 : 1) : kind == 0 ? n_cu * AZG_V80_NET_SHARE_256 / 256 : kind == 4 ? n_cu * 13 / 32 : (kind == 2 || kind == 3) ? n_cu * 13 / 16 : n_cu * 49 / 64;
        n_sel = n_cu - n_net;
    }
    if (n_sel > T) n_sel = T;
    if (n_net > (T + bs - 1) / bs) n_net = (T + bs - 1) / bs;
    if (n_net + n_sel > n_cu)
        return fail(me + ": n_net + n_sel exceeds the CUs of the device (every workgroup of the pipeline must be resident)");
    if ((long long)n_sel * ASYNC_RS < T) return fail(me + ": more than 128 trees per select workgroup");
    if (T >= (1 << 20)) return fail(me + ": at most 2^20 - 1 trees");
    AsyncSlot* sl = (AsyncSlot*)azg_forest_attached(f, "async_v80");
    if (sl && (sl->device != device || sl->n_cu != n_cu || sl->leaf_stride != leaf_stride || sl->T != T))
        return fail(me + ": the forest's pipeline buffers were made for another device / leaf layout (one game and one GPU per forest)");
    if (!sl) {
        // built in a local object and attached to the forest only when every allocation has succeeded: a half-made slot would be found by
        // the next call and launched with null queues
        AsyncSlot* n = new AsyncSlot();
        memset(n, 0, sizeof(*n));
        memset(&n->host, 0xFF, sizeof(n->host));
        int rb = 6;
        while ((1 << rb) < ASYNC_RING_LAPS * T) rb++;
        n->ring_bits = rb; n->device = device; n->n_cu = n_cu; n->leaf_stride = leaf_stride; n->T = T;
        const bool ok =
            hipMalloc(&n->devbuf, sizeof(AsyncArgs)) == hipSuccess && hipMalloc(&n->aleaf, (size_t)T * leaf_stride) == hipSuccess &&
            hipMalloc(&n->ctl, sizeof(AsyncCtl)) == hipSuccess && hipMalloc(&n->ring, sizeof(unsigned long long) << rb) == hipSuccess &&
            hipMalloc(&n->ready, sizeof(uint32_t) * ASYNC_RS * n_cu) == hipSuccess &&           // (sized for any split: it may change from launch to launch)
            hipMalloc(&n->ts, sizeof(uint32_t) * ASYNC_RS * n_cu) == hipSuccess && hipMalloc(&n->evald, sizeof(uint32_t) * (size_t)T) == hipSuccess &&
            hipMemset(n->evald, 1, sizeof(uint32_t) * (size_t)T) == hipSuccess && hipMalloc(&n->pending, (size_t)T) == hipSuccess && hipMemset(n->pending, 0, (size_t)T) == hipSuccess &&
            hipMalloc(&n->carry, sizeof(uint32_t) * (size_t)T) == hipSuccess && hipMemset(n->carry, 0, sizeof(uint32_t) * (size_t)T) == hipSuccess &&       // (non-zero: whatever a tree is waiting for when the pipeline first sees it has been evaluated)
            hipMalloc(&n->prof, sizeof(unsigned long long) * ASYNC_NPROF) == hipSuccess &&
            hipMalloc(&n->wginfo, sizeof(unsigned long long) * (4 * 1024 + 280 + 512 + 104 * 40)) == hipSuccess &&
            hipMemset(n->prof, 0, sizeof(unsigned long long) * ASYNC_NPROF) == hipSuccess &&
            hipMemset(n->wginfo, 0, sizeof(unsigned long long) * (4 * 1024 + 280 + 512 + 104 * 40)) == hipSuccess && hipMemset(n->aleaf, 0, (size_t)T * leaf_stride) == hipSuccess &&
            hipMemset(n->ts, 0, sizeof(uint32_t) * ASYNC_RS * n_cu) == hipSuccess &&
            hipEventCreateWithFlags(&n->fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&n->join, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&n->join_net, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            const hipError_t e = hipGetLastError();
            async_slot_free(n);
            return fail(me + ": could not allocate the pipeline's queues (" + hipGetErrorString(e) + ")");
        }
        sl = n;
        azg_forest_attach(f, "async_v80", sl, async_slot_free);
    }
    sl->n_sel = n_sel; sl->n_net = n_net;
    AsyncArgs want;
    memset(&want, 0, sizeof(want));
    want.F = *dev;
    // (the forest's work / level budget stays: a call that runs into it is followed by the next one at once, so no tree waits for another
    // inside a launch -- but a launch ends only when every tree has had its calls, and a tree whose simulations all end on terminal nodes
    // would otherwise run its whole search inside ONE call: measured 4.1 ms launches of 48 rounds where the mean tree needs 3.4 ms)
    if (hash) {
    } else if (kind == 0) want.W = h2_weights(w, descale);
    else if (kind >= 2) {                             // the 43-pointer table + 16 descale factors of azg_nn_mb1d_forward_h2
        const float* const* wf = (const float* const*)w;
        Mb1dNetW& N = want.MB;
        for (int i = 0; i < 16; i++) N.ds[i] = descale[i];
        N.W0 = wf[0]; N.b0 = wf[1];
        for (int b = 0; b < 3; b++) {
            const float* const* q = wf + 2 + 11 * b;
            N.blk[b] = Mb1dBlockW{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10]};
        }
        const float* const* h = wf + 35;
        N.Wpi1 = h[0]; N.bpi1 = h[1]; N.Wpi2 = h[2]; N.bpi2 = h[3]; N.Wv1 = h[4]; N.bv1 = h[5]; N.Wv2 = h[6]; N.bv2 = h[7];
    } else {
        const float* const* wf = (const float* const*)w;
        want.C5 = Conv5NetW{wf[0], wf[1], wf[2], wf[3], wf[4], wf[5], wf[6], wf[7], wf[8], wf[9], wf[10], wf[11], wf[12], wf[13]};
        want.c5_descale = descale[0];
    }
    want.aleaf = sl->aleaf; want.leaf_valid = leaf_valid; want.needs_eval = needs_eval; want.pi = pi; want.v = v;
    want.ctl = sl->ctl; want.ring = sl->ring; want.ready = sl->ready; want.ts_ready = sl->ts; want.evald = sl->evald; want.pending = sl->pending; want.carry = sl->carry; want.prof = sl->prof; want.wginfo = sl->wginfo;
    want.noise = (alpha != 0.0 && noise_stride == -2) ? 1 : 0;
    want.n_sel = n_sel; want.ring_bits = sl->ring_bits;
    hipStream_t s = (hipStream_t)stream;
    if (shared_budget) HIPCHK(hipMemsetAsync(sl->carry, 0, sizeof(uint32_t) * (size_t)T, s));      // (per-tree balances mean nothing to a shared budget)
    else if (rounds >= (1 << 22)) return fail(me + ": at most 2^22 - 1 rounds per launch with per-tree budgets");
    // One launch of the two kernels.  Per-tree budgets: TWO -- the second grants no calls, it only runs what the first left over if it ended early
    // (the platform froze a workgroup, "Recovery"): after it every tree has had its `rounds` calls again, without the host looking at the
    // outcome of the first.  (Nothing left over: every tree is claimed once and retires -- ~30 us.)
    auto launch_once = [&](int rounds) -> int {
    want.rounds = rounds;
    if (shared_budget) {                              // `rounds` x T calls for the trees together; no tree is held back by a share of its own
        want.total_calls = (unsigned long long)rounds * (unsigned long long)T;
        want.rounds = (1 << 24) - 2;
    }
    want.batch_wait = batch_wait_ticks >= 0 ? batch_wait_ticks : 150;
    // (a launch that stalls ends early and the next one carries on: 50 ms of a wave's own idling)
    { const char* e = getenv("AZG_ASYNC_TIMEOUT_MS"); int ms = e ? atoi(e) : 50; ms = ms < 1 ? 1 : (ms > 20000 ? 20000 : ms); want.timeout_ticks = ms * 100000; }   // (32-bit ticks of 10 ns: <= 20 s)
    // test hook (tests/test_gpu_selfplay.py): AZG_ASYNC_TEST_STALL=k cuts the k-th launch of the process short -- its time-out is 20 us, so the first
    // wave that idles ends it -- to exercise the recovery path without waiting for the platform to freeze a workgroup
    {
        static int n_launch = 0;
        const char* e = getenv("AZG_ASYNC_TEST_STALL");
        if (e) { if (++n_launch == atoi(e)) want.timeout_ticks = 2000; } else n_launch = 0;
    }
    if (memcmp(&sl->host, &want, sizeof(want)) != 0) {
        sl->host = want;
        HIPCHK(hipMemcpyAsync(sl->devbuf, &sl->host, sizeof(AsyncArgs), hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemsetAsync(sl->ctl, 0, sizeof(AsyncCtl), s));
    HIPCHK(hipMemsetAsync(sl->ring, 0, sizeof(unsigned long long) << sl->ring_bits, s));
    k_async_requeue<<<dim3((T + 255) / 256), dim3(256), 0, s>>>(sl->devbuf);       // (see "Recovery"; in stream order behind the resets above)
    HIPCHK(hipGetLastError());
    // fork from the caller's stream into the two private streams, join both before returning
    HIPCHK(hipEventRecord(sl->fork, s));
    HIPCHK(hipStreamWaitEvent(D.net_stream, sl->fork, 0));
    HIPCHK(hipStreamWaitEvent(D.sel_stream, sl->fork, 0));
    int rc;
    if (hash) {
        switch (kind) {
            case 0: rc = async_launch_net<NetHash<SplendorDev<2>>>(sl->devbuf, n_net, D.net_stream); break;
            case 1: rc = async_launch_net<NetHash<SantoriniDev<1>>>(sl->devbuf, n_net, D.net_stream); break;
            case 2: rc = async_launch_net<NetHash<SplendorDev<3>>>(sl->devbuf, n_net, D.net_stream); break;
            case 3: rc = async_launch_net<NetHash<SplendorDev<4>>>(sl->devbuf, n_net, D.net_stream); break;
            case 4: rc = async_launch_net<NetHash<AzulDev>>(sl->devbuf, n_net, D.net_stream); break;
#ifdef AZG_ASYNC_SANTORINI11
            case 5: rc = async_launch_net<NetHash<SantoriniDev<11>>>(sl->devbuf, n_net, D.net_stream); break;
#endif
            default: return fail(me + ": no descent kernel for this game in the pipeline");
        }
    } else {
        switch (kind) {
            case 0: rc = async_launch_net<NetV80>(sl->devbuf, n_net, D.net_stream); break;
            case 1: rc = async_launch_net<NetC5>(sl->devbuf, n_net, D.net_stream); break;
            case 2: rc = async_launch_net<NetSpl3>(sl->devbuf, n_net, D.net_stream); break;
            case 3: rc = async_launch_net<NetSpl4>(sl->devbuf, n_net, D.net_stream); break;
            case 4: rc = async_launch_net<NetAzul>(sl->devbuf, n_net, D.net_stream); break;
            default: return fail(me + ": no engine net for this game in the pipeline");
        }
    }
    if (rc) return -1;
    if (azg_async_launch_select(kind, sl->devbuf, n_sel, D.sel_stream)) return -1;
    HIPCHK(hipEventRecord(sl->join_net, D.net_stream));
    HIPCHK(hipEventRecord(sl->join, D.sel_stream));
    HIPCHK(hipStreamWaitEvent(s, sl->join_net, 0));
    HIPCHK(hipStreamWaitEvent(s, sl->join, 0));
    return 0;
    };
    if (rounds > 0 && launch_once(rounds)) return -1;
    if (!shared_budget && launch_once(0)) return -1;
    return 0;
}

// include/azg.h: the pipeline for a Splendor-2p forest with the V80 net
extern "C" int azg_forest_async_rounds_v80_h2(azg_forest* f, uint8_t* leaf_valid, uint8_t* needs_eval, float* pi, float* v, int noise_stride,
                                              const void* const* w, const float* descale, int rounds, int n_net, int n_sel, int batch_wait_ticks,
                                              int shared_budget, void* stream) {
    return async_rounds_impl("azg_forest_async_rounds_v80_h2", 0, 0, f, leaf_valid, needs_eval, pi, v, noise_stride, w, descale, rounds, n_net, n_sel,
                             batch_wait_ticks, shared_budget, stream);
}
// include/azg.h: the pipeline for Splendor 3 / 4 players and Azul with their MobileNet-1d nets (geometry = AZG_NET_* of azg_nn_mb1d_forward_h2)
extern "C" int azg_forest_async_rounds_mb1d_h2(azg_forest* f, int geometry, uint8_t* leaf_valid, uint8_t* needs_eval, float* pi, float* v, int noise_stride,
                                               const void* const* w, const float* descale, int rounds, int n_net, int n_sel, int batch_wait_ticks,
                                               int shared_budget, void* stream) {
    const int kind = geometry == AZG_NET_SPLENDOR3 ? 2 : geometry == AZG_NET_SPLENDOR4 ? 3 : geometry == AZG_NET_AZUL ? 4 : -1;
    if (kind < 0) return fail("azg_forest_async_rounds_mb1d_h2: geometry must be AZG_NET_SPLENDOR3, AZG_NET_SPLENDOR4 or AZG_NET_AZUL");
    return async_rounds_impl("azg_forest_async_rounds_mb1d_h2", kind, 0, f, leaf_valid, needs_eval, pi, v, noise_stride, w, descale, rounds, n_net, n_sel,
                             batch_wait_ticks, shared_budget, stream);
}
// include/azg.h: the pipeline for a Santorini no-gods forest with the V89 net (14 pointers of azg_nn_conv5_forward_h2, its descale)
extern "C" int azg_forest_async_rounds_conv5_h2(azg_forest* f, uint8_t* leaf_valid, uint8_t* needs_eval, float* pi, float* v, int noise_stride,
                                                const float* const* w, float descale, int rounds, int n_net, int n_sel, int batch_wait_ticks,
                                                int shared_budget, void* stream) {
    return async_rounds_impl("azg_forest_async_rounds_conv5_h2", 1, 0, f, leaf_valid, needs_eval, pi, v, noise_stride, (const void* const*)w, &descale, rounds,
                             n_net, n_sel, batch_wait_ticks, shared_budget, stream);
}

// include/azg_testaids.h: the pipeline with the integer hash-net as its evaluator, for every game that has a descent kernel here
extern "C" int azg_forest_async_rounds_hashnet(azg_forest* f, uint8_t* leaf_valid, uint8_t* needs_eval, float* pi, float* v, int noise_stride, int rounds,
                                               int n_net, int n_sel, int batch_wait_ticks, int shared_budget, void* stream) {
    if (!f) return fail("azg_forest_async_rounds_hashnet: null argument");
    int game = 0, variant = 0;
    double alpha = 0.0;
    (void)azg_forest_dev_internal(f, &game, &variant, &alpha);
    const int kind = game == AZG_SPLENDOR ? (variant == 2 ? 0 : variant == 3 ? 2 : variant == 4 ? 3 : -1)
                     : game == AZG_SANTORINI ? (variant == 1 ? 1 : -1) : game == AZG_AZUL ? 4 : -1;
    if (kind < 0) return fail("azg_forest_async_rounds_hashnet: the pipeline has descent kernels for Splendor 2 - 4 players, Santorini without gods and Azul");
    return async_rounds_impl("azg_forest_async_rounds_hashnet", kind, 1, f, leaf_valid, needs_eval, pi, v, noise_stride, nullptr, nullptr, rounds, n_net, n_sel,
                             batch_wait_ticks, shared_budget, stream);
}

#endif  // AZG_ASYNC_PART_NET