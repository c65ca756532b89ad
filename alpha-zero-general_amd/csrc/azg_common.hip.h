// azg_common.hip.h -- wave-level primitives for the gfx950 self-play engine (one 64-lane wavefront per tree / per state).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AZG_WAVE 64
#define AZG_NONE 0xFFFFFFFFu
#define AZG_NANQ (-42.0)     /* MCTS.py:11 sentinel for "never visited" */
#define AZG_EPS 1e-8          /* MCTS.py:10 */
#define AZG_MAX_PLAYERS_DEV 4

namespace azg {

#ifdef AZG_WAVE_LOCAL_SYNC
// (the translation unit of the per-CU round kernel: the lane id is laundered through an empty volatile asm, so that the compiler cannot
// hoist the descent's lane-derived address arithmetic out of the kernel's round loop -- and keep it alive, spilled, across the net phase)
__device__ __forceinline__ int lane_id() { int l = threadIdx.x & 63; asm volatile("" : "+v"(l)); return l; }
#else
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
#endif

// LDS ordering inside ONE wavefront (every forest / env kernel gives a tree or a state to one wave): a wavefront-scope fence -- the LDS and
// the vector-memory path serve one wave's requests in order, so only the COMPILER must be kept from reordering -- plus the wave barrier
// intrinsic.  Rounds 1-3 used __syncthreads() here (the s_barrier is dropped for a single-wave workgroup, its waitcnt is what mattered):
// that waitcnt is `vmcnt(0) lgkmcnt(0)`, i.e. every LDS hand-over between lanes also DRAINED every global load and store the wave had
// in flight -- 46 full drains in k_select, which serialised each memory round trip with the LDS work that could have run under it.
// AZG_SYNC_DRAINS restores the old behaviour (A/B runs).  In a workgroup of several tree waves (azg_fused.hip.h) a workgroup barrier
// would be wrong anyway.
__device__ __forceinline__ void wave_sync() {
#ifdef AZG_SYNC_DRAINS
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

// RNG contract of include/azg.h
struct Rng {
    uint64_t seed, stream, counter;
    __device__ __forceinline__ double u01() {
        uint64_t x = mix64(mix64(mix64(seed ^ 0x9E3779B97F4A7C15ULL) + stream) + counter);
        counter++;
        return (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
};

__device__ __forceinline__ uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    return __longlong_as_double((long long)shfl_xor_u64((uint64_t)__double_as_longlong(v), m));
}
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double bcast_f64(double v, int src) {
    return __longlong_as_double((long long)bcast_u64((uint64_t)__double_as_longlong(v), src));
}

// ---- DPP cross-lane moves (VALU data path, no LDS crossbar round trip like ds_bpermute) ----
// dpp_ctrl: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140,
// row_bcast15 = 0x142 (lane 15 of each row -> next row), row_bcast31 = 0x143 (lane 31 -> rows 2,3)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
    return ((uint64_t)dpp_u32<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32) | dpp_u32<CTRL, ROW_MASK>((uint32_t)v);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
}
// wave-wide reduction with a commutative, associative, idempotent-on-self-free op (each lane is combined exactly once):
// 4 butterfly steps inside each row of 16, then two row broadcasts; the result is read from lane 63 (wave-uniform).
#define AZG_DPP_REDUCE_U64(v, OP)                                                   \
    do {                                                                            \
        uint64_t o_;                                                                \
        o_ = dpp_u64<0xB1>(v); v = OP(v, o_);                                       \
        o_ = dpp_u64<0x4E>(v); v = OP(v, o_);                                       \
        o_ = dpp_u64<0x141>(v); v = OP(v, o_);                                      \
        o_ = dpp_u64<0x140>(v); v = OP(v, o_);                                      \
    } while (0)

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#define AZG_OP_ADD(a, b) ((a) + (b))
    AZG_DPP_REDUCE_U64(v, AZG_OP_ADD);                 // every lane of a row holds its row's sum
#undef AZG_OP_ADD
    return readlane_u64(v, 0) + readlane_u64(v, 16) + readlane_u64(v, 32) + readlane_u64(v, 48);
}
// maximum of an f64 over the wave (no NaNs: -inf for idle lanes), wave-uniform result.  One v_max_f64 per combine, written as asm:
// fmax() lowers to llvm.maxnum, which quiets a possible signalling NaN in EACH operand first -- two extra v_max_f64 x, x per step, 45
// instructions per reduction instead of 23 -- and the inputs here are PUCT scores and -inf.
__device__ __forceinline__ double max_f64_nonan(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double wave_max_f64(double x) {
    uint64_t v = (uint64_t)__double_as_longlong(x);
#define AZG_OP_FMAX(a, b) ((uint64_t)__double_as_longlong(max_f64_nonan(__longlong_as_double((long long)(a)), __longlong_as_double((long long)(b)))))
    AZG_DPP_REDUCE_U64(v, AZG_OP_FMAX);
#undef AZG_OP_FMAX
    const double a = __longlong_as_double((long long)readlane_u64(v, 0)), b = __longlong_as_double((long long)readlane_u64(v, 16));
    const double c = __longlong_as_double((long long)readlane_u64(v, 32)), d = __longlong_as_double((long long)readlane_u64(v, 48));
    return max_f64_nonan(max_f64_nonan(a, b), max_f64_nonan(c, d));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}

// argmax with lowest-index tie-break == the reference's ascending scan with strict '>' (MCTS.py:216-228)
__device__ __forceinline__ void wave_argmax_f64(double& u, int& idx) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double ou = shfl_xor_f64(u, m);
        int oi = __shfl_xor(idx, m, 64);
        bool take = (ou > u) || (ou == u && oi < idx);
        u = take ? ou : u;
        idx = take ? oi : idx;
    }
}

// 64-bit hash of a zero-padded state held in LDS as n_dwords dwords: multilinear sum_i a_i * x_i over Z_2^64 with a
// distinct odd 32-bit multiplier per position (one v_mad_u64_u32 per dword; two states that differ in a single dword can
// never collide), reduced over the wave and avalanched once on the wave-uniform result.  Only the table placement and
// the 10-bit tag depend on it: node identity is always the full state.
__device__ __forceinline__ uint64_t wave_hash_state(const uint32_t* lds_dwords, int n_dwords) {
    uint64_t acc = 0;
    for (int i = lane_id(); i < n_dwords; i += 64) {
        const uint32_t a = ((uint32_t)(i + 1) * 0x9E3779B1u) | 1u;
        acc += (uint64_t)lds_dwords[i] * (uint64_t)a;
    }
    acc = wave_sum_u64(acc);
    return mix64(acc ^ 0xD6E8FEB86659FD93ULL);
}

// Env step by lane 0 on the LDS state (games whose make_move is lane-serial): all lanes call, the next player and the
// RNG counter come back wave-uniform, the LDS state is synchronised.
template <class G>
__device__ __forceinline__ int lane0_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
    int np = 0;
    if (lane_id() == 0) np = G::make_move(st, move, player, seed, rng);
    np = __builtin_amdgcn_readfirstlane(np);
    rng.counter = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rng.counter >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rng.counter);
    wave_sync();
    return np;
}

__device__ __forceinline__ int first_lane(uint64_t ballot) { return __ffsll((unsigned long long)ballot) - 1; }

template <int N> struct RoundUp16 { static constexpr int value = (N + 15) / 16 * 16; };

// LDS scratch a game's make_move wants BEHIND the state it is given (G::MOVE_SCRATCH bytes at st + G::SP; 0 unless the game says so): the
// forest's Smem has its `tmp` there (forest.hip.h), the env-step kernel allocates it with the state (kernels.hip.h k_env_next_state)
template <class G, class = void> struct MoveScratch { static constexpr int value = 0; };
template <class G> struct MoveScratch<G, std::void_t<decltype(G::MOVE_SCRATCH)>> { static constexpr int value = G::MOVE_SCRATCH; };

}  // namespace azg
