// azg_common.cuh -- wave-level primitives for the gfx950 self-play engine (one 64-lane wavefront per tree / per state).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AZG_WAVE 64
#define AZG_NONE 0xFFFFFFFFu
#define AZG_NANQ (-42.0)     /* MCTS.py:11 sentinel for "never visited" */
#define AZG_EPS 1e-8          /* MCTS.py:10 */
#define AZG_MAX_PLAYERS_DEV 4

namespace azg {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// LDS ordering inside a single-wave workgroup: s_barrier is a no-op for one wave, the waitcnt it carries is what matters.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

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

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    return v;
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

// 64-bit hash of a zero-padded state held in LDS as n_dwords dwords (order-sensitive through the position salt).
__device__ __forceinline__ uint64_t wave_hash_state(const uint32_t* lds_dwords, int n_dwords) {
    uint64_t acc = 0;
    for (int i = lane_id(); i < n_dwords; i += 64)
        acc += mix64((uint64_t)lds_dwords[i] + ((uint64_t)(i + 1) << 32) * 0x9E3779B1ULL + 0x2545F4914F6CDD1DULL * (uint64_t)(i + 1));
    acc = wave_sum_u64(acc);
    return mix64(acc ^ 0xD6E8FEB86659FD93ULL);
}

__device__ __forceinline__ int first_lane(uint64_t ballot) { return __ffsll((unsigned long long)ballot) - 1; }

template <int N> struct RoundUp16 { static constexpr int value = (N + 15) / 16 * 16; };

}  // namespace azg
