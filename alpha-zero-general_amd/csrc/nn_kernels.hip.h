// nn_kernels.hip.h -- fp32 policy/value net building blocks for gfx950 (the only MFMA work on the self-play path).
//
// The reference evaluates the net with ONNX Runtime / torch on the CPU, one leaf at a time or 8 at a time
// (GenericNNetWrapper.py:94-157).  Here one lock-step round evaluates T leaves at once; the V80 network
// (splendor/SplendorNNet.py:262-283,397-440) is a chain of *skinny* GEMMs (M = 7*T rows, K,N in {56,168,392,81}) plus
// per-(sample,channel) glue, so instead of library GEMM calls (which pick poor tiles for K = 56) there are three kernels (+ a layout kernel):
//
//   k_linear      out = act(A' @ W + bias) (+ R)      MFMA v_mfma_f32_16x16x4_f32 (exact f32, == an fmaf chain), weights
//                                                      resident in LDS, A fragments straight from HBM as float4;
//                                                      A' = A optionally scaled per (sample,k) -- this fuses the
//                                                      SqueezeExcitation multiply into the project GEMM's operand
//   k_dw_pool     depthwise Linear(7->7) over the token axis + folded BN + activation + SE squeeze (avg / max)
//                 (the two SE fully-connected layers E -> Q -> E are two more k_linear launches, hardsigmoid epilogue)
//   k_heads_out   masked softmax of the policy logits (exp(log_softmax), GenericNNetWrapper.py:107) and the value tail
#pragma once
// the non-template kernels of this header get internal linkage in a translation unit that only wants its device helpers (azg_async.hip)
#ifndef AZG_NN_KERNEL
#define AZG_NN_KERNEL
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

// The net has a tolerance contract (|pi|,|v| within 1e-5 of the reference's fp32 outputs), not a bit-exact one: FMA
// contraction is allowed in this header (the MCTS / env code keeps -ffp-contract=off).
#pragma clang fp contract(fast)

namespace azg {

// The thread index as the net kernels' helper functions see it.  In the translation unit of the pipeline's PERSISTENT net kernel
// (azg_async.hip defines AZG_NN_OPAQUE_TID) it is laundered through an empty asm at every use: the forward runs inside one long loop there,
// everything derived from the raw threadIdx.x is loop-invariant, and the compiler hoisted ~50 per-lane operand addresses of the Santorini
// forward to the kernel's entry and SPILLED them (k_async_net<NetC5>: 196 B of scratch, one scratch reload per basic block of the
// forward, each waiting with vmcnt(0) behind the weight prefetch; the stand-alone kernel has no scratch).  Opaque, they are recomputed
// where they are used -- a handful of VALU instructions per helper call.
__device__ __forceinline__ int nn_tid() {
#ifdef AZG_NN_OPAQUE_TID
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
#else
    return (int)threadIdx.x;
#endif
}


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));     // v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 operands

// wave-wide f32 max / sum through DPP butterflies (rows of 16) + four readlanes; every lane gets the result
// ---- split-precision operand helpers (bf16 x 3, see nn_conv5x5.hip.h) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_rn(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = bf16_rn(x);
    float r = x - __uint_as_float(h << 16);
    m = bf16_rn(r);
    r = r - __uint_as_float(m << 16);
    l = bf16_rn(r);
}
// two values at once on the hardware converter (v_cvt_pk_bf16_f32, round to nearest even like bf16_rn): word = bf16(a) | bf16(b) << 16
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2_t));
}
__device__ __forceinline__ void split3x2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = bf16_pk(a, b);
    a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xFFFF0000u);
    m = bf16_pk(a, b);
    a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xFFFF0000u);
    l = bf16_pk(a, b);
}
// byte offset of (row, 16-byte chunk q) inside one plane
__device__ __forceinline__ int pl_off(int row, int q) { return row * 128 + ((q ^ (row & 7)) << 4); }
// four consecutive channels (ch0 % 4 == 0) of one cell -> the three planes
__device__ __forceinline__ void store_split4(uint8_t* planes, int plane_bytes, int row, int ch0, float4 o) {
    uint32_t h[2], m[2], l[2];
    split3x2(o.x, o.y, h[0], m[0], l[0]); split3x2(o.z, o.w, h[1], m[1], l[1]);
    uint8_t* dst = planes + pl_off(row, ch0 >> 3) + ((ch0 & 4) << 1);
    *(uint2*)dst = make_uint2(h[0], h[1]);
    *(uint2*)(dst + plane_bytes) = make_uint2(m[0], m[1]);
    *(uint2*)(dst + 2 * plane_bytes) = make_uint2(l[0], l[1]);
}
__device__ __forceinline__ float bf16_lo_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi_f32(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ float4 load_split4(const uint8_t* planes, int plane_bytes, int row, int ch0) {
    const uint8_t* src = planes + pl_off(row, ch0 >> 3) + ((ch0 & 4) << 1);
    const uint2 h = *(const uint2*)src, m = *(const uint2*)(src + plane_bytes), l = *(const uint2*)(src + 2 * plane_bytes);
    return make_float4((bf16_lo_f32(h.x) + bf16_lo_f32(m.x)) + bf16_lo_f32(l.x), (bf16_hi_f32(h.x) + bf16_hi_f32(m.x)) + bf16_hi_f32(l.x),
                       (bf16_lo_f32(h.y) + bf16_lo_f32(m.y)) + bf16_lo_f32(l.y), (bf16_hi_f32(h.y) + bf16_hi_f32(m.y)) + bf16_hi_f32(l.y));
}


__device__ __forceinline__ float nn_wave_max(float x) {
    x = fmaxf(x, __uint_as_float(dpp_u32<0xB1>(__float_as_uint(x)))); x = fmaxf(x, __uint_as_float(dpp_u32<0x4E>(__float_as_uint(x))));
    x = fmaxf(x, __uint_as_float(dpp_u32<0x141>(__float_as_uint(x)))); x = fmaxf(x, __uint_as_float(dpp_u32<0x140>(__float_as_uint(x))));
    const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 0)), b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 16));
    const float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 32)), d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float nn_wave_sum(float x) {
    x += __uint_as_float(dpp_u32<0xB1>(__float_as_uint(x))); x += __uint_as_float(dpp_u32<0x4E>(__float_as_uint(x)));
    x += __uint_as_float(dpp_u32<0x141>(__float_as_uint(x))); x += __uint_as_float(dpp_u32<0x140>(__float_as_uint(x)));
    const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 0)), b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 16));
    const float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 32)), d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 48));
    return (a + b) + (c + d);
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2, ACT_HSIGMOID = 3 };

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == ACT_RELU) return x > 0.f ? x : 0.f;
    if (act == ACT_HSWISH) { float t = fminf(fmaxf(x + 3.f, 0.f), 6.f); return x * t * (1.f / 6.f); }
    if (act == ACT_HSIGMOID) return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    return x;
}
// activation of two packed values
__device__ __forceinline__ f32x2 act_apply2(f32x2 x, int act) {
    if (act == ACT_RELU) return f32x2{fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    if (act == ACT_HSWISH || act == ACT_HSIGMOID) {
        // relu6(x + 3) / 6 == clamp(x / 6 + 1 / 2, 0, 1): ONE packed FMA with the CLAMP output modifier (round 4; was a packed add, a
        // v_med3_f32 per component and a packed multiply by 1 / 6).  The compiler does not fold a clamp into v_pk_fma_f32: inline asm.
        f32x2 t;
        const f32x2 k = f32x2{1.f / 6.f, 1.f / 6.f};
        asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,0,0] clamp" : "=v"(t) : "v"(x), "s"(k));
        return act == ACT_HSWISH ? x * t : t;
    }
    return x;
}
__device__ __forceinline__ float hardsigmoid(float x) { return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); }

// out[M][N] = act(A'[M][K] @ W + bias[N]) (+ R[M][N]);  A'[r][k] = A[r][k] * rowscale[r / rpg][k] if rowscale.
//
// W is pre-padded by the host to Wp[Kp][NP] (Kp = K rounded up to 16, NP = 16*NT, zero padding) so that staging and
// fragment reads need no bounds logic.  MFMA 16x16x4 f32: lane l supplies A[row = l&15][k'] and B[k'][col = l&15] for the
// k' of its lane group g = l>>4.  The reduction order over K is free as long as A and B agree, so group g takes
// k' = 16*s + 4*g + j (j = 0..3) of K-chunk s: every lane then reads ONE float4 of its A row per chunk straight from
// global memory (16 rows x 64 contiguous bytes per wave instruction) and A never goes through LDS.
//   KSPLIT == 0 (tall M): Wp resident in LDS, each wave walks its own 16-row tiles, no barrier in the tile loop.
//   KSPLIT == 1 (M ~ T, long K: the flatten->Linear heads): one 16-row tile per workgroup, the 4 waves split the K
//               chunks, B fragments come straight from L2, partial sums are reduced through LDS.
template <int NT, int KSPLIT, int NCH>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                                const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                                const float* __restrict__ rowscale, int rpg, float* __restrict__ out,
                                                int ldc, int M, int K, int Kp, int N, int act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NP = NT * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int n_chunks = Kp >> 4;
    const int n_tiles = (M + 15) / 16;
    if (!KSPLIT) {
        // async global->LDS DMA (global_load_lds_dwordx4): 1 KiB per wave instruction, destination = uniform base +
        // lane*16, all requests in flight at once (Kp*NP*4 is a multiple of 1 KiB because Kp and NP are multiples of 16)
        const int n_kb = (Kp * NP) >> 8;
        for (int c = wave; c < n_kb; c += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wp + (size_t)c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(smem + (size_t)c * 256), 16, 0, 0);
        __syncthreads();
    }
    const int tile_step = KSPLIT ? gridDim.x : gridDim.x * 4;
    for (int tile = KSPLIT ? blockIdx.x : blockIdx.x * 4 + wave; tile < n_tiles; tile += tile_step) {
        const int row = tile * 16 + r16;
        const bool row_ok = row < M;
        const float* arow = A + (size_t)(row_ok ? row : 0) * lda;
        const float* srow = rowscale ? rowscale + (size_t)((row_ok ? row : 0) / rpg) * K : nullptr;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // all A chunks of this tile are requested up front (NCH float4 per lane) so the HBM latency is paid once
        float4 av[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int s = KSPLIT ? wave + 4 * c : c;
            const int k0 = 16 * s + 4 * g;
            av[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && s < n_chunks && k0 < K) av[c] = *(const float4*)(arow + k0);
        }
        if (srow) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int s = KSPLIT ? wave + 4 * c : c;
                const int k0 = 16 * s + 4 * g;
                if (row_ok && s < n_chunks && k0 < K) {
                    const float4 sc = *(const float4*)(srow + k0);
                    av[c].x *= sc.x; av[c].y *= sc.y; av[c].z *= sc.z; av[c].w *= sc.w;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int s = KSPLIT ? wave + 4 * c : c;
            if (s < n_chunks) {
                const int k0 = 16 * s + 4 * g;
                const float* wb = (KSPLIT ? Wp : smem) + (size_t)k0 * NP + r16;
                const float a4[4] = {av[c].x, av[c].y, av[c].z, av[c].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], wb[j * NP + nt * 16], acc[nt], 0, 0, 0);
                }
            }
        }
        if (KSPLIT) {
            // reduce the 4 waves' partial tiles through LDS: red[wave][nt][reg][lane]
            __syncthreads();
            float* red = smem;
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[((wave * NT + nt) * 4 + r) * 64 + lane] = acc[nt][r];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        acc[nt][r] = ((red[((0 * NT + nt) * 4 + r) * 64 + lane] + red[((1 * NT + nt) * 4 + r) * 64 + lane]) +
                                      red[((2 * NT + nt) * 4 + r) * 64 + lane]) + red[((3 * NT + nt) * 4 + r) * 64 + lane];
            }
        }
        if (!KSPLIT || wave == 0) {
            // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int col = nt * 16 + r16;
                if (col < N) {
                    const float b = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int orow = tile * 16 + g * 4 + r;
                        if (orow < M) {
                            float v = act_apply(acc[nt][r] + b, act);
                            if (R) v += R[(size_t)orow * ldr + col];
                            out[(size_t)orow * ldc + col] = v;
                        }
                    }
                }
            }
        }
    }
}

// Weight-stationary variant (the one the net uses): out^T = W^T @ A'^T.  The WEIGHT tile is the MFMA "A" operand and
// lives in registers for the whole kernel (NCH*4 VGPRs per lane for one 16-column tile), activations are the "B" operand:
// lane (g = l>>4, r = l&15) reads ONE float4 of activation row `tile*16 + r` per 16-wide K chunk straight from HBM/L2.
// No LDS, no barrier; the C/D layout (i = 4*g + reg -> output column, j = r -> row) leaves every lane with 4 CONSECUTIVE
// output columns of its row, so bias / activation / residual / store are float4-wide.  A workgroup's 4 waves take 4
// different column tiles of the same row tiles (activation re-reads hit L1); grid.y covers the remaining column tiles.
// The next row tile's activations (and residual) are requested before the current tile's MFMAs (software prefetch).
template <int NCH>
__global__ __launch_bounds__(256) void k_linear_ws(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                                   int NP, const float* __restrict__ biasp, const float* __restrict__ R,
                                                   int ldr, const float* __restrict__ rowscale, int rpg,
                                                   float* __restrict__ out, int ldc, int M, int K, int N, int act,
                                                   int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int nt = blockIdx.y * 4 + wave;
    if (nt * 16 >= N) return;
    float w[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) w[c][j] = Wp[(size_t)(16 * c + 4 * g + j) * NP + nt * 16 + r16];
    const int col0 = nt * 16 + 4 * g;
    const float4 b4 = biasp ? *(const float4*)(biasp + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_tiles = (M + 15) / 16;
    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(n_tiles, t_begin + tiles_per_wg);
    float4 a_nxt[NCH];
    float4 r_nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int tile) {
        const int row = tile * 16 + r16;
        const bool ok = row < M;
        const float* arow = A + (size_t)(ok ? row : 0) * lda;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int k0 = 16 * c + 4 * g;
            a_nxt[c] = (ok && k0 < K) ? *(const float4*)(arow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (rowscale) {
            const float* srow = rowscale + (size_t)((ok ? row : 0) / rpg) * K;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int k0 = 16 * c + 4 * g;
                if (ok && k0 < K) {
                    const float4 sc = *(const float4*)(srow + k0);
                    a_nxt[c].x *= sc.x; a_nxt[c].y *= sc.y; a_nxt[c].z *= sc.z; a_nxt[c].w *= sc.w;
                }
            }
        }
        if (R) r_nxt = (ok && col0 + 3 < ((N + 3) & ~3)) ? *(const float4*)(R + (size_t)row * ldr + col0)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; tile++) {
        float4 a_cur[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) a_cur[c] = a_nxt[c];
        const float4 r_cur = r_nxt;
        if (tile + 1 < t_end) fetch(tile + 1);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][0], a_cur[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][1], a_cur[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][2], a_cur[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][3], a_cur[c].w, acc, 0, 0, 0);
        }
        const int row = tile * 16 + r16;
        if (row < M) {
            float4 v;
            v.x = act_apply(acc[0] + b4.x, act) + r_cur.x;
            v.y = act_apply(acc[1] + b4.y, act) + r_cur.y;
            v.z = act_apply(acc[2] + b4.z, act) + r_cur.z;
            v.w = act_apply(acc[3] + b4.w, act) + r_cur.w;
            float* o = out + (size_t)row * ldc + col0;
            if (col0 + 3 < N) *(float4*)o = v;
            else {
                if (col0 + 0 < N) o[0] = v.x;
                if (col0 + 1 < N) o[1] = v.y;
                if (col0 + 2 < N) o[2] = v.z;
            }
        }
    }
}

// H[b*L + m][c] <- act( sd[c] * sum_l Wd[m][l] * H[b*L + l][c] + bd[c] );  pooled[b][c] = mean_m / max_m of the result
// (L = 7 tokens for Splendor, 6 for Azul)
// (LinearNormActivation depthwise + SqueezeExcitation1d pooling, SplendorNNet.py:148-187).  One (sample, channel) per
// thread, channel-fastest => coalesced; wide grid for memory-level parallelism (2 x 19 MB of traffic at T = 4096).
template <int L>
__global__ __launch_bounds__(256) void k_dw_pool(float* __restrict__ H, int ldh, const float* __restrict__ Wd,
                                                 const float* __restrict__ sd, const float* __restrict__ bd,
                                                 float* __restrict__ pooled, int B, int E, int act, int pool_max) {
    __shared__ float w[L * L];
    if (threadIdx.x < L * L) w[threadIdx.x] = Wd[threadIdx.x];
    __syncthreads();
    const long long total = (long long)B * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / E), c = (int)(i - (long long)b * E);
        float* base = H + (size_t)b * L * ldh + c;
        float in[L];
#pragma unroll
        for (int l = 0; l < L; l++) in[l] = base[(size_t)l * ldh];
        const float s = sd[c], bb = bd[c];
        float pool = pool_max ? -INFINITY : 0.f;
#pragma unroll
        for (int m = 0; m < L; m++) {
            float a = 0.f;
#pragma unroll
            for (int l = 0; l < L; l++) a += w[m * L + l] * in[l];
            a = act_apply(a * s + bb, act);
            base[(size_t)m * ldh] = a;
            pool = pool_max ? fmaxf(pool, a) : pool + a;
        }
        pooled[(size_t)b * E + c] = pool_max ? pool : pool / (float)L;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused InvertedResidual1d block (SplendorNNet.py:189-202 with :148-187), one workgroup = 16 samples = 112 token rows:
//   x[112][C=56] --expand GEMM+BN+act--> h[112][E=168] --depthwise Linear(7->7)+BN+act, SE squeeze--> pooled[16][E]
//   --SE fc1+ReLU--> [16][Q=40] --SE fc2+Hardsigmoid--> sc[16][E];   out = (h * sc) @ Wp + BN + x      (project + residual)
// The expanded activations (19 MB per block at T = 4096) never leave the CU: they live in LDS between the phases, so a
// block costs one read of x and one write of out (2 x 6.4 MB) instead of ~115 MB of HBM traffic for the unfused chain.
// All four GEMMs use MFMA f32 16x16x4 with the WEIGHTS as the A operand held in registers (prefetched from L2 at kernel
// start, overlapping the x load) and the activations as the B operand read as float4 from LDS; 12 waves per workgroup:
//   expand   11 column tiles x 7 row tiles      wave w < 11 owns column tile w
//   SE fc1   3 column tiles x 1 row tile         waves 0..2
//   SE fc2   11 column tiles x 1 row tile        waves 0..10
//   project  4 column tiles x 7 row tiles        wave w owns column tile w&3 and row tiles {w>>2, (w>>2)+3, (w>>2)+6}
// Hard-wired to the V80 geometry C = 56, E = 168, Q = 40; the zero-padded weights We[64][176], W1[176][48], W2[48][176],
// Wp[176][64] are passed in MFMA fragment order (FRAG above).
struct V80BlockW {
    const float *We, *be, *Wd, *sd, *bd, *W1, *b1, *W2, *b2, *Wp, *bp;     // be/b1/b2/bp zero-padded to the padded widths
};
// What the whole-net fusion hangs onto the three blocks (k_v80_block MODE 1..3):
//   MODE 1 (trunk):       boards int8 [B][56][7] -> f32 tile (layout change in LDS) -> first_layer Linear(56,56)+BN
//                         (SplendorNNet.py:397-401) -> block -> xout in HBM
//   MODE 2 (policy head): block -> Flatten -> Linear(392,81)+ReLU -> Linear(81,81) -> masked softmax -> pi in HBM
//   MODE 3 (value head):  block -> Flatten -> Linear(392,P)+ReLU -> Linear(P,P) -> tanh -> v in HBM   (:414-440)
// The block output tile stays in LDS (in place of x, row stride 60), so the flatten index of the head weights is
// k = l*60 + c (rows with c >= 56 are zero): Wh1 is [432][96] (policy) / [432][16] (value), zero padded.
struct V80NetW {
    const float *W0, *b0;                  // first layer [64][64], [64]
    const float *Wh1, *bh1, *Wh2, *bh2;    // policy: [432][96], [96], [96][96], [96];  value: [432][16], [16], [P][P], [P]
                                           // (W0, Wh1 and the policy Wh2 in fragment order; the value Wh2 is plain)
};

// weight operands are stored in MFMA fragment order: frag[tile nt][K chunk c][lane][j] = W[16c + 4*(lane>>4) + j][16nt + (lane&15)],
// so a wave fetches one chunk of one column tile as a single 1 KiB global_load_dwordx4.  Where K = 8 mod 16 (56, 168) the
// last chunk is stored "half": frag[nt][c_last][lane][j] = W[16c + 2*(lane>>4) + j] for j < 2 (j >= 2 unused), so that
// chunk costs two MFMAs instead of four
#define FRAG(ptr, NCH, nt, c) (*(const float4*)((ptr) + ((((size_t)(nt) * (NCH) + (c)) * 64 + lane) << 2)))

#ifdef AZG_NN_PHASE_TIMES
static __device__ long long g_v80_phase[4][16];
#define AZG_PH(k) do { if (blockIdx.x == 7 && threadIdx.x == 0) g_v80_phase[MODE][k] = clock64(); } while (0)
#else
#define AZG_PH(k)
#endif

// The block (+ optional first layer in front, + optional head tail behind) on one 16-sample tile.  X = the tile's [112][60]
// f32 buffer in LDS, H = the rest of the workgroup's LDS ([112][172] + 2 x [16][172] + [16][52] + 64 floats).
//   INLDS:  the input tile is already in X (whole-net kernel) instead of xin in HBM
//   OUTLDS: the block output replaces X in place (always the case for the head MODEs); SAVE: and is copied to xsave
//   SPX:    the input tile lives in LDS as three bf16 planes [112][64] (hi + mid + lo, the layout of nn_conv5x5.hip.h) instead of
//           the f32 [112][60] tile, written split once by the producing epilogue, and the expand GEMM runs on bf16 x 3 operands (six
//           v_mfma_f32_16x16x32_bf16 per product; W.We = [11 tiles][2 K chunks of 32][3 planes][64 lanes][8] bf16).  The trunk block
//           rewrites the planes in place; the head blocks leave them alone (the value head reuses the trunk output) and put their
//           f32 output O [112][60] over the dead pooled / SE-hidden buffers for the head tail.
template <int ACT, int POOLMAX, int MODE, bool INLDS, bool OUTLDS, bool SAVE, bool SPX = false>
__device__ __forceinline__ void v80_block_body(float* X, float* H, float* xsave, const float* __restrict__ xin,
                                               float* __restrict__ xout, const V80BlockW& W, int B,
                                               const int8_t* __restrict__ boards, const V80NetW& N,
                                               const uint8_t* __restrict__ valid, float* __restrict__ pi_out,
                                               float* __restrict__ v_out, int P) {
    constexpr int C = 56, E = 168, NS = 16, ROWS = NS * 7, XS = 60, HS = 172, QS = 52, FK = 7 * XS /* 420 */, A = 81;
    static_assert(!SPX || ((MODE == 1 || INLDS) && !SAVE), "SPX: whole-net kernel only");
    // (SPX: SC first, so that pooled + SE hidden + depthwise weights + 12 KB behind them form the 26.9 KB of the head blocks' output tile)
    float* SC = SPX ? H + ROWS * HS : H + ROWS * HS + NS * HS;          // scales [NS][HS]
    float* PL = SPX ? SC + NS * HS : H + ROWS * HS;                     // pooled [NS][HS]
    float* SH = SPX ? PL + NS * HS : SC + NS * HS;                      // SE hidden [NS][QS]
    float* WD = SH + NS * QS;           // [49]
    float* OT = PL;                     // SPX, head blocks: the block output O [ROWS][XS] f32 (PL, SH, WD are dead by the project phase)
    uint8_t* XPL = (uint8_t*)X;         // SPX: the three planes
    constexpr int PB = ROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int b0 = blockIdx.x * NS;
    const int nrows = min(ROWS, (B - b0) * 7);
    const int nt_p = wave & 3;
    AZG_PH(0);

    float4 we[4], w2r[3], w1r[11], wpr[11];
    uint4 weS[6];                       // SPX: [K chunk][plane] fragments of the wave's expand tile
    const int nt_e = wave < 11 ? wave : 0, nt_1 = wave < 3 ? wave : 0;
    auto load_we = [&]() {
        if (SPX) {
#pragma unroll
            for (int k = 0; k < 6; k++) weS[k] = ((const uint4*)W.We)[((size_t)nt_e * 6 + k) * 64 + lane];
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) we[c] = FRAG(W.We, 4, nt_e, c);
        }
    };
    auto load_w2_wp = [&]() {
#pragma unroll
        for (int c = 0; c < 3; c++) w2r[c] = FRAG(W.W2, 3, nt_e, c);
#pragma unroll
        for (int c = 0; c < 11; c++) wpr[c] = FRAG(W.Wp, 11, nt_p, c);
    };
    if (MODE == 1) {
        // ---- P0': board tile (int8, [s][c][l]) -> X0[s*7+l][c] f32 (aliases H), then first layer -> X ----
        float* X0 = H;
        const int nb = min(NS, B - b0);
        const uint32_t* bsrc = (const uint32_t*)(boards + (size_t)b0 * (7 * C));
        uint32_t bv[3];                                   // the board tile is requested before any weight
#pragma unroll
        for (int k = 0; k < 3; k++) { const int i = tid + 768 * k; bv[k] = (i < NS * (7 * C / 4) && i / (7 * C / 4) < nb) ? bsrc[i] : 0u; }
        float4 w0r[4];
#pragma unroll
        for (int c = 0; c < 4; c++) w0r[c] = FRAG(N.W0, 4, nt_p, c);
        const float4 b04 = *(const float4*)(N.b0 + nt_p * 16 + 4 * g);
        load_we();                                        // block weights stream in behind the first layer
        load_w2_wp();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + 768 * k;
            if (i < NS * (7 * C / 4)) {
                const int s = i / (7 * C / 4), d = i - s * (7 * C / 4);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int idx = 4 * d + q, c = idx / 7, l = idx - c * 7;
                    X0[(s * 7 + l) * XS + c] = (float)(int8_t)(bv[k] >> (8 * q));
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int rt = wave >> 2; rt < 7; rt += 3) {
            const float* xr = X0 + (rt * 16 + r16) * XS + 4 * g;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float4 a = *(const float4*)(xr + 16 * c);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[c].x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[c].y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[c].z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[c].w, a.w, acc, 0, 0, 0);
            }
            {   // half chunk k = 48..55 (FRAG_HALF order: lane group g holds k = 48 + 2g + {0,1})
                const f32x2 a = *(const f32x2*)(xr + 48 - 2 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[3].x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[3].y, a.y, acc, 0, 0, 0);
            }
            const int col0 = nt_p * 16 + 4 * g;
            if (SPX)                  // all 64 columns of the planes (56..63 get 0: zero-padded W0 / b0)
                store_split4(XPL, PB, rt * 16 + r16, col0, make_float4(acc[0] + b04.x, acc[1] + b04.y, acc[2] + b04.z, acc[3] + b04.w));
            else if (col0 < XS)       // columns 56..59 get 0 (zero-padded W0 / b0)
                *(float4*)(X + (rt * 16 + r16) * XS + col0) =
                    make_float4(acc[0] + b04.x, acc[1] + b04.y, acc[2] + b04.z, acc[3] + b04.w);
        }
        __syncthreads();
    }

    AZG_PH(1);
    // ---- x tile requested first (HBM), then the weight fragments (L2) in the order the phases need them: the vmcnt
    //      counter retires in order, so P0 / P1 only wait for what they use while the rest keeps streaming in ----
    float4 xv[3];
    if (MODE != 1 && !INLDS) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + 768 * k, row = i / (XS / 4), c4 = i - row * (XS / 4);
            xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < ROWS * (XS / 4) && row < nrows && c4 < C / 4) xv[k] = *(const float4*)(xin + ((size_t)b0 * 7 + row) * C + 4 * c4);
        }
    }
    if (MODE != 1) load_we();
    const float4 be4 = *(const float4*)(W.be + nt_e * 16 + 4 * g);
    const float wd_stage = tid < 49 ? W.Wd[tid] : 0.f;       // requested now, parked in LDS after the expand phase (used by P2)
    if (MODE != 1 && !INLDS) {
        // ---- P0: x tile -> LDS (contiguous rows, float4); pad columns 56..59 zeroed ----
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + 768 * k, row = i / (XS / 4), c4 = i - row * (XS / 4);
            if (i < ROWS * (XS / 4)) *(float4*)(X + row * XS + 4 * c4) = xv[k];
        }
    }
    if (MODE != 1) load_w2_wp();
    const float4 b14 = *(const float4*)(W.b1 + nt_1 * 16 + 4 * g);
    const float4 b24 = *(const float4*)(W.b2 + nt_e * 16 + 4 * g);
    const float4 bp4 = *(const float4*)(W.bp + nt_p * 16 + 4 * g);
    __syncthreads();
    AZG_PH(2);

    // ---- P1: expand + BN + act -> H ----
    if (wave < 11) {
#pragma unroll 1
        for (int rt = 0; rt < 7; rt++) {
            const float* xr = X + (rt * 16 + r16) * XS + 4 * g;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (SPX) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint8_t* src = XPL + pl_off(rt * 16 + r16, 4 * c + g);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)src), am = __builtin_bit_cast(bf16x8, *(const uint4*)(src + PB)),
                                 al = __builtin_bit_cast(bf16x8, *(const uint4*)(src + 2 * PB));
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, weS[3 * c]), wm = __builtin_bit_cast(bf16x8, weS[3 * c + 1]),
                                 wl = __builtin_bit_cast(bf16x8, weS[3 * c + 2]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float4 a = *(const float4*)(xr + 16 * c);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[c].x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[c].y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[c].z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[c].w, a.w, acc, 0, 0, 0);
            }
            {   // half chunk k = 48..55
                const f32x2 a = *(const f32x2*)(xr + 48 - 2 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[3].x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[3].y, a.y, acc, 0, 0, 0);
            }
            }
            const int col0 = nt_e * 16 + 4 * g;
            if (col0 < E) {
                const f32x2 lo = act_apply2(f32x2{acc[0] + be4.x, acc[1] + be4.y}, ACT);
                const f32x2 hi = act_apply2(f32x2{acc[2] + be4.z, acc[3] + be4.w}, ACT);
                *(float4*)(H + (rt * 16 + r16) * HS + col0) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
    if (tid < 49) WD[tid] = wd_stage;
    __syncthreads();
    AZG_PH(3);
    if (wave < 3) {                    // SE fc1 weights: requested now (the expand fragments are dead), they land during P2
#pragma unroll
        for (int c = 0; c < 11; c++) w1r[c] = FRAG(W.W1, 11, nt_1, c);
    }

    // ---- P2: depthwise Linear(7->7) over the token axis + BN + act (in place) + SE squeeze ----
    float wd[49];                      // wave-uniform 7x7 weights -> SGPRs (one LDS broadcast read each, once per wave)
#pragma unroll
    for (int k = 0; k < 49; k++) wd[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, WD[k])));
    // two adjacent channels per thread as one f32x2: the 49 multiply-adds of the token mix run as v_pk_fma_f32
    for (int i = tid; i < NS * (E / 2); i += 768) {
        const int s = i / (E / 2), c = 2 * (i - s * (E / 2));
        float* base = H + (s * 7) * HS + c;
        f32x2 in[7];
#pragma unroll
        for (int l = 0; l < 7; l++) in[l] = *(const f32x2*)(base + l * HS);
        const f32x2 scl = *(const f32x2*)(W.sd + c), bb = *(const f32x2*)(W.bd + c);
        f32x2 pool = POOLMAX ? f32x2{-INFINITY, -INFINITY} : f32x2{0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 7; m++) {
            f32x2 a = f32x2{0.f, 0.f};
#pragma unroll
            for (int l = 0; l < 7; l++) a += wd[m * 7 + l] * in[l];
            a = act_apply2(a * scl + bb, ACT);
            *(f32x2*)(base + m * HS) = a;
            if (POOLMAX) { pool.x = fmaxf(pool.x, a.x); pool.y = fmaxf(pool.y, a.y); } else pool += a;
        }
        if (!POOLMAX) pool = pool * (1.f / 7.f);
        *(f32x2*)(PL + s * HS + c) = pool;
    }
    __syncthreads();
    AZG_PH(4);

    // ---- P3: SE fc1 + ReLU : SH[16][48] = relu(PL[16][168] @ W1 + b1) ----
    if (wave < 3) {
        const float* pr = PL + r16 * HS + 4 * g;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 10; c++) {
            const float4 a = *(const float4*)(pr + 16 * c);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[c].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[c].y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[c].z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[c].w, a.w, acc, 0, 0, 0);
        }
        {   // half chunk k = 160..167
            const f32x2 a = *(const f32x2*)(pr + 160 - 2 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[10].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[10].y, a.y, acc, 0, 0, 0);
        }
        float4 v;
        v.x = fmaxf(acc[0] + b14.x, 0.f); v.y = fmaxf(acc[1] + b14.y, 0.f);
        v.z = fmaxf(acc[2] + b14.z, 0.f); v.w = fmaxf(acc[3] + b14.w, 0.f);
        *(float4*)(SH + r16 * QS + nt_1 * 16 + 4 * g) = v;
    }
    __syncthreads();
    AZG_PH(5);

    // ---- P4: SE fc2 + Hardsigmoid : SC[16][168] ----
    if (wave < 11) {
        const float* hr = SH + r16 * QS + 4 * g;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float4 a = *(const float4*)(hr + 16 * c);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[c].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[c].y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[c].z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[c].w, a.w, acc, 0, 0, 0);
        }
        const int col0 = nt_e * 16 + 4 * g;
        if (col0 < E) {
            float4 v;
            v.x = hardsigmoid(acc[0] + b24.x); v.y = hardsigmoid(acc[1] + b24.y);
            v.z = hardsigmoid(acc[2] + b24.z); v.w = hardsigmoid(acc[3] + b24.w);
            *(float4*)(SC + r16 * HS + col0) = v;
        }
    }
    __syncthreads();
    AZG_PH(6);

    // policy head: the first Linear's weight fragments stream in from L2 while P5 runs (we / w1r / w2r are dead by now)
    float4 wr[14], w2h[6];
    const int ht_nt = wave % 6, ht_half = wave / 6;
    const int ht_beg = ht_half * 14, ht_n = ht_half ? 13 : 14;          // 27 K chunks of 16 (432 >= 420)
    if (MODE == 2) {
#pragma unroll
        for (int cc = 0; cc < 14; cc++) wr[cc] = cc < ht_n ? FRAG(N.Wh1, 27, ht_nt, ht_beg + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave < 6) {
#pragma unroll
            for (int c = 0; c < 6; c++) w2h[c] = FRAG(N.Wh2, 6, wave, c);
        }
    }

    // ---- P5: project (SE-scaled operand) + BN + residual -> HBM (MODE 0/1) or in place of x in LDS (MODE 2/3) ----
#pragma unroll 1
    for (int rt = wave >> 2; rt < 7; rt += 3) {
        const int row = rt * 16 + r16;
        const float* hr = H + row * HS + 4 * g;
        const float* sr = SC + (row / 7) * HS + 4 * g;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 10; c++) {
            float4 a = *(const float4*)(hr + 16 * c);
            const float4 s4 = *(const float4*)(sr + 16 * c);
            a.x *= s4.x; a.y *= s4.y; a.z *= s4.z; a.w *= s4.w;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[c].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[c].y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[c].z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[c].w, a.w, acc, 0, 0, 0);
        }
        {   // half chunk k = 160..167
            const f32x2 a = *(const f32x2*)(hr + 160 - 2 * g) * *(const f32x2*)(sr + 160 - 2 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[10].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[10].y, a.y, acc, 0, 0, 0);
        }
        const int col0 = nt_p * 16 + 4 * g;
        if (SPX) {
            const float4 xr = load_split4(XPL, PB, row, col0);       // (columns 56..63 of the planes are zero)
            const float4 o4 = make_float4(acc[0] + bp4.x + xr.x, acc[1] + bp4.y + xr.y, acc[2] + bp4.z + xr.z, acc[3] + bp4.w + xr.w);
            if (MODE == 1) store_split4(XPL, PB, row, col0, o4);     // trunk: in place, a lane rewrites the elements it read
            else if (col0 < XS) *(float4*)(OT + row * XS + col0) = o4;
        } else if (MODE >= 2 || OUTLDS) {
            if (col0 < XS) {          // each element is read (residual) and overwritten by the same lane
                float* xp = X + row * XS + col0;
                const float4 xr = *(const float4*)xp;
                const float4 o4 = make_float4(acc[0] + bp4.x + xr.x, acc[1] + bp4.y + xr.y, acc[2] + bp4.z + xr.z,
                                              acc[3] + bp4.w + xr.w);
                *(float4*)xp = o4;
                if (SAVE) *(float4*)(xsave + row * XS + col0) = o4;
            }
        } else if (row < nrows && col0 < C) {
            const float4 xr = *(const float4*)(X + row * XS + col0);
            float4 v;
            v.x = acc[0] + bp4.x + xr.x; v.y = acc[1] + bp4.y + xr.y;
            v.z = acc[2] + bp4.z + xr.z; v.w = acc[3] + bp4.w + xr.w;
            *(float4*)(xout + ((size_t)b0 * 7 + row) * C + col0) = v;
        }
    }

    AZG_PH(7);
    if (MODE == 2) {
        // ---- policy head tail on the block output O = X viewed as [16][420] ----
        constexpr int LS = 100;
        float* RED = H;                      // [2][16][LS]   K-halves of the first Linear
        float* HID = RED + 2 * 16 * LS;      // [16][LS]
        float* LG = HID + 16 * LS;           // [16][LS]
        const int nt = ht_nt, half = ht_half, c_beg = ht_beg, n_c = ht_n;
        __syncthreads();                     // O complete, H free
        AZG_PH(8);
        {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < 14; cc++) {
                const int k0 = 16 * (c_beg + cc) + 4 * g;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cc < n_c && k0 < FK) a = *(const float4*)((SPX ? OT : X) + r16 * FK + k0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[cc].x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[cc].y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[cc].z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[cc].w, a.w, acc, 0, 0, 0);
            }
            *(float4*)(RED + (half * 16 + r16) * LS + nt * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        if (tid < 16 * 24) {
            const int s = tid / 24, col = 4 * (tid - s * 24);
            const float4 p0 = *(const float4*)(RED + s * LS + col), p1 = *(const float4*)(RED + (16 + s) * LS + col);
            const float4 bb = *(const float4*)(N.bh1 + col);
            *(float4*)(HID + s * LS + col) = make_float4(fmaxf(p0.x + p1.x + bb.x, 0.f), fmaxf(p0.y + p1.y + bb.y, 0.f),
                                                         fmaxf(p0.z + p1.z + bb.z, 0.f), fmaxf(p0.w + p1.w + bb.w, 0.f));
        }
        __syncthreads();
        if (wave < 6) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float4 a = *(const float4*)(HID + r16 * LS + 16 * c + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2h[c].x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2h[c].y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2h[c].z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2h[c].w, a.w, acc, 0, 0, 0);
            }
            const float4 bb = *(const float4*)(N.bh2 + wave * 16 + 4 * g);
            *(float4*)(LG + r16 * LS + wave * 16 + 4 * g) =
                make_float4(acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
        }
        __syncthreads();
        AZG_PH(9);
        // masked softmax == exp(log_softmax(where(valid, logits, -1e8))) (GenericNNetWrapper.py:105-107), one wave per sample
        for (int s = wave; s < NS; s += 12) {
            const int b = b0 + s;
            if (b >= B) continue;
            const int a1 = lane + 64;
            float x0 = valid[(size_t)b * A + lane] ? LG[s * LS + lane] : -1e8f;
            float x1 = a1 < A ? (valid[(size_t)b * A + a1] ? LG[s * LS + a1] : -1e8f) : -INFINITY;
            const float mx = nn_wave_max(fmaxf(x0, x1));
            x0 = expf(x0 - mx);
            x1 = a1 < A ? expf(x1 - mx) : 0.f;
            const float sum = nn_wave_sum(x0 + x1);
            pi_out[(size_t)b * A + lane] = x0 / sum;
            if (a1 < A) pi_out[(size_t)b * A + a1] = x1 / sum;
        }
        AZG_PH(10);
    }

    if (MODE == 3) {
        // ---- value head tail: Linear(392,P) -> ReLU -> Linear(P,P) -> tanh ----
        float* RED = H;                      // [12][16][16]
        __syncthreads();
        {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int c = wave + 12 * cc;
                const int k0 = 16 * c + 4 * g;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < 27) {
                    w4 = FRAG(N.Wh1, 27, 0, c);
                    if (k0 < FK) a = *(const float4*)((SPX ? OT : X) + r16 * FK + k0);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, a.w, acc, 0, 0, 0);
            }
            *(float4*)(RED + (wave * 16 + r16) * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        if (tid < NS * P) {
            const int s = tid / P, p = tid - s * P, b = b0 + s;
            if (b < B) {
                float a = N.bh2[p];
                for (int j = 0; j < P; j++) {
                    float h = N.bh1[j];
                    for (int w = 0; w < 12; w++) h += RED[(w * 16 + s) * 16 + j];
                    a += fmaxf(h, 0.f) * N.Wh2[j * P + p];
                }
                v_out[(size_t)b * P + p] = tanhf(a);
            }
        }
    }
}

template <int ACT, int POOLMAX, int MODE>
AZG_NN_KERNEL __global__ __launch_bounds__(768) void k_v80_block(const float* __restrict__ xin, float* __restrict__ xout, V80BlockW W,
                                                   int B, const int8_t* __restrict__ boards, V80NetW N,
                                                   const uint8_t* __restrict__ valid, float* __restrict__ pi_out,
                                                   float* __restrict__ v_out, int P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    v80_block_body<ACT, POOLMAX, MODE, false, false, false>(smem, smem + 112 * 60, nullptr, xin, xout, W, B, boards, N, valid,
                                                           pi_out, v_out, P);
}

// The whole V80 forward of one 16-sample tile in ONE workgroup pass: first layer + trunk block (its output stays in LDS and
// is copied to a second tile buffer), policy head block + tail on the first copy, value head block + tail on the second.
// Nothing but the int8 boards, the valid masks and pi / v crosses HBM; LDS = 2 x [112][60] + the block's 129.5 KB = 156.4 KB.
AZG_NN_KERNEL __global__ __launch_bounds__(768) void k_v80_net(V80BlockW Wt, V80BlockW Wp, V80BlockW Wv, V80NetW N0, V80NetW Np, V80NetW Nv,
                                                 const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid, int B,
                                                 int P, float* __restrict__ pi_out, float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* XT = X + 112 * 60;
    float* H = XT + 112 * 60;
    v80_block_body<1, 0, 1, false, true, true>(X, H, XT, nullptr, nullptr, Wt, B, boards, N0, nullptr, nullptr, nullptr, P);
    __syncthreads();
    v80_block_body<2, 1, 2, true, false, false>(X, H, nullptr, nullptr, nullptr, Wp, B, nullptr, Np, valid, pi_out, nullptr, P);
    __syncthreads();
    v80_block_body<2, 1, 3, true, false, false>(XT, H, nullptr, nullptr, nullptr, Wv, B, nullptr, Nv, nullptr, nullptr, v_out, P);
}

// The same forward with the tile the three blocks read kept as bf16 planes and the expand GEMMs on bf16 x 3 operands (SPX above):
// LDS = 3 x [112][64] bf16 + the block's buffers + 12 KB = 154.3 KB; one copy of the trunk output serves both heads.
AZG_NN_KERNEL __global__ __launch_bounds__(768) void k_v80_net_spx(V80BlockW Wt, V80BlockW Wp, V80BlockW Wv, V80NetW N0, V80NetW Np, V80NetW Nv,
                                                     const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid, int B,
                                                     int P, float* __restrict__ pi_out, float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                  // three bf16 planes [112][64]
    float* H = X + 3 * 112 * 128 / 4;
    v80_block_body<1, 0, 1, false, true, false, true>(X, H, nullptr, nullptr, nullptr, Wt, B, boards, N0, nullptr, nullptr, nullptr, P);
    __syncthreads();
    v80_block_body<2, 1, 2, true, false, false, true>(X, H, nullptr, nullptr, nullptr, Wp, B, nullptr, Np, valid, pi_out, nullptr, P);
    __syncthreads();
    v80_block_body<2, 1, 3, true, false, false, true>(X, H, nullptr, nullptr, nullptr, Wv, B, nullptr, Nv, nullptr, nullptr, v_out, P);
}

// boards int8 [B][C][L] (reference layout) -> x f32 [B][L][ldx] (channels-last; columns C..ldx-1 zeroed)
AZG_NN_KERNEL __global__ __launch_bounds__(256) void k_board_to_x(const int8_t* __restrict__ boards, float* __restrict__ x, int B, int C,
                                                    int L, int ldx) {
    const long long total = (long long)B * L * ldx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / (L * ldx)), rem = (int)(i - (long long)b * L * ldx), l = rem / ldx, c = rem - l * ldx;
        x[i] = c < C ? (float)boards[(size_t)b * L * C + c * L + l] : 0.f;
    }
}

// pi[b] = softmax(where(valid, logits, -1e8)) ; v[b] = tanh(relu(vhid[b]) @ Wv2 + bv2)      one wave per sample
AZG_NN_KERNEL __global__ __launch_bounds__(64) void k_heads_out(const float* __restrict__ logits, int ldl,
                                                  const uint8_t* __restrict__ valid, const float* __restrict__ vhid,
                                                  int ldv, const float* __restrict__ Wv2, const float* __restrict__ bv2,
                                                  float* __restrict__ pi, float* __restrict__ v, int B, int A, int P) {
    const int b = blockIdx.x, l = threadIdx.x;
    if (b >= B) return;
    float x[4];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int a = l + 64 * k;
        x[k] = -INFINITY;
        if (a < A) { x[k] = valid[(size_t)b * A + a] ? logits[(size_t)b * ldl + a] : -1e8f; mx = fmaxf(mx, x[k]); }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = (l + 64 * k < A) ? expf(x[k] - mx) : 0.f; s += x[k]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
#pragma unroll
    for (int k = 0; k < 4; k++) if (l + 64 * k < A) pi[(size_t)b * A + l + 64 * k] = x[k] / s;
    if (l < P) {
        float a = bv2[l];
        for (int j = 0; j < P; j++) { float hj = vhid[(size_t)b * ldv + j]; a += (hj > 0.f ? hj : 0.f) * Wv2[j * P + l]; }
        v[(size_t)b * P + l] = tanhf(a);
    }
}

}  // namespace azg

#pragma clang fp contract(off)
