// azg_nn.hip -- C-ABI (include/azg.h) of the policy/value net kernels: host-side launch code (NeuralNet.predict for a whole
// leaf batch).  Second translation unit of libazg_hip.so; compiled with the default flags (the forest unit, azg.hip, forbids
// scalarised global loads, see build.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

// this unit also holds the per-CU round kernel (azg_fused.hip.h, included at the end): its workgroups run 16 independent tree waves, so
// wave_sync() must be a wavefront fence here (the net kernels synchronise with __syncthreads() directly and are not affected)
#define AZG_WAVE_LOCAL_SYNC 1
#include "../../include/azg.h"
#include "azg_host.h"
#include "azg_common.hip.h"
#include "nn_kernels.hip.h"
#include "nn_v80_h2.hip.h"
#include "nn_mb1d.hip.h"
#include "nn_conv5x5.hip.h"

using namespace azg;

// ---- policy/value net building blocks (NeuralNet.predict, GenericNNetWrapper.py:94-120; V80: SplendorNNet.py:262-283) ----
template <int NT, int KSPLIT, int NCH>
static int launch_linear(const float* A, int lda, const float* Wp, const float* bias, const float* R, int ldr,
                         const float* rowscale, int rpg, float* out, int ldc, int M, int K, int Kp, int N, int act,
                         hipStream_t s) {
    constexpr int NP = NT * 16;
    const int tiles = (M + 15) / 16;
    size_t lds;
    int grid;
    if (KSPLIT) {
        lds = (size_t)4 * NT * 4 * 64 * sizeof(float);
        grid = tiles;
    } else {
        lds = (size_t)Kp * NP * sizeof(float);
        grid = (tiles + 3) / 4;
        if (grid > 512) grid = 512;
    }
    if (lds > 160 * 1024) return fail("azg_nn_linear: weight tile exceeds LDS");
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        HIPCHK(hipFuncSetAttribute((const void*)k_linear<NT, KSPLIT, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    k_linear<NT, KSPLIT, NCH><<<dim3(grid), dim3(256), lds, s>>>(A, lda, Wp, bias, R, ldr, rowscale, rpg, out, ldc, M, K, Kp,
                                                                  N, act);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_linear(const float* A, int lda, const float* Wp, int Kp, int NP, const float* bias, const float* R,
                             int ldr, const float* rowscale, int rows_per_group, float* out, int ldc, int M, int K, int N,
                             int act, int ksplit, void* stream) {
    if (!A || !Wp || !out || M <= 0) return fail("azg_nn_linear: null/empty argument");
    if (K % 4 || lda % 4 || Kp % 16 || Kp < K || NP % 16 || NP < N || (rowscale && rows_per_group <= 0))
        return fail("azg_nn_linear: K, lda multiples of 4; Wp padded to [Kp % 16 == 0][NP % 16 == 0]");
    hipStream_t s = (hipStream_t)stream;
    const int nt = NP / 16;
    const int chunks = Kp / 16;
    const int per_wave = ksplit ? (chunks + 3) / 4 : chunks;
    const int nch = per_wave <= 2 ? 2 : (per_wave <= 4 ? 4 : (per_wave <= 7 ? 7 : (per_wave <= 11 ? 11 : 0)));
    if (!nch) return fail("azg_nn_linear: K too long for this variant (K <= 176, or <= 704 with ksplit)");
#define ARGS A, lda, Wp, bias, R, ldr, rowscale, rows_per_group, out, ldc, M, K, Kp, N, act, s
#define LIN_N(NTV, KS)                                                                     \
    switch (nch) {                                                                         \
        case 2: return launch_linear<NTV, KS, 2>(ARGS);                                    \
        case 4: return launch_linear<NTV, KS, 4>(ARGS);                                    \
        case 7: return launch_linear<NTV, KS, 7>(ARGS);                                    \
        default: return launch_linear<NTV, KS, 11>(ARGS);                                  \
    }
#define LIN(NTV) do { if (ksplit) { LIN_N(NTV, 1) } else { LIN_N(NTV, 0) } } while (0)
    switch (nt) {
        case 1: LIN(1);
        case 4: LIN(4);
        case 6: LIN(6);
        case 11: LIN(11);
        case 12: LIN(12);
        default: return fail("azg_nn_linear: NP/16 must be 1, 4, 6, 11 or 12");
    }
#undef LIN
#undef LIN_N
#undef ARGS
}

template <int NCH>
static int launch_linear_ws(const float* A, int lda, const float* Wp, int NP, const float* biasp, const float* R, int ldr,
                            const float* rowscale, int rpg, float* out, int ldc, int M, int K, int N, int act,
                            hipStream_t s) {
    const int tiles = (M + 15) / 16;
    const int col_groups = ((N + 15) / 16 + 3) / 4;
    // enough workgroups to fill the chip (~2048 waves), but several row tiles per wave so the weight fragment and the
    // software prefetch are amortised
    int tiles_per_wg = (tiles * col_groups + 2047) / 2048;
    if (tiles_per_wg < 1) tiles_per_wg = 1;
    const int gx = (tiles + tiles_per_wg - 1) / tiles_per_wg;
    k_linear_ws<NCH><<<dim3(gx, col_groups), dim3(256), 0, s>>>(A, lda, Wp, NP, biasp, R, ldr, rowscale, rpg, out, ldc, M, K, N,
                                                                act, tiles_per_wg);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_linear_ws(const float* A, int lda, const float* Wp, int Kp, int NP, const float* bias_padded,
                                const float* R, int ldr, const float* rowscale, int rows_per_group, float* out, int ldc,
                                int M, int K, int N, int act, void* stream) {
    if (!A || !Wp || !out || M <= 0) return fail("azg_nn_linear_ws: null/empty argument");
    if (K % 4 || lda % 4 || ldc % 4 || (R && ldr % 4) || Kp % 16 || Kp < K || NP % 16 || NP < N ||
        (rowscale && rows_per_group <= 0))
        return fail("azg_nn_linear_ws: K, lda, ldc, ldr multiples of 4; Wp padded to [Kp % 16 == 0][NP % 16 == 0]");
    hipStream_t s = (hipStream_t)stream;
#define WS(NCHV) return launch_linear_ws<NCHV>(A, lda, Wp, NP, bias_padded, R, ldr, rowscale, rows_per_group, out, ldc, M, K, N, act, s)
    switch (Kp / 16) {
        case 1: WS(1);
        case 2: WS(2);
        case 3: WS(3);
        case 4: WS(4);
        case 6: WS(6);
        case 8: WS(8);
        case 11: WS(11);
        case 17: WS(17);
        case 25: WS(25);
        default: return fail("azg_nn_linear_ws: Kp/16 must be 1, 2, 3, 4, 6, 8, 11, 17 or 25");
    }
#undef WS
}

extern "C" int azg_nn_dw_pool_l(float* H, int ldh, const float* Wd, const float* sd, const float* bd, float* pooled, int B,
                                int E, int L, int act, int pool_max, void* stream) {
    if (!H || !Wd || !pooled || B <= 0) return fail("azg_nn_dw_pool: null/empty argument");
    const long long total = (long long)B * E;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (L == 7) k_dw_pool<7><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(H, ldh, Wd, sd, bd, pooled, B, E, act, pool_max);
    else if (L == 6) k_dw_pool<6><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(H, ldh, Wd, sd, bd, pooled, B, E, act, pool_max);
    else if (L == 2) k_dw_pool<2><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(H, ldh, Wd, sd, bd, pooled, B, E, act, pool_max);
    else if (L == 15) k_dw_pool<15><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(H, ldh, Wd, sd, bd, pooled, B, E, act, pool_max);
    else return fail("azg_nn_dw_pool: token count must be 2, 6, 7 or 15 (Minivilles, Azul, Splendor, The Little Prince)");
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_dw_pool(float* H, int ldh, const float* Wd, const float* sd, const float* bd, float* pooled, int B,
                              int E, int act, int pool_max, void* stream) {
    return azg_nn_dw_pool_l(H, ldh, Wd, sd, bd, pooled, B, E, 7, act, pool_max, stream);
}

static constexpr size_t V80_LDS = (size_t)(112 * 60 + 112 * 172 + 2 * 16 * 172 + 16 * 52 + 64) * sizeof(float);

template <int A_, int P_, int M_>
static int launch_v80(const float* xin, float* xout, const V80BlockW& W, int B, const int8_t* boards, const V80NetW& N,
                      const uint8_t* valid, float* pi, float* v, int P, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)k_v80_block<A_, P_, M_>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr = true;
    }
    k_v80_block<A_, P_, M_><<<dim3((B + 15) / 16), dim3(768), V80_LDS, s>>>(xin, xout, W, B, boards, N, valid, pi, v, P);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_v80_block(const float* xin, float* xout, const float* const* w /* 11 device pointers */, int B,
                                int act, int pool_max, void* stream) {
    if (!xin || !xout || !w || B <= 0) return fail("azg_nn_v80_block: null/empty argument");
    V80BlockW W{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10]};
    V80NetW N{};
    hipStream_t s = (hipStream_t)stream;
    if (act == 1 && !pool_max) return launch_v80<1, 0, 0>(xin, xout, W, B, nullptr, N, nullptr, nullptr, nullptr, 0, s);
    if (act == 1 && pool_max) return launch_v80<1, 1, 0>(xin, xout, W, B, nullptr, N, nullptr, nullptr, nullptr, 0, s);
    if (act == 2 && !pool_max) return launch_v80<2, 0, 0>(xin, xout, W, B, nullptr, N, nullptr, nullptr, nullptr, 0, s);
    if (act == 2 && pool_max) return launch_v80<2, 1, 0>(xin, xout, W, B, nullptr, N, nullptr, nullptr, nullptr, 0, s);
    return fail("azg_nn_v80_block: act must be 1 (ReLU) or 2 (Hardswish)");
}

static int v80_forward(const int8_t* boards, const uint8_t* valid, const float* const* w /* 43 */, int B,
                       int P, float* x_trunk, float* pi, float* v, void* stream, bool split) {
    if (!boards || !valid || !w || !x_trunk || !pi || !v || B <= 0) return fail("azg_nn_v80_forward: null/empty argument");
    if (P < 2 || P > 4) return fail("azg_nn_v80_forward: 2 <= P <= 4");
    V80BlockW Wt{w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12]};
    V80BlockW Wp{w[13], w[14], w[15], w[16], w[17], w[18], w[19], w[20], w[21], w[22], w[23]};
    V80BlockW Wv{w[24], w[25], w[26], w[27], w[28], w[29], w[30], w[31], w[32], w[33], w[34]};
    V80NetW N0{w[0], w[1], nullptr, nullptr, nullptr, nullptr};
    V80NetW Np{nullptr, nullptr, w[35], w[36], w[37], w[38]};
    V80NetW Nv{nullptr, nullptr, w[39], w[40], w[41], w[42]};
    hipStream_t s = (hipStream_t)stream;
    // V80 geometry (SplendorNNet.py:262-283): trunk ReLU + mean-SE, both heads Hardswish + max-SE
    if (split) {
        static bool attr_s = false;
        constexpr size_t lds_s = (size_t)3 * 112 * 128 + (V80_LDS - (size_t)112 * 60 * sizeof(float)) + 12288;
        if (!attr_s) {
            HIPCHK(hipFuncSetAttribute((const void*)k_v80_net_spx, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_s = true;
        }
        k_v80_net_spx<<<dim3((B + 15) / 16), dim3(768), lds_s, s>>>(Wt, Wp, Wv, N0, Np, Nv, boards, valid, B, P, pi, v);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (getenv("AZG_NN_THREE_LAUNCHES")) {       // the per-block path (kept for A/B measurements)
        if (launch_v80<1, 0, 1>(nullptr, x_trunk, Wt, B, boards, N0, nullptr, nullptr, nullptr, P, s)) return -1;
        if (launch_v80<2, 1, 2>(x_trunk, nullptr, Wp, B, nullptr, Np, valid, pi, nullptr, P, s)) return -1;
        if (launch_v80<2, 1, 3>(x_trunk, nullptr, Wv, B, nullptr, Nv, nullptr, nullptr, v, P, s)) return -1;
        return 0;
    }
    static bool attr = false;
    constexpr size_t lds = V80_LDS + (size_t)112 * 60 * sizeof(float);
    if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)k_v80_net, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    k_v80_net<<<dim3((B + 15) / 16), dim3(768), lds, s>>>(Wt, Wp, Wv, N0, Np, Nv, boards, valid, B, P, pi, v);
    HIPCHK(hipGetLastError());
    return 0;
}


extern "C" int azg_nn_v80_forward(const int8_t* boards, const uint8_t* valid, const float* const* w /* 43 */, int B,
                                  int P, float* x_trunk, float* pi, float* v, void* stream) {
    return v80_forward(boards, valid, w, B, P, x_trunk, pi, v, stream, false);
}

extern "C" int azg_nn_v80_forward_split(const int8_t* boards, const uint8_t* valid, const float* const* w /* 43 */, int B,
                                        int P, float* x_trunk, float* pi, float* v, void* stream) {
    return v80_forward(boards, valid, w, B, P, x_trunk, pi, v, stream, true);
}

// The V80 forward on fp16 hi+lo split operands with token-major tiles (nn_v80_h2.hip.h): one launch, one workgroup per 16 samples
extern "C" int azg_nn_v80_forward_h2(const int8_t* boards, const uint8_t* valid, const void* const* w /* 43 */,
                                     const float* descale /* 16, host */, int B, int P, float* pi, float* v, void* stream) {
    if (!boards || !valid || !w || !descale || !pi || !v || B <= 0) return fail("azg_nn_v80_forward_h2: null/empty argument");
    if (P < 2 || P > 4) return fail("azg_nn_v80_forward_h2: 2 <= P <= 4");
    const H2Weights HW = h2_weights(w, descale);
    hipStream_t s = (hipStream_t)stream;
    static int waves_h2 = 0;          // AZG_V80_WAVES=16: the 16-wave / 128-VGPR variant (the form the fused round kernel uses); default 12
    if (!waves_h2) {
        HIPCHK(hipFuncSetAttribute((const void*)k_v80_net_h2<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_v80_net_h2<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        waves_h2 = (getenv("AZG_V80_WAVES") && atoi(getenv("AZG_V80_WAVES")) == 16) ? 16 : 12;
    }
    if (waves_h2 == 16) k_v80_net_h2<16><<<dim3((B + 15) / 16), dim3(1024), H2_LDS, s>>>(HW, boards, valid, B, P, pi, v);
    else k_v80_net_h2<12><<<dim3((B + 15) / 16), dim3(768), H2_LDS, s>>>(HW, boards, valid, B, P, pi, v);
    HIPCHK(hipGetLastError());
    return 0;
}

#ifdef AZG_NN_PHASE_TIMES
extern "C" int azg_nn_debug_phase_times(long long* out /* [4][16] */) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v80_phase), sizeof(long long) * 64));
    return 0;
}
extern "C" int azg_nn_debug_phase_times_c5(long long* out /* [32] */) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c5_phase), sizeof(long long) * 32));
    HIPCHK(hipMemcpyFromSymbol(out + 28, HIP_SYMBOL(g_c5_first), sizeof(long long) * 3));      // stamps inside the first convolution
    return 0;
}
extern "C" int azg_nn_debug_phase_times_h2(long long* out /* [4][16] */) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2_phase), sizeof(long long) * 64));
    return 0;
}
#endif

// ---- whole MobileNet-1d forward, any supported geometry, one launch (nn_mb1d.hip.h) ----
// (the geometries CfgSplendor2 / 3 / 4, CfgAzul, CfgMinivilles2, CfgTLP3: nn_mb1d.hip.h)

template <class CF, bool H2>
static int launch_mb1d(const Mb1dNetW& N, const int8_t* boards, const uint8_t* valid, int B, float* pi, float* v, hipStream_t s) {
    constexpr size_t lds = (size_t)CF::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 160 * 1024, "geometry does not fit the LDS of a CU");
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)k_mb1d_net<CF, H2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    k_mb1d_net<CF, H2><<<dim3((B + CF::NS - 1) / CF::NS), dim3(768), lds, s>>>(N, boards, valid, B, pi, v);
    HIPCHK(hipGetLastError());
    return 0;
}

template <bool H2>
static int mb1d_forward(int geometry, const int8_t* boards, const uint8_t* valid, const float* const* w, const float* descale,
                        int B, float* pi, float* v, void* stream) {
    if (!boards || !valid || !w || !pi || !v || B <= 0 || (H2 && !descale)) return fail("azg_nn_mb1d_forward: null/empty argument");
    Mb1dNetW N;
    for (int i = 0; i < 16; i++) N.ds[i] = H2 ? descale[i] : 1.f;
    N.W0 = w[0]; N.b0 = w[1];
    for (int b = 0; b < 3; b++) {
        const float* const* q = w + 2 + 11 * b;
        N.blk[b] = Mb1dBlockW{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10]};
    }
    const float* const* h = w + 35;
    N.Wpi1 = h[0]; N.bpi1 = h[1]; N.Wpi2 = h[2]; N.bpi2 = h[3]; N.Wv1 = h[4]; N.bv1 = h[5]; N.Wv2 = h[6]; N.bv2 = h[7];
    hipStream_t s = (hipStream_t)stream;
    switch (geometry) {
        case AZG_NET_SPLENDOR2: return launch_mb1d<CfgSplendor2, H2>(N, boards, valid, B, pi, v, s);
        case AZG_NET_SPLENDOR3: return launch_mb1d<CfgSplendor3, H2>(N, boards, valid, B, pi, v, s);
        case AZG_NET_SPLENDOR4: return launch_mb1d<CfgSplendor4, H2>(N, boards, valid, B, pi, v, s);
        case AZG_NET_AZUL: return launch_mb1d<CfgAzul, H2>(N, boards, valid, B, pi, v, s);
        case AZG_NET_MINIVILLES2: return launch_mb1d<CfgMinivilles2, H2>(N, boards, valid, B, pi, v, s);
        case AZG_NET_TLP3: return launch_mb1d<CfgTLP3, H2>(N, boards, valid, B, pi, v, s);
        default: return fail("azg_nn_mb1d_forward: unknown geometry");
    }
}

extern "C" int azg_nn_mb1d_forward(int geometry, const int8_t* boards, const uint8_t* valid, const float* const* w, int B,
                                   float* pi, float* v, void* stream) {
    return mb1d_forward<false>(geometry, boards, valid, w, nullptr, B, pi, v, stream);
}

// the same forward with the GEMM phases on f16 x 2 split-precision operands: matrices as h2 fragments, descale[16] on the host
extern "C" int azg_nn_mb1d_forward_h2(int geometry, const int8_t* boards, const uint8_t* valid, const void* const* w,
                                      const float* descale, int B, float* pi, float* v, void* stream) {
    return mb1d_forward<true>(geometry, boards, valid, (const float* const*)w, descale, B, pi, v, stream);
}

// ---- Santorini ResNet V88/V89 (no-gods geometry: A = 162, P = 2, 5 residual blocks), one launch (nn_conv5x5.hip.h) ----
static int conv5_launch(const int8_t* boards, const uint8_t* valid, const float* const* w, int n_blocks, int A, int P, int B,
                        float* pi, float* v, void* stream, int split /* 0 f32, 3 bf16 x 3, 2 f16 x 2 */, float descale = 1.f) {
    if (!boards || !valid || !w || !pi || !v || B <= 0) return fail("azg_nn_conv5_forward: null/empty argument");
    if (n_blocks != 5 || A != 162 || P != 2) return fail("azg_nn_conv5_forward: built for 5 residual blocks, A = 162, P = 2");
    Conv5NetW N{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13]};
    static bool attr[3] = {false, false, false};
    if (split == 2) {
        // two activation tiles of two f16 planes (+ two zero rows each); the second tile also holds the f32 board staging tile at the
        // start and the f32 trunk output + head buffers at the end of the kernel (54.4 KB + 9.8 KB)
        // + the LDS copies of the head / FC matrices (k_conv5_net: WST_N floats)
        constexpr size_t lds = (size_t)C5_LDS_LEAD + (size_t)2 * 202 * 128 + 65536 + (size_t)(2 * 25 * 162 + 25 * 64 + 64 * 2 + 64) * sizeof(float);
        static_assert(lds <= 160 * 1024, "k_conv5_net<.., 2>: LDS");
        if (!attr[2]) {
            HIPCHK(hipFuncSetAttribute((const void*)k_conv5_net<5, 162, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[2] = true;
        }
        k_conv5_net<5, 162, 2, 2><<<dim3((B + 7) / 8), dim3(768), lds, (hipStream_t)stream>>>(N, boards, valid, B, pi, v, descale);
    } else if (split) {
        constexpr size_t lds = (size_t)2 * 3 * 202 * 128;              // two activation tiles of three bf16 planes (+ two zero rows each)
        if (!attr[1]) {
            HIPCHK(hipFuncSetAttribute((const void*)k_conv5_net<5, 162, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[1] = true;
        }
        k_conv5_net<5, 162, 2, 3><<<dim3((B + 7) / 8), dim3(768), lds, (hipStream_t)stream>>>(N, boards, valid, B, pi, v, 1.f);
    } else {
        constexpr size_t lds = (size_t)2 * 200 * 68 * sizeof(float);
        if (!attr[0]) {
            HIPCHK(hipFuncSetAttribute((const void*)k_conv5_net<5, 162, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[0] = true;
        }
        k_conv5_net<5, 162, 2, 0><<<dim3((B + 7) / 8), dim3(768), lds, (hipStream_t)stream>>>(N, boards, valid, B, pi, v, 1.f);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_conv5_forward(const int8_t* boards, const uint8_t* valid, const float* const* w, int n_blocks, int A,
                                    int P, int B, float* pi, float* v, void* stream) {
    return conv5_launch(boards, valid, w, n_blocks, A, P, B, pi, v, stream, 0);
}

extern "C" int azg_nn_conv5_forward_split(const int8_t* boards, const uint8_t* valid, const float* const* w, int n_blocks, int A,
                                          int P, int B, float* pi, float* v, void* stream) {
    return conv5_launch(boards, valid, w, n_blocks, A, P, B, pi, v, stream, 3);
}

extern "C" int azg_nn_conv5_forward_h2(const int8_t* boards, const uint8_t* valid, const float* const* w, float descale, int n_blocks,
                                       int A, int P, int B, float* pi, float* v, void* stream) {
    return conv5_launch(boards, valid, w, n_blocks, A, P, B, pi, v, stream, 2, descale);
}

// ---- Santorini-with-gods net V78 (10 InvertedResidual blocks, A = 1782, P = 2): trunk + value head, then the policy FC (nn_conv5x5.hip.h) ----
static int s78_launch(const int8_t* boards, const uint8_t* valid, const float* const* w, int n_blocks, int A, int P, int B, float* pi,
                      float* v, void* stream, int split /* 0 f32, 3 bf16 x 3, 2 f16 x 2 */, float ds_e = 1.f, float ds_p = 1.f) {
    if (!boards || !valid || !w || !pi || !v || B <= 0) return fail("azg_nn_s78_forward: null/empty argument");
    if (n_blocks != 10 || A != 1782 || P != 2) return fail("azg_nn_s78_forward: built for 10 blocks, A = 1782, P = 2");
    S78NetW N{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15], w[16], w[17], w[18]};
    if (split == 2) {
        constexpr size_t lds = (size_t)(2 + 4) * 202 * 128 + 8 * 32 * sizeof(float);      // X: two f16 planes; H: the f32 expanded tile + two f16 planes behind it
        static bool attr = false;
        if (!attr) {
            HIPCHK(hipFuncSetAttribute((const void*)k_s78_net_split<10, 1782, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        k_s78_net_split<10, 1782, 2, 2><<<dim3((B + 7) / 8), dim3(768), lds, (hipStream_t)stream>>>(N, boards, valid, B, pi, v, ds_e, ds_p);
    } else if (split) {
        constexpr size_t lds = (size_t)2 * 3 * 202 * 128 + 8 * 32 * sizeof(float);
        static bool attr = false;
        if (!attr) {
            HIPCHK(hipFuncSetAttribute((const void*)k_s78_net_split<10, 1782, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        k_s78_net_split<10, 1782, 2, 3><<<dim3((B + 7) / 8), dim3(768), lds, (hipStream_t)stream>>>(N, boards, valid, B, pi, v, 1.f, 1.f);
    } else {
        constexpr size_t lds = (size_t)(112 * 68 + 112 * 196 + 4 * 32) * sizeof(float);
        static bool attr = false;
        if (!attr) {
            HIPCHK(hipFuncSetAttribute((const void*)k_s78_net<10, 1782, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        k_s78_net<10, 1782, 2><<<dim3((B + 3) / 4), dim3(768), lds, (hipStream_t)stream>>>(N, boards, valid, B, pi, v);
    }
    HIPCHK(hipGetLastError());
    static const int pol2 = getenv("AZG_S78_POLICY2") ? atoi(getenv("AZG_S78_POLICY2")) : 1;    // (0: the one-launch form, k_s78_policy_h2)
    if (split == 2 && pol2) {               // policy FC in two launches: 64 samples x a quarter of the columns per workgroup, then the softmax
        // workspace for the raw logits [B][1792] (the pi rows hold the policy features until every column quarter has read them): kept per
        // device, grown on demand (the first call of a size happens in the warm-up rounds, before any graph capture)
        static float* ws[64];
        static size_t ws_rows[64];
        int dv = 0;
        HIPCHK(hipGetDevice(&dv));
        if (dv < 0 || dv >= 64) return fail("azg_nn_s78_forward_h2: device index");
        if (ws_rows[dv] < (size_t)B) {
            if (ws[dv]) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(ws[dv]); ws[dv] = nullptr; ws_rows[dv] = 0; }
            HIPCHK(hipMalloc(&ws[dv], (size_t)B * 1792 * sizeof(float)));
            ws_rows[dv] = (size_t)B;
        }
        constexpr size_t lds_g = (size_t)2 * 64 * (160 * 2 + 16);
        static bool attr_g = false;
        if (!attr_g) {
            HIPCHK(hipFuncSetAttribute((const void*)k_s78_policy_gemm_h2<1782, 132>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));
            attr_g = true;
        }
        k_s78_policy_gemm_h2<1782, 132><<<dim3((B + 63) / 64, 4), dim3(768), lds_g, (hipStream_t)stream>>>((const uint4*)N.Wfp, N.bfp, B, pi, ws[dv]);
        HIPCHK(hipGetLastError());
        k_s78_policy_softmax<1782><<<dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(ws[dv], valid, B, pi);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (split == 2) {                       // (the policy FC on f16 x 2 operands as well: w[11] then holds its hi / lo fragments + the descale)
        constexpr size_t lds_h2 = (size_t)(16 * (160 + 4) + 16 * (112 * 16 + 4)) * sizeof(float);
        static bool attr_h2 = false;
        if (!attr_h2) {
            HIPCHK(hipFuncSetAttribute((const void*)k_s78_policy_h2<1782, 132>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h2));
            attr_h2 = true;
        }
        k_s78_policy_h2<1782, 132><<<dim3((B + 15) / 16), dim3(768), lds_h2, (hipStream_t)stream>>>((const uint4*)N.Wfp, N.bfp, valid, B, pi);
        HIPCHK(hipGetLastError());
        return 0;
    }
    constexpr size_t lds_pi = (size_t)(16 * (144 + 4) + 16 * (112 * 16 + 4)) * sizeof(float);
    static bool attr_pi = false;
    if (!attr_pi) {
        HIPCHK(hipFuncSetAttribute((const void*)k_s78_policy<1782, 132>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pi));
        attr_pi = true;
    }
    k_s78_policy<1782, 132><<<dim3((B + 15) / 16), dim3(768), lds_pi, (hipStream_t)stream>>>(N.Wfp, N.bfp, valid, B, pi);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_s78_forward(const int8_t* boards, const uint8_t* valid, const float* const* w, int n_blocks, int A, int P,
                                  int B, float* pi, float* v, void* stream) {
    return s78_launch(boards, valid, w, n_blocks, A, P, B, pi, v, stream, 0);
}

extern "C" int azg_nn_s78_forward_split(const int8_t* boards, const uint8_t* valid, const float* const* w, int n_blocks, int A, int P,
                                        int B, float* pi, float* v, void* stream) {
    return s78_launch(boards, valid, w, n_blocks, A, P, B, pi, v, stream, 3);
}

extern "C" int azg_nn_s78_forward_h2(const int8_t* boards, const uint8_t* valid, const float* const* w, float ds_e, float ds_p, int n_blocks,
                                     int A, int P, int B, float* pi, float* v, void* stream) {
    return s78_launch(boards, valid, w, n_blocks, A, P, B, pi, v, stream, 2, ds_e, ds_p);
}

extern "C" int azg_nn_board_to_x_ld(const int8_t* boards, float* x, int B, int C, int L, int ldx, void* stream) {
    if (!boards || !x || B <= 0 || L <= 0 || ldx < C) return fail("azg_nn_board_to_x: null/empty argument");
    const long long total = (long long)B * L * ldx;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    k_board_to_x<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(boards, x, B, C, L, ldx);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_nn_board_to_x(const int8_t* boards, float* x, int B, int C, void* stream) {
    return azg_nn_board_to_x_ld(boards, x, B, C, 7, C, stream);
}

extern "C" int azg_nn_heads_out(const float* logits, int ldl, const uint8_t* valid, const float* vhid, int ldv,
                                const float* Wv2, const float* bv2, float* pi, float* v, int B, int A, int P,
                                void* stream) {
    if (!logits || !valid || !pi || !v || B <= 0) return fail("azg_nn_heads_out: null/empty argument");
    if (A > 256 || P > 4) return fail("azg_nn_heads_out: A <= 256, P <= 4");
    k_heads_out<<<dim3(B), dim3(64), 0, (hipStream_t)stream>>>(logits, ldl, valid, vhid, ldv, Wv2, bv2, pi, v, B, A, P);
    HIPCHK(hipGetLastError());
    return 0;
}

#include "azg_fused.hip.h"
