// nn_conv5x5.cuh -- the Santorini ResNet (santorini/SantoriniNNet.py nn_version 88/89: first 3x3 conv 2 -> 64 + BN + ReLU,
// NB SimpleResBlocks :71-84 of two 3x3 convs 64 -> 64, SimpleHead policy / value heads :17-40) as ONE launch per leaf
// batch.  A workgroup owns 8 samples = 200 board cells; the two activation tiles [200][64] f32 live in LDS, each 3x3
// convolution is an implicit GEMM out[cell][co] = sum_{tap, ci} in[cell + tap][ci] * W[tap*64 + ci][co] on
// v_mfma_f32_16x16x4_f32 (weights = A operand in fragment order, activations = B operand read as float4 from LDS; a tap
// that leaves the 5x5 board contributes zeros).  12 waves = 4 output-channel tiles x 3 groups of cell tiles; a wave keeps
// the weight fragments of one kernel row (3 taps x 64 channels = 48 VGPRs) in registers while it walks its cell tiles, and
// the accumulators of its (at most 5) cell tiles across the three kernel rows.  The heads are a few thousand MACs per
// sample and run on the vector ALUs.  (MIOpen's Winograd path needs 11 launches of ~100 us for the same batch.)
#pragma once
#include "nn_kernels.cuh"

namespace azg {

#pragma clang fp contract(fast)

struct Conv5NetW {
    const float *W0, *b0;             // first conv: [9*16][64] fragment order (input channels padded 2 -> 16), bias[64]
    const float *Wc, *bc;             // 2*NB trunk convs: each [9*64][64] fragment order (BN folded), bias [2*NB][64]
    const float *Wp, *bp;             // policy head 1x1 conv [64][CP2] plain + bias[CP2]  (CP2 = 2)
    const float *Wfp, *bfp;           // policy FC [CP2*25][A] plain (row = c*25 + cell), bias [A]
    const float *Wv, *bv;             // value head 1x1 conv [64] + bias[1]
    const float *Wf1, *bf1;           // value fc1 [25][64], bias [64]
    const float *Wf2, *bf2;           // value fc2 [64][P], bias [P]
};

// one 3x3 convolution over the workgroup's tile.  KC = input-channel chunks of 16 per tap.
//   IN [ROWS][CS] -> OUT [ROWS][CS] = relu(conv(IN) + bias (+ RES)); OUT may alias RES (same lane reads and writes an element)
template <int KC, int NS>
__device__ __forceinline__ void conv3x3_tile(const float* __restrict__ Wfrag, const float* __restrict__ bias,
                                             const float* IN, float* OUT, const float* RES) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, CS = 68, KCH = 9 * KC, RG = 3, MAXT = (RT + RG - 1) / RG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int ct = wave & 3, rg = wave >> 2;
    // this lane's cell of each of the wave's tiles: row index, and which of the 9 taps stay on the board
    int row[MAXT];
    uint32_t tapmask[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int rt = rg + RG * i, r = rt * 16 + r16;
        row[i] = r;
        uint32_t m = 0;
        if (rt < RT && r < ROWS) {
            const int cell = r % 25, y = cell / 5, x = cell - 5 * y;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < 5 && xx >= 0 && xx < 5) m |= 1u << t;
            }
        }
        tapmask[i] = m;
    }
    f32x4 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < 3; ky++) {
        float4 w[3 * KC];
#pragma unroll
        for (int c = 0; c < 3 * KC; c++) w[c] = FRAG(Wfrag, KCH, ct, ky * 3 * KC + c);
#pragma unroll
        for (int i = 0; i < MAXT; i++) {
            if (rg + RG * i >= RT) continue;
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int t = ky * 3 + kx;
                const bool on = (tapmask[i] >> t) & 1u;
                const float* src = IN + (on ? row[i] + (ky - 1) * 5 + (kx - 1) : 0) * CS + 4 * g;
#pragma unroll
                for (int c = 0; c < KC; c++) {
                    float4 a = *(const float4*)(src + 16 * c);
                    if (!on) a = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 ww = w[kx * KC + c];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ww.x, a.x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ww.y, a.y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ww.z, a.z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ww.w, a.w, acc[i], 0, 0, 0);
                }
            }
        }
    }
    const float4 b = *(const float4*)(bias + ct * 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        if (rg + RG * i >= RT || row[i] >= ROWS) continue;
        float4 o = make_float4(acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w);
        float* dst = OUT + row[i] * CS + ct * 16 + 4 * g;
        if (RES) {
            const float4 r = *(const float4*)(RES + row[i] * CS + ct * 16 + 4 * g);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        *(float4*)dst = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    }
}

template <int NB, int A, int P>
__global__ __launch_bounds__(768) void k_conv5_net(Conv5NetW N, const int8_t* __restrict__ boards,
                                                   const uint8_t* __restrict__ valid, int B, float* __restrict__ pi_out,
                                                   float* __restrict__ v_out) {
    constexpr int NS = 8, ROWS = NS * 25, CS = 68, CP2 = 2, AS = (A + 3) / 4 * 4 + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                        // [ROWS][CS]
    float* Y = X + ROWS * CS;               // [ROWS][CS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * NS, nb = min(NS, B - b0);
    // ---- board int8 [s][y][x][3] -> Y[s*25 + cell][plane 0..1], channels 2..15 zero (the first conv reads 16) ----
    for (int i = tid; i < ROWS * 4; i += 768) *(float4*)(Y + (i >> 2) * CS + 4 * (i & 3)) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < nb * 25 * 2; i += 768) {
        const int r = i >> 1, pl = i & 1;
        Y[r * CS + pl] = (float)boards[(size_t)b0 * 75 + r * 3 + pl];
    }
    __syncthreads();
    conv3x3_tile<1, NS>(N.W0, N.b0, Y, X, nullptr);
    __syncthreads();
#pragma unroll 1
    for (int blk = 0; blk < NB; blk++) {
        conv3x3_tile<4, NS>(N.Wc + (size_t)(2 * blk) * (9 * 64 * 64), N.bc + (2 * blk) * 64, X, Y, nullptr);
        __syncthreads();
        conv3x3_tile<4, NS>(N.Wc + (size_t)(2 * blk + 1) * (9 * 64 * 64), N.bc + (2 * blk + 1) * 64, Y, X, X);
        __syncthreads();
    }
    // ---- heads (SimpleHead): 1x1 conv + BN + ReLU -> flatten (channel-major) -> FC ----
    float* HP = Y;                          // [NS][CP2*25]   policy head features
    float* HV = HP + NS * CP2 * 25;         // [NS][25]       value head features
    float* LG = HV + NS * 25;               // [NS][AS]       logits
    float* H1 = LG + NS * AS;               // [NS][64]       value fc1
    for (int i = tid; i < NS * 25 * (CP2 + 1); i += 768) {
        const int r = i / (CP2 + 1), c = i - r * (CP2 + 1);
        const float* xr = X + r * CS;
        float a = c < CP2 ? N.bp[c] : N.bv[0];
#pragma unroll 8
        for (int k = 0; k < 64; k++) a += xr[k] * (c < CP2 ? N.Wp[k * CP2 + c] : N.Wv[k]);
        a = fmaxf(a, 0.f);
        const int s = r / 25, cell = r - 25 * s;
        if (c < CP2) HP[s * (CP2 * 25) + c * 25 + cell] = a; else HV[s * 25 + cell] = a;
    }
    __syncthreads();
    for (int i = tid; i < NS * A; i += 768) {
        const int s = i / A, a = i - s * A;
        float acc = N.bfp[a];
        for (int k = 0; k < CP2 * 25; k++) acc += HP[s * (CP2 * 25) + k] * N.Wfp[k * A + a];
        LG[s * AS + a] = acc;
    }
    for (int i = tid; i < NS * 64; i += 768) {
        const int s = i >> 6, j = i & 63;
        float acc = N.bf1[j];
        for (int k = 0; k < 25; k++) acc += HV[s * 25 + k] * N.Wf1[k * 64 + j];
        H1[s * 64 + j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    // masked softmax == exp(log_softmax(where(valid, logits, -1e8))), one wave per sample
    for (int s = wave; s < nb; s += 12) {
        const int b = b0 + s;
        float x[(A + 63) / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < (A + 63) / 64; k++) {
            const int a = lane + 64 * k;
            x[k] = -INFINITY;
            if (a < A) x[k] = valid[(size_t)b * A + a] ? LG[s * AS + a] : -1e8f;
            mx = fmaxf(mx, x[k]);
        }
        mx = nn_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < (A + 63) / 64; k++) { x[k] = (lane + 64 * k < A) ? expf(x[k] - mx) : 0.f; sum += x[k]; }
        sum = nn_wave_sum(sum);
#pragma unroll
        for (int k = 0; k < (A + 63) / 64; k++)
            if (lane + 64 * k < A) pi_out[(size_t)b * A + lane + 64 * k] = x[k] / sum;
    }
    if (tid < nb * P) {
        const int s = tid / P, p = tid - s * P;
        float acc = N.bf2[p];
        for (int j = 0; j < 64; j++) acc += H1[s * 64 + j] * N.Wf2[j * P + p];
        v_out[(size_t)(b0 + s) * P + p] = tanhf(acc);
    }
}

#pragma clang fp contract(off)

}  // namespace azg
