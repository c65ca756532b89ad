// nn_mb1d.hip.h -- the MobileNetV3-1d policy/value net (splendor/SplendorNNet.py:259-283,397-440 for 2-4 players,
// azul/AzulNNet.py:91-113,130-142) as ONE launch for any geometry: the generic sibling of k_v80_net (nn_kernels.hip.h), which
// is hand-laid-out for the 2-player Splendor shape.  Same data flow -- a workgroup owns NS samples, every activation lives
// in LDS, all GEMMs are v_mfma_f32_16x16x4_f32 with the weight tile as the A operand (fragment order, see FRAG) and the
// activations as the B operand read as float4 from LDS -- but tile loops instead of a fixed wave->tile map:
//   first layer -> trunk block (X -> X2) -> policy block (X2 -> O) + Linear/ReLU/Linear/masked softmax
//                                        -> value  block (X2 -> O) + Linear/ReLU/Linear/tanh
// Channel counts are zero-padded to multiples of 16 (weights, biases) so no bounds logic is needed inside the GEMMs; LDS
// rows carry 4 floats of padding (row stride = 4 mod 8 floats: conflict-free float4 fragment reads).  The LDS is cleared
// once at kernel start: every padding element that a K loop can reach is then a finite number times a zero weight.
#pragma once
#include "nn_kernels.hip.h"
#include "nn_v80_h2.hip.h"

namespace azg {

#pragma clang fp contract(fast)

struct Mb1dBlockW { const float *We, *be, *Wd, *sd, *bd, *W1, *b1, *W2, *b2, *Wp, *bp; };
struct Mb1dNetW {
    const float *W0, *b0;
    Mb1dBlockW blk[3];                                  // trunk, policy head, value head
    const float *Wpi1, *bpi1, *Wpi2, *bpi2;             // [L*OS -> A] (rows l*OS + c), [A -> A]      fragment order
    const float *Wv1, *bv1, *Wv2, *bv2;                 // [L*OS -> P] fragment order; Wv2 [P][P] plain (in, out)
    // H2 kernels (k_mb1d_net<CF, true>): every matrix above except Wd / Wv2 is an h2 fragment array (nn_v80_h2.hip.h: K zero-padded
    // to a multiple of 32, [N/16][K/32][2 planes hi, lo][64 lanes][8] f16 of W * 2^k) and ds[] holds 2^-k / 64 per matrix:
    // W0, {We, W1, W2, Wp} x 3 blocks, Wpi1, Wpi2, Wv1
    float ds[16];
};

constexpr int mb_r16(int n) { return (n + 15) / 16 * 16; }

// Geometry of one net.  E / Q / CO / ACT / PMAX per block (trunk, policy head, value head).
template <int L_, int C_, int NS_, int A_, int P_, int E0, int E1, int E2, int Q0, int Q1, int Q2, int CO1, int ACT0, int ACT12,
          int PMAX12>
struct Mb1dCfg {
    static constexpr int L = L_, C = C_, NS = NS_, A = A_, P = P_, NW = 12;
    static constexpr int E[3] = {E0, E1, E2}, Q[3] = {Q0, Q1, Q2}, CO[3] = {C_, CO1, C_};
    static constexpr int ACT[3] = {ACT0, ACT12, ACT12}, PMAX[3] = {0, PMAX12, PMAX12};
    static constexpr int ROWS = NS * L, ROWSP = mb_r16(ROWS), RT = ROWSP / 16;
    static constexpr int CP = mb_r16(C), XS = CP + 4;
    static constexpr int COPmax = mb_r16(CO1 > C_ ? CO1 : C_), OS = COPmax + 4;     // head block output row stride
    static constexpr int EPmax = mb_r16(E0 > E1 ? (E0 > E2 ? E0 : E2) : (E1 > E2 ? E1 : E2)), HS = EPmax + 28;     // (+28: = 4 mod 8 floats, and room for the project GEMM's f16 planes of its K-padded operand, mb_block)
    static constexpr int QPmax = mb_r16(Q0 > Q1 ? (Q0 > Q2 ? Q0 : Q2) : (Q1 > Q2 ? Q1 : Q2)), QS = QPmax + 4;
    static constexpr int AP = mb_r16(A), AS = AP + 4;
    // LDS map (floats)
    static constexpr int XA_SZ = ROWSP * (XS > OS ? XS : OS), X2_SZ = ROWSP * XS, H_SZ = ROWSP * HS;
    static constexpr int PL_SZ = NS * HS, SC_SZ = (ROWSP / L + 1) * HS, SH_SZ = 16 * QS;
    static constexpr int KS_PI = AP / 16 >= NW ? 1 : NW / (AP / 16);    // K slices of the policy head GEMMs
    static constexpr int HEAD_PI = (KS_PI + 1) * 16 * AS, HEAD_V = NW * 16 * 20;     // RED[KS][16][AS] + HID; value RED[NW][16][20]
    static constexpr int HEAD_SZ = HEAD_PI > HEAD_V ? HEAD_PI : HEAD_V;              // aliases H
    static constexpr int H_ALLOC = H_SZ > HEAD_SZ ? H_SZ : HEAD_SZ;
    // the token-mix matrix Wd[L][L]: up to 8 tokens it is pinned in scalar registers (row stride L); beyond that it stays in LDS (rows padded to
    // a multiple of 4 floats) and is read by broadcast
    static constexpr bool WD_REGS = L_ * L_ <= 64;
    static constexpr int WD_LD = WD_REGS ? L_ : (L_ + 3) / 4 * 4, WD_SZ = WD_REGS ? 64 : L_ * WD_LD;
    static constexpr int LDS_FLOATS = XA_SZ + X2_SZ + H_ALLOC + PL_SZ + SC_SZ + SH_SZ + WD_SZ;
};

// One GEMM phase over the workgroup:  out(row, 16*ct + 4g .. +3) = epi( sum_k in[row][k] * W[k][col] )
//   KCH K-chunks of 16, NT column tiles, RTN row tiles; loadB(rt, c) returns this lane's float4 of the B operand
//   (activation row rt*16 + r16, K offset 16c + 4g); epi(ct, rt, acc) consumes the C tile (lane: row r16, 4 columns).
// The weight fragments of a column tile stay in registers across the row tiles a wave computes for it.
template <int KCH, int NT, int RTN, int NW, class LoadB, class Epi>
__device__ __forceinline__ void mb_gemm(const float* __restrict__ Wfrag, LoadB loadB, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the NT x RTN output tiles in column-major order, an equal contiguous share per wave: a wave changes its column tile
    // (= reloads the weight fragments into registers) at most (NT + NW - 1) / NW + 1 times
    constexpr int U = NT * RTN;
    const int u0 = wave * U / NW, u1 = (wave + 1) * U / NW;
    float4 w[KCH];
    int cur = -1;
#pragma unroll 1
    for (int u = u0; u < u1; u++) {
        const int ct = u / RTN, rt = u - ct * RTN;
        if (ct != cur) {
#pragma unroll
            for (int c = 0; c < KCH; c++) w[c] = FRAG(Wfrag, KCH, ct, c);
            cur = ct;
        }
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KCH; c++) {
            const float4 a = loadB(rt, c);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].w, a.w, acc, 0, 0, 0);
        }
        epi(ct, rt, acc);
    }
}

// The same phase on f16 x 2 split-precision operands (nn_v80_h2.hip.h: three v_mfma_f32_16x16x32_f16 per K chunk of 32 instead of
// eight f32 MFMAs of a quarter of the rate each).  The activations stay f32 in LDS (this kernel's layout is generic) and are split
// as they are read: 8 values = 20 VALU instructions per three MFMAs, which makes the phase VALU-bound at about a third of the f32
// MFMA time.  loadB(rt, k0) returns the float4 at K offset k0; KCH32 chunks of 32 (the weights are zero beyond the real K, the LDS
// holds finite numbers everywhere); `ds` = 2^-k / 64 brings the accumulator back.
// (Round 5: the H2 kernels keep 64 * x in the LDS -- every writer of an activation scales it once, H2_AS, an exact power of two, and the few
// readers that are not GEMM operands scale back -- so the split of a GEMM operand no longer multiplies: 12 instead of 20 VALU instructions
// per 8 values, in every (column tile, row tile, K chunk) step; the outputs keep their bits.)
__device__ __forceinline__ void mb_split8(float4 a, float4 b, uint4& hi, uint4& lo) {
    h2_split2(a.x, a.y, hi.x, lo.x); h2_split2(a.z, a.w, hi.y, lo.y);
    h2_split2(b.x, b.y, hi.z, lo.z); h2_split2(b.z, b.w, hi.w, lo.w);
}
template <int KCH32, int NT, int RTN, int NW, class LoadB, class Epi>
__device__ __forceinline__ void mb_gemm_h2(const float* __restrict__ Wfrag_, float ds, LoadB loadB, Epi epi) {
    const uint4* __restrict__ Wfrag = (const uint4*)Wfrag_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    constexpr int U = NT * RTN;
    const int u0 = wave * U / NW, u1 = (wave + 1) * U / NW;
    uint4 wh[KCH32], wl[KCH32];
    int cur = -1;
#pragma unroll 1
    for (int u = u0; u < u1; u++) {
        const int ct = u / RTN, rt = u - ct * RTN;
        if (ct != cur) {
#pragma unroll
            for (int c = 0; c < KCH32; c++) { wh[c] = H2FRAG(Wfrag, KCH32, ct, c, 0); wl[c] = H2FRAG(Wfrag, KCH32, ct, c, 1); }
            cur = ct;
        }
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int c = 0; c < KCH32; c++) {
            uint4 ah, al;
            mb_split8(loadB(rt, 32 * c + 8 * g), loadB(rt, 32 * c + 8 * g + 4), ah, al);
            if (c & 1) a1 = h2_mma(wh[c], wl[c], ah, al, a1); else a0 = h2_mma(wh[c], wl[c], ah, al, a0);
        }
        epi(ct, rt, (a0 + a1) * ds);
    }
}
// The same GEMM phase with the B operand already split: two f16 planes (hi, lo) of row stride PRS halves in LDS -- a K chunk of a row
// tile is two 16-byte reads and no VALU work at all.  Used for the project GEMM, whose operand (the expanded tile times the SE scale) is
// written once per block and read once per column tile of the output.
template <int KCH32, int NT, int RTN, int NW, int PRS, class Epi>
__device__ __forceinline__ void mb_gemm_h2p(const float* __restrict__ Wfrag_, float ds, const uint8_t* __restrict__ hi, const uint8_t* __restrict__ lo, Epi epi) {
    const uint4* __restrict__ Wfrag = (const uint4*)Wfrag_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    constexpr int U = NT * RTN;
    const int u0 = wave * U / NW, u1 = (wave + 1) * U / NW;
    uint4 wh[KCH32], wl[KCH32];
    int cur = -1;
#pragma unroll 1
    for (int u = u0; u < u1; u++) {
        const int ct = u / RTN, rt = u - ct * RTN;
        if (ct != cur) {
#pragma unroll
            for (int c = 0; c < KCH32; c++) { wh[c] = H2FRAG(Wfrag, KCH32, ct, c, 0); wl[c] = H2FRAG(Wfrag, KCH32, ct, c, 1); }
            cur = ct;
        }
        const int off = ((rt * 16 + r16) * PRS + 8 * g) * 2;
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int c = 0; c < KCH32; c++) {
            const uint4 ah = *(const uint4*)(hi + off + 64 * c), al = *(const uint4*)(lo + off + 64 * c);
            if (c & 1) a1 = h2_mma(wh[c], wl[c], ah, al, a1); else a0 = h2_mma(wh[c], wl[c], ah, al, a0);
        }
        epi(ct, rt, (a0 + a1) * ds);
    }
}
template <int KCH32, int NT, int NW, int AS, int NROWS>
__device__ __forceinline__ int mb_head_gemm_h2(const float* __restrict__ Wfrag_, float ds, const float* __restrict__ in, int ldi,
                                               float* __restrict__ RED) {
    const uint4* __restrict__ Wfrag = (const uint4*)Wfrag_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int rr = r16 < NROWS ? r16 : 0;
    constexpr int KS = NT >= NW ? 1 : NW / NT;
    constexpr int PER = (KCH32 + KS - 1) / KS;
    for (int u = wave; u < NT * KS; u += NW) {
        const int ct = u % NT, ks = u / NT;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c_end = (ks + 1) * PER < KCH32 ? (ks + 1) * PER : KCH32;
#pragma unroll 2
        for (int c = ks * PER; c < c_end; c++) {
            const uint4 wh = H2FRAG(Wfrag, KCH32, ct, c, 0), wl = H2FRAG(Wfrag, KCH32, ct, c, 1);
            uint4 ah, al;
            mb_split8(*(const float4*)(in + rr * ldi + 32 * c + 8 * g), *(const float4*)(in + rr * ldi + 32 * c + 8 * g + 4), ah, al);
            acc = h2_mma(wh, wl, ah, al, acc);
        }
        acc = acc * ds;
        *(float4*)(RED + (ks * 16 + r16) * AS + ct * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    return KS;
}

// Flatten -> Linear on one 16-row tile (rows = samples): K is long and every weight is used once, so the fragments are
// streamed; the NW waves split (column tile, K slice) and leave partial sums in RED[slice][16][AS].  Returns the slice count.
template <int KCH, int NT, int NW, int AS, int NROWS>
__device__ __forceinline__ int mb_head_gemm(const float* __restrict__ Wfrag, const float* __restrict__ in, int ldi,
                                            float* __restrict__ RED) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const int rr = r16 < NROWS ? r16 : 0;                // rows >= NROWS do not exist: recompute row 0, never used
    constexpr int KS = NT >= NW ? 1 : NW / NT;           // K slices
    constexpr int PER = (KCH + KS - 1) / KS;
    for (int u = wave; u < NT * KS; u += NW) {
        const int ct = u % NT, ks = u / NT;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c_end = (ks + 1) * PER < KCH ? (ks + 1) * PER : KCH;
#pragma unroll 4
        for (int c = ks * PER; c < c_end; c++) {
            const float4 w = FRAG(Wfrag, KCH, ct, c);
            const float4 a = *(const float4*)(in + rr * ldi + 16 * c + 4 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc, 0, 0, 0);
        }
        *(float4*)(RED + (ks * 16 + r16) * AS + ct * 16 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    return KS;
}

// InvertedResidual1d block bi (SplendorNNet.py:189-202): IN [ROWSP][XS] -> OUT [..][OSTRIDE]
// one GEMM phase in either arithmetic: K16 chunks of 16 for the f32 path, (K16 + 1) / 2 chunks of 32 for the f16 x 2 path
template <bool H2, int K16, int NT, int RTN, int NW, class LoadK, class Epi>
__device__ __forceinline__ void mb_gemm_any(const float* __restrict__ Wfrag, float ds, LoadK loadK, Epi epi) {
    const int g = (threadIdx.x & 63) >> 4;
    if (H2) mb_gemm_h2<(K16 + 1) / 2, NT, RTN, NW>(Wfrag, ds, loadK, epi);
    else mb_gemm<K16, NT, RTN, NW>(Wfrag, [&](int rt, int c) { return loadK(rt, 16 * c + 4 * g); }, epi);
}

template <class CF, int BI, int CIN_P, int OSTRIDE, bool H2 = false>
__device__ __forceinline__ void mb_block(const Mb1dBlockW& W, const float* IN, float* OUT, float* H, float* PL, float* SC,
                                         float* SH, float* WD, bool residual, const float* ds = nullptr) {
    constexpr int L = CF::L, NS = CF::NS, NW = CF::NW, RT = CF::RT, XS = CF::XS, HS = CF::HS, QS = CF::QS;
    constexpr int EP = mb_r16(CF::E[BI]), QP = mb_r16(CF::Q[BI]), COP = mb_r16(CF::CO[BI]);
    constexpr int ACT = CF::ACT[BI], PMAX = CF::PMAX[BI];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
    constexpr float SA = H2 ? H2_AS : 1.f, ISA = H2 ? H2_IAS : 1.f;       // what the LDS holds of an activation x: SA * x
    for (int i = tid; i < L * L; i += NW * 64) WD[(i / L) * CF::WD_LD + i % L] = W.Wd[i];
    // ---- expand + BN + act -> H ----
    (void)g;
    mb_gemm_any<H2, CIN_P / 16, EP / 16, RT, NW>(
        W.We, H2 ? ds[0] : 1.f, [&](int rt, int k0) { return *(const float4*)(IN + (rt * 16 + r16) * XS + k0); },
        [&](int ct, int rt, f32x4 acc) {
            const float4 b = *(const float4*)(W.be + ct * 16 + 4 * g);
            const f32x2 lo = act_apply2(f32x2{acc[0] + b.x, acc[1] + b.y}, ACT) * SA, hi = act_apply2(f32x2{acc[2] + b.z, acc[3] + b.w}, ACT) * SA;
            *(float4*)(H + (rt * 16 + r16) * HS + ct * 16 + 4 * g) = make_float4(lo.x, lo.y, hi.x, hi.y);
        });
    __syncthreads();
    // ---- depthwise Linear(L->L) over the tokens + BN + act (in place) + SE squeeze ----
    {
        float wd[CF::WD_REGS ? L * L : 1];
        if constexpr (CF::WD_REGS) {
#pragma unroll
            for (int k = 0; k < L * L; k++) wd[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, WD[k])));
        }
        for (int i = tid; i < NS * (EP / 2); i += NW * 64) {
            const int s = i / (EP / 2), c = 2 * (i - s * (EP / 2));
            float* base = H + (s * L) * HS + c;
            f32x2 in[L];
#pragma unroll
            for (int l = 0; l < L; l++) in[l] = *(const f32x2*)(base + l * HS);
            const f32x2 scl = *(const f32x2*)(W.sd + c), bb = *(const f32x2*)(W.bd + c) * SA;      // (the sums below are SA * the reference's)
            f32x2 pool = PMAX ? f32x2{-INFINITY, -INFINITY} : f32x2{0.f, 0.f};
#pragma unroll
            for (int m = 0; m < L; m++) {
                f32x2 a = f32x2{0.f, 0.f};
#pragma unroll
                for (int l = 0; l < L; l++) {
                    if constexpr (CF::WD_REGS) a += wd[m * L + l] * in[l];
                    else a += WD[m * CF::WD_LD + l] * in[l];
                }
                a = H2 ? act_apply2((a * scl + bb) * ISA, ACT) * SA : act_apply2(a * scl + bb, ACT);
                *(f32x2*)(base + m * HS) = a;
                if (PMAX) { pool.x = fmaxf(pool.x, a.x); pool.y = fmaxf(pool.y, a.y); } else pool += a;
            }
            if (!PMAX) pool = pool * (1.f / (float)L);
            *(f32x2*)(PL + s * HS + c) = pool;
        }
    }
    __syncthreads();
    // ---- SE fc1 + ReLU -> SH[16][QS];  SE fc2 + Hardsigmoid -> SC[.][HS]  (rows = samples; rows >= NS are scratch) ----
    mb_gemm_any<H2, EP / 16, QP / 16, 1, NW>(
        W.W1, H2 ? ds[1] : 1.f, [&](int, int k0) { return *(const float4*)(PL + r16 * HS + k0); },
        [&](int ct, int, f32x4 acc) {
            const float4 b = *(const float4*)(W.b1 + ct * 16 + 4 * g);
            *(float4*)(SH + r16 * QS + ct * 16 + 4 * g) =
                make_float4(fmaxf(acc[0] + b.x, 0.f) * SA, fmaxf(acc[1] + b.y, 0.f) * SA, fmaxf(acc[2] + b.z, 0.f) * SA, fmaxf(acc[3] + b.w, 0.f) * SA);
        });
    __syncthreads();
    mb_gemm_any<H2, QP / 16, EP / 16, 1, NW>(
        W.W2, H2 ? ds[2] : 1.f, [&](int, int k0) { return *(const float4*)(SH + r16 * QS + k0); },
        [&](int ct, int, f32x4 acc) {
            if (r16 < NS) {
                const float4 b = *(const float4*)(W.b2 + ct * 16 + 4 * g);
                *(float4*)(SC + r16 * HS + ct * 16 + 4 * g) =
                    make_float4(hardsigmoid(acc[0] + b.x), hardsigmoid(acc[1] + b.y), hardsigmoid(acc[2] + b.z), hardsigmoid(acc[3] + b.w));
            }
        });
    __syncthreads();
    // the project GEMM's K extent (chunks of 32) and the row stride of its operand planes: + 8 halves = conflict-free 16-byte reads
    constexpr int KP32 = (EP / 16 + 1) / 2, EPK = 32 * KP32, PRS = EPK + 8;
    static_assert(!H2 || (PRS * 4 <= HS * 4 && PRS % 8 == 0), "the two f16 planes fit the expanded tile");
    uint8_t* const PHI = (uint8_t*)H;
    uint8_t* const PLO = PHI + CF::ROWSP * PRS * 2;
    if constexpr (H2) {
        // The operand of the project GEMM, (expanded tile) x (SE scale), is formed ONCE, split into its f16 hi / lo planes, and written
        // over the expanded tile: every thread first loads its share of the products (registers), the workgroup meets, then the planes are
        // stored (the same values, the same split as when the GEMM split them as it read them: the outputs keep their bits).  Columns
        // EP .. EPK of the K padding are zeroed: the weights there are zero, but a stale bit pattern read as f16 may be a NaN.
        constexpr int Q4 = EPK / 4, TOT = CF::ROWSP * Q4, NE = (TOT + NW * 64 - 1) / (NW * 64);
        float4 v[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int i = tid + e * NW * 64, row = i / Q4, k0 = 4 * (i - row * Q4);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < TOT && k0 < EP) {
                a = *(const float4*)(H + row * HS + k0);
                const float4 s4 = *(const float4*)(SC + (row / L) * HS + k0);
                a.x *= s4.x; a.y *= s4.y; a.z *= s4.z; a.w *= s4.w;
            }
            v[e] = a;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int i = tid + e * NW * 64, row = i / Q4, k0 = 4 * (i - row * Q4);
            if (i < TOT) {
                uint32_t h0, l0, h1, l1;
                h2_split2(v[e].x, v[e].y, h0, l0); h2_split2(v[e].z, v[e].w, h1, l1);
                *(uint2*)(PHI + (row * PRS + k0) * 2) = make_uint2(h0, h1);
                *(uint2*)(PLO + (row * PRS + k0) * 2) = make_uint2(l0, l1);
            }
        }
        __syncthreads();
    }
    // ---- project (SE-scaled operand) + BN (+ residual) -> OUT ----
    auto project_epi = [&](int ct, int rt, f32x4 acc) {
            const int row = rt * 16 + r16, col0 = ct * 16 + 4 * g;
            const float4 b = *(const float4*)(W.bp + col0);
            float4 o = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
            if (residual) {
                const float4 x = *(const float4*)(IN + row * XS + col0);
                o.x += x.x * ISA; o.y += x.y * ISA; o.z += x.z * ISA; o.w += x.w * ISA;
            }
            *(float4*)(OUT + row * OSTRIDE + col0) = make_float4(o.x * SA, o.y * SA, o.z * SA, o.w * SA);
        };
    if constexpr (H2) mb_gemm_h2p<KP32, COP / 16, RT, NW, PRS>(W.Wp, ds[3], PHI, PLO, project_epi);
    else
        mb_gemm<EP / 16, COP / 16, RT, NW>(
            W.Wp,
            [&](int rt, int c) {
                const int row = rt * 16 + r16, k0 = 16 * c + 4 * g;
                float4 a = *(const float4*)(H + row * HS + k0);
                const float4 s4 = *(const float4*)(SC + (row / L) * HS + k0);
                a.x *= s4.x; a.y *= s4.y; a.z *= s4.z; a.w *= s4.w;
                return a;
            },
            project_epi);
    __syncthreads();
}

// The forward of one workgroup (samples NS * wg ..) as a device function: the body of k_mb1d_net and -- IND -- of the asynchronous pipeline's
// net kernel (azg_async.hip.h): sample s of the workgroup is then tree sidx[s] (LDS; < 0 = no sample), `boards` / `valid` are both the
// pipeline's leaf-record array (kernels.hip.h AsyncLeaf<G>: int8 state [SP] + valid bit mask u64[AW], stride AL_STRIDE, mask at AL_MASK),
// read past the L1; the samples' masks are fetched with the boards into `smask` (LDS u64 [NS][AW]); pi / v rows are written WRITE-THROUGH
// at the tree's index.  The weight table is read through the CONSTANT address space wherever it is used (the kernel's own argument
// segment / the pipeline's argument block).
typedef const Mb1dNetW __attribute__((address_space(4))) * Mb1dNetWC;
template <class T>
__device__ __forceinline__ T mb_ldc(const T __attribute__((address_space(4))) * p) {          // word-wise copy out of the constant address space
    static_assert(sizeof(T) % 4 == 0, "word-sized struct");
    const uint32_t __attribute__((address_space(4))) * w = (const uint32_t __attribute__((address_space(4))) *)p;
    uint32_t a[sizeof(T) / 4];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); k++) a[k] = w[k];
    T out;
    __builtin_memcpy(&out, a, sizeof(T));
    return out;
}
struct Mb1dDs4 { float v[4]; };
#define N (*Np)
template <class CF, bool H2, bool IND, int AL_STRIDE = 0, int AL_MASK = 0>
__device__ __forceinline__ void mb1d_net_body(float* smem, const Mb1dNetWC Np, const int8_t* __restrict__ boards,
                                              const uint8_t* __restrict__ valid, int B, float* __restrict__ pi_out,
                                              float* __restrict__ v_out, const int wg, const int* sidx = nullptr,
                                              unsigned long long* smask = nullptr) {
    constexpr int L = CF::L, C = CF::C, NS = CF::NS, NW = CF::NW, RT = CF::RT, XS = CF::XS, OS = CF::OS, HS = CF::HS, A = CF::A,
                  P = CF::P, AS = CF::AS, CP = CF::CP;
    if (H2) h2_fp16_saturate_mode();        // out-of-range activations saturate instead of becoming inf (inf * zero padding = NaN)
    constexpr float SA = H2 ? H2_AS : 1.f;  // what the LDS holds of an activation x: SA * x (mb_split8)
    float* XA = smem;                       // first-layer output, later the head blocks' output O (row stride OS)
    float* X2 = XA + CF::XA_SZ;             // trunk output
    float* H = X2 + CF::X2_SZ;              // expanded activations; board tile before, head temporaries after
    float* PL = H + CF::H_ALLOC;
    float* SC = PL + CF::PL_SZ;
    float* SH = SC + CF::SC_SZ;
    float* WD = SH + CF::SH_SZ;
    int tid_ = threadIdx.x;
    if (IND) asm volatile("" : "+v"(tid_));            // (opaque inside the pipeline's persistent loop)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int b0 = wg * NS, nb = IND ? NS : min(NS, B - b0);
    constexpr int AWI = (A + 63) / 64;
    unsigned long long ind_mask = 0ull;
    if constexpr (IND) {
        if (tid < NS * AWI) {
            const int b = sidx[tid / AWI];
            if (b >= 0) ind_mask = __hip_atomic_load((const unsigned long long*)(valid + (size_t)b * AL_STRIDE + AL_MASK) + tid % AWI, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    for (int i = tid; i < CF::LDS_FLOATS / 4; i += NW * 64) ((float4*)smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (IND) { if (tid < NS * AWI) smask[tid] = ind_mask; }
    __syncthreads();
    // ---- board tile int8 [s][c][l] -> X0[s*L + l][c] f32 (in H), first layer (+BN) -> XA ----
    float* X0 = H;
    {
        const int8_t* src = boards + (size_t)b0 * (C * L);
        for (int i = tid; i < nb * C * L; i += NW * 64) {
            const int s = i / (C * L), rem = i - s * (C * L), c = rem / L, l = rem - c * L;
            if constexpr (IND) {
                const int b = sidx[s];
                X0[(s * L + l) * XS + c] =
                    b >= 0 ? SA * (float)(int8_t)__hip_atomic_load((const uint8_t*)boards + (size_t)b * AL_STRIDE + rem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            } else
            X0[(s * L + l) * XS + c] = SA * (float)src[i];
        }
    }
    __syncthreads();
    (void)g;
    mb_gemm_any<H2, CP / 16, CP / 16, RT, NW>(
        N.W0, N.ds[0], [&](int rt, int k0) { return *(const float4*)(X0 + (rt * 16 + r16) * XS + k0); },
        [&](int ct, int rt, f32x4 acc) {
            const float4 b = *(const float4*)(N.b0 + ct * 16 + 4 * g);
            *(float4*)(XA + (rt * 16 + r16) * XS + ct * 16 + 4 * g) = make_float4((acc[0] + b.x) * SA, (acc[1] + b.y) * SA, (acc[2] + b.z) * SA, (acc[3] + b.w) * SA);
        });
    __syncthreads();
    {
        const Mb1dBlockW Wb = mb_ldc(&Np->blk[0]);
        const Mb1dDs4 d4 = mb_ldc((const Mb1dDs4 __attribute__((address_space(4))) *)(Np->ds + 1));
        mb_block<CF, 0, CP, XS, H2>(Wb, XA, X2, H, PL, SC, SH, WD, true, d4.v);
    }

    // ================= policy head =================
    {
        const Mb1dBlockW Wb = mb_ldc(&Np->blk[1]);
        const Mb1dDs4 d4 = mb_ldc((const Mb1dDs4 __attribute__((address_space(4))) *)(Np->ds + 5));
        mb_block<CF, 1, CP, OS, H2>(Wb, X2, XA, H, PL, SC, SH, WD, CF::CO[1] == C, d4.v);
    }
    {
        constexpr int KCH1 = (L * OS + 15) / 16, NT1 = CF::AP / 16;
        float* RED = H;
        float* HID = RED + CF::KS_PI * 16 * AS;
        const int ks1 = H2 ? mb_head_gemm_h2<(KCH1 + 1) / 2, NT1, NW, AS, NS>(N.Wpi1, N.ds[13], XA, L * OS, RED)
                           : mb_head_gemm<KCH1, NT1, NW, AS, NS>(N.Wpi1, XA, L * OS, RED);
        __syncthreads();
        for (int i = tid; i < 16 * (CF::AP / 4); i += NW * 64) {
            const int s = i / (CF::AP / 4), col = 4 * (i - s * (CF::AP / 4));
            float4 v = *(const float4*)(N.bpi1 + col);
            for (int k = 0; k < ks1; k++) {
                const float4 p = *(const float4*)(RED + (k * 16 + s) * AS + col);
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            *(float4*)(HID + s * AS + col) = make_float4(fmaxf(v.x, 0.f) * SA, fmaxf(v.y, 0.f) * SA, fmaxf(v.z, 0.f) * SA, fmaxf(v.w, 0.f) * SA);
        }
        __syncthreads();
        const int ks2 = H2 ? mb_head_gemm_h2<(NT1 + 1) / 2, NT1, NW, AS, 16>(N.Wpi2, N.ds[14], HID, AS, RED)
                           : mb_head_gemm<NT1, NT1, NW, AS, 16>(N.Wpi2, HID, AS, RED);
        __syncthreads();
        // masked softmax == exp(log_softmax(where(valid, logits, -1e8))) (GenericNNetWrapper.py:105-107), one wave per sample
        for (int s = wave; s < nb; s += NW) {
            const int b = IND ? sidx[s] : b0 + s;
            if (IND && b < 0) continue;
            float x[(A + 63) / 64];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < (A + 63) / 64; k++) {
                const int a = lane + 64 * k;
                x[k] = -INFINITY;
                if (a < A) {
                    float lg = N.bpi2[a];
                    for (int q = 0; q < ks2; q++) lg += RED[(q * 16 + s) * AS + a];
                    if constexpr (IND) x[k] = ((smask[s * AWI + k] >> lane) & 1ull) ? lg : -1e8f;
                    else x[k] = valid[(size_t)b * A + a] ? lg : -1e8f;
                }
                mx = fmaxf(mx, x[k]);
            }
            mx = nn_wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < (A + 63) / 64; k++) { x[k] = (lane + 64 * k < A) ? expf(x[k] - mx) : 0.f; sum += x[k]; }
            sum = nn_wave_sum(sum);
#pragma unroll
            for (int k = 0; k < (A + 63) / 64; k++)
                if (lane + 64 * k < A) {
                    if constexpr (IND) __hip_atomic_store((uint32_t*)pi_out + (size_t)b * A + lane + 64 * k, __float_as_uint(x[k] / sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else pi_out[(size_t)b * A + lane + 64 * k] = x[k] / sum;
                }
        }
        __syncthreads();
    }

    // ================= value head =================
    {
        const Mb1dBlockW Wb = mb_ldc(&Np->blk[2]);
        const Mb1dDs4 d4 = mb_ldc((const Mb1dDs4 __attribute__((address_space(4))) *)(Np->ds + 9));
        mb_block<CF, 2, CP, OS, H2>(Wb, X2, XA, H, PL, SC, SH, WD, true, d4.v);
    }
    {
        constexpr int KCH1 = (L * OS + 15) / 16;
        float* RED = H;                      // [NW][16][20]
        const int ks = H2 ? mb_head_gemm_h2<(KCH1 + 1) / 2, 1, NW, 20, NS>(N.Wv1, N.ds[15], XA, L * OS, RED)
                          : mb_head_gemm<KCH1, 1, NW, 20, NS>(N.Wv1, XA, L * OS, RED);
        __syncthreads();
        if (tid < nb * P) {
            const int s = tid / P, p = tid - s * P;
            const int b = IND ? sidx[s] : b0 + s;
            if (!IND || b >= 0) {
                float a = N.bv2[p];
                for (int j = 0; j < P; j++) {
                    float h = N.bv1[j];
                    for (int w = 0; w < ks; w++) h += RED[(w * 16 + s) * 20 + j];
                    a += fmaxf(h, 0.f) * N.Wv2[j * P + p];
                }
                if constexpr (IND) __hip_atomic_store((uint32_t*)v_out + (size_t)b * P + p, __float_as_uint(tanhf(a)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v_out[(size_t)b * P + p] = tanhf(a);
            }
        }
    }
}
#undef N

template <class CF, bool H2 = false>
__global__ __launch_bounds__(768) void k_mb1d_net(Mb1dNetW N /* first argument: offset 0 of the kernel argument segment, read through it */,
                                                  const int8_t* __restrict__ boards,
                                                  const uint8_t* __restrict__ valid, int B, float* __restrict__ pi_out,
                                                  float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)N;
    mb1d_net_body<CF, H2, false>(smem, (Mb1dNetWC)__builtin_amdgcn_kernarg_segment_ptr(), boards, valid, B, pi_out, v_out, (int)blockIdx.x);
}

#pragma clang fp contract(off)

// the supported geometries (azg.h AZG_NET_*)
//                        L   C  NS    A  P   E0   E1   E2  Q0  Q1  Q2 CO1 A0 A12 PMAX
typedef Mb1dCfg<7, 56, 8, 81, 2, 168, 168, 168, 40, 40, 40, 56, 1, 2, 1> CfgSplendor2;
typedef Mb1dCfg<7, 71, 8, 81, 3, 213, 213, 213, 56, 56, 56, 71, 1, 2, 1> CfgSplendor3;
typedef Mb1dCfg<7, 88, 8, 81, 4, 264, 264, 264, 64, 64, 64, 88, 1, 2, 1> CfgSplendor4;
typedef Mb1dCfg<6, 23, 16, 180, 2, 115, 115, 46, 32, 32, 16, 46, 1, 2, 0> CfgAzul;
// two more games whose shipped checkpoints are nets of this family (MinivillesNNet.py:101-123 nn_version 82, 2 players: [58][2] board;
// TLPNNet.py:175-196 nn_version 83, 3 players: [55][15] board)
typedef Mb1dCfg<2, 58, 16, 21, 2, 174, 174, 174, 40, 40, 40, 58, 1, 2, 1> CfgMinivilles2;
typedef Mb1dCfg<15, 55, 8, 9, 3, 82, 82, 82, 24, 24, 24, 55, 1, 2, 1> CfgTLP3;

}  // namespace azg
