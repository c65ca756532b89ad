// nn_v80_h2.hip.h -- the V80 policy/value forward (splendor/SplendorNNet.py:262-283,397-440; NeuralNet.predict for a leaf batch,
// GenericNNetWrapper.py:94-120) designed around two facts of gfx950:
//
//  1. f32-input MFMA runs at 1/16 of the 16-bit rate.  Every f32 operand (weight or activation) is carried as TWO f16 numbers,
//     hi = rn16(x) and lo = rn16(x - hi): 22 significant bits.  A product is three v_mfma_f32_16x16x32_f16 (lo*hi, hi*lo, hi*hi;
//     lo*lo is 2^-22 of the product) accumulated in f32.  Weights are pre-scaled by a power of two per matrix so that their lo
//     parts stay normal f16 numbers, activations by 2^6; the epilogue multiplies the accumulator by the exact inverse.
//     An activation tile lives in LDS as two f16 planes (hi, lo) -- the footprint of the f32 tile -- written split once by the
//     epilogue that produces it.
//  2. The block's depthwise Linear(7 -> 7) mixes the 7 TOKENS of one (sample, channel) (SplendorNNet.py:148-187).  With the
//     tile's rows ordered TOKEN-MAJOR (row = token * 16 + sample) a 16-row MFMA tile is one token of the 16 samples, so the wave that
//     owns a 16-channel column tile of the expand GEMM ends up with all 7 tokens of (sample = lane & 15, 4 channels) in ITS OWN
//     accumulator registers: the depthwise layer, its BN + activation and the squeeze (mean / max over tokens) are per-lane
//     register arithmetic on MFMA results, the SE scale comes out of the fc2 MFMA in the same lane layout, and the expanded
//     activations are written to LDS once -- already scaled -- as the project GEMM's operand.
//
// One workgroup = 16 samples, 12 waves; one pass = first layer + trunk block + policy block/head + value block/head; nothing but
// boards, valid masks, weights and pi / v crosses HBM.  Five workgroup barriers per block.
#pragma once
#include "nn_kernels.hip.h"

#pragma clang fp contract(on)

namespace azg {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

constexpr float H2_AS = 64.f, H2_IAS = 1.f / 64.f;        // activation planes hold 64 * x

// device pointer tables (weights in h2 fragment order: [N/16 col tiles][K/32 chunks][2 planes hi, lo][64 lanes] x 16 bytes,
// halves j = 0..7 of lane = W_plane[32*chunk + 8*(lane>>4) + j][16*tile + (lane&15)] * 2^k; s* = 2^-k / 64 undoes both scales)
struct H2BlockW {
    const uint4 *We, *W1, *W2, *Wp;                       // [64][176], [192][48], [64][176], [192][64] (zero padded)
    const float *be, *Wd, *sd, *bd, *b1, *b2, *bp;        // [176], [49] (Hardswish blocks: Wd / 6), [176], [176], [48], [176], [64]
    float se, s1, s2, sp;
};
struct H2NetW {
    const uint4 *W0, *Wpi1, *Wpi2, *Wv1;                  // [64][64], [448][96] (k = token*64 + c), [96][96], [448][16]
    const float *b0, *bpi1, *bpi2, *bv1, *Wv2, *bv2;      // [64], [96], [96], [16], [P][P] plain, [P]
    float s0, spi1, spi2, sv1;
};

// the 43 pointer slots + 16 descale factors of azg_nn_v80_forward_h2 (include/azg.h) -> the kernel's argument structs (host side)
struct H2Weights { H2BlockW Wt, Wp, Wv; H2NetW N; };
static inline H2Weights h2_weights(const void* const* w, const float* descale) {
    auto blk = [&](int o, int d) {
        return H2BlockW{(const uint4*)w[o], (const uint4*)w[o + 5], (const uint4*)w[o + 7], (const uint4*)w[o + 9],
                        (const float*)w[o + 1], (const float*)w[o + 2], (const float*)w[o + 3], (const float*)w[o + 4],
                        (const float*)w[o + 6], (const float*)w[o + 8], (const float*)w[o + 10],
                        descale[d], descale[d + 1], descale[d + 2], descale[d + 3]};
    };
    return H2Weights{blk(2, 1), blk(13, 5), blk(24, 9),
                     H2NetW{(const uint4*)w[0], (const uint4*)w[35], (const uint4*)w[37], (const uint4*)w[39],
                            (const float*)w[1], (const float*)w[36], (const float*)w[38], (const float*)w[40], (const float*)w[41],
                            (const float*)w[42], descale[0], descale[13], descale[14], descale[15]}};
}

// The kernel's ~116 dwords of pointers and scales do not fit the scalar registers beside what a block needs, and the compiler spilled
// them to VGPR lanes (v_writelane / v_readlane are VALU instructions, in phases that are VALU-bound).  The forward therefore takes ONE
// pointer to the H2Weights in the constant address space (the kernel's own argument segment, or the round arguments of the per-CU
// kernel) and every block loads what it uses with scalar loads where it uses it, through a pointer the compiler cannot see through
// (so that nothing is kept live from one block to the next).
typedef const H2Weights __attribute__((address_space(4))) * H2WeightsC;
template <class T>
__device__ __forceinline__ T h2_load_const(const T __attribute__((address_space(4))) * p) {
    static_assert(sizeof(T) % 4 == 0, "word-sized struct");
    const uint32_t __attribute__((address_space(4))) * s = (const uint32_t __attribute__((address_space(4))) *)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = s[k];      // scalar loads; the words a block does not use are dropped
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}
__device__ __forceinline__ H2WeightsC h2_opaque(H2WeightsC p) {
    uint64_t v = (uint64_t)(uintptr_t)p;
    asm volatile("" : "+s"(v));
    return (H2WeightsC)(uintptr_t)v;
}

// byte offset of (row, 16-byte chunk q = 8 halves) in a plane of row stride RS (RS = 128 mod 256: with the XOR the ds_read_b128
// lane groups of an MFMA operand fetch -- 16 rows x 4 adjacent chunks -- hit 16 distinct 16-byte bank columns)
__device__ __forceinline__ int h2_off(int row, int q, int RS) { return row * RS + ((q ^ (row & 7)) << 4); }

// Activation range of the f16 x 2 kernels.  The planes hold 64 * x as f16, so |x| >= 1023.5 does not fit.  Every h2 kernel sets the
// FP16_OVFL bit of its waves' MODE register (bit 23): an f16 result that overflows is CLAMPED to +-65504 instead of becoming inf, at no
// instruction cost, so an out-of-range activation saturates (hi = 65504, lo = the clamped remainder: up to ~2047 is still carried) and
// can never turn into inf - inf = NaN in the lo part or inf * 0 = NaN against a zero-padded weight.  The f32-operand kernels
// (h2 = False) have the full f32 range; include/azg.h states the contract next to the 1e-5 one.
__device__ __forceinline__ void h2_fp16_saturate_mode() {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // hwreg(HW_REG_MODE, 23, 1) = 1
}

__device__ __forceinline__ void h2_split2(float a, float b, uint32_t& h, uint32_t& l) {
    const f16x2_t hh = __builtin_convertvector(f32x2{a, b}, f16x2_t);            // v_cvt_pk_f16_f32 (round to nearest even)
    h = __builtin_bit_cast(uint32_t, hh);
#ifndef AZG_H2_SPLIT_CVT
    // lo = rn16(x - hi) as ONE mixed-precision FMA per value (f16 source widened on the fly, f16 result written to one half of the
    // destination): three instructions per pair instead of five (two v_cvt_f32_f16, a packed subtract, a v_cvt_pk_f16_f32); x - hi is
    // exact in f32 either way, so the bits are the same.  (The compiler turns the C expression back into the five-instruction form.)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
#else
    const f32x2 back = __builtin_convertvector(hh, f32x2);
    const f16x2_t ll = __builtin_convertvector(f32x2{a - back.x, b - back.y}, f16x2_t);
    l = __builtin_bit_cast(uint32_t, ll);
#endif
}
// four consecutive channels (ch0 % 4 == 0) of one row -> both planes (PD = byte distance hi plane -> lo plane); o is UNSCALED
__device__ __forceinline__ void h2_store4(uint8_t* hi, int PD, int RS, int row, int ch0, f32x4 o, float mul = H2_AS) {
    o = o * mul;
    uint32_t h0, l0, h1, l1;
    h2_split2(o[0], o[1], h0, l0);
    h2_split2(o[2], o[3], h1, l1);
    uint8_t* dst = hi + h2_off(row, ch0 >> 3, RS) + ((ch0 & 4) << 1);
    *(uint2*)dst = make_uint2(h0, h1);
    *(uint2*)(dst + PD) = make_uint2(l0, l1);
}
// hi + lo of a packed pair of f16 x 2 values as f32: one mixed-precision FMA per value (f32(hi) * 1 + f32(lo); was two conversions and
// half a packed add per value; hi + lo is exact in f32 either way)
__device__ __forceinline__ f32x2 h2_join2(uint32_t h, uint32_t l) {
#ifndef AZG_H2_SPLIT_CVT
    f32x2 r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r.x) : "v"(h), "v"(l));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r.y) : "v"(h), "v"(l));
    return r;
#else
    return __builtin_convertvector(__builtin_bit_cast(f16x2_t, h), f32x2) + __builtin_convertvector(__builtin_bit_cast(f16x2_t, l), f32x2);
#endif
}
__device__ __forceinline__ f32x4 h2_load4(const uint8_t* hi, int PD, int RS, int row, int ch0, float mul = H2_IAS) {
    const uint8_t* src = hi + h2_off(row, ch0 >> 3, RS) + ((ch0 & 4) << 1);
    const uint2 h = *(const uint2*)src, l = *(const uint2*)(src + PD);
    const f32x2 a = h2_join2(h.x, l.x), b = h2_join2(h.y, l.y);
    return f32x4{a.x, a.y, b.x, b.y} * mul;
}
// acc += W * A for one K chunk of 32: the three products of relative weight >= 2^-11, smallest first
__device__ __forceinline__ f32x4 h2_mma(uint4 wh, uint4 wl, uint4 ah, uint4 al, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, ah), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, al), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, ah), acc, 0, 0, 0);
    return acc;
}
// weight fragment (tile nt, chunk c, plane p) of a matrix with NCH chunks
// (every weight / bias request goes through an explicitly GLOBAL pointer: the pointers come out of a word-wise copy from the constant
// address space and are generic to the compiler, which made every request a flat_load -- counted against the LDS counter too)
typedef uint32_t h2_u32x4 __attribute__((ext_vector_type(4)));
#define H2G(T, p) ((const __attribute__((address_space(1))) T*)(p))
#define H2FRAG(ptr, NCH, nt, c, p) __builtin_bit_cast(uint4, H2G(h2_u32x4, ptr)[((((size_t)(nt) * (NCH) + (c)) * 2 + (p)) << 6) + lane])

// ReLU, or 6 * Hardswish = x * clamp(x + 3, 0, 6): the 1/6 is folded into the next linear step (the token-mix weights the host
// passes for a Hardswish block are Wd / 6; the pooled value and the project operand take it with their plane scale)
__device__ __forceinline__ f32x4 h2_act6(f32x4 x, int act) {
    if (act == ACT_RELU) return f32x4{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f), fmaxf(x[2], 0.f), fmaxf(x[3], 0.f)};
    const f32x4 t = x + 3.f;
    return x * f32x4{__builtin_amdgcn_fmed3f(t[0], 0.f, 6.f), __builtin_amdgcn_fmed3f(t[1], 0.f, 6.f), __builtin_amdgcn_fmed3f(t[2], 0.f, 6.f),
                     __builtin_amdgcn_fmed3f(t[3], 0.f, 6.f)};
}

// 6 * Hardswish(x) from x6 = 6 x on the packed-f32 pipe: 6 x * clamp(x / 6 + 1 / 2, 0, 1) = x6 * clamp(x6 / 36 + 1 / 2): one v_pk_fma_f32 with
// the CLAMP output modifier + one v_pk_mul_f32 per two values, where x * med3(x + 3, 0, 6) takes a packed add, two v_med3_f32 and a packed
// multiply (the compiler does not fold a clamp into a packed FMA: inline asm).  The factor 6 on the input is free: it goes into the
// scale and bias of the FMA that produces x.  (Rounding differs from the med3 form in the last bit of the clamp argument.)
__device__ __forceinline__ f32x2 h2_hswish6_pk(f32x2 x6) {
    f32x2 t;
    const f32x2 k = f32x2{1.f / 36.f, 1.f / 36.f};
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,0,0] clamp" : "=v"(t) : "v"(x6), "s"(k));
    return x6 * t;
}
// act(acc * scale + bias) for the block's activation; Hardswish blocks return 6 * Hardswish (see h2_act6)
template <int ACT>
__device__ __forceinline__ f32x4 h2_scale_bias_act6(f32x4 acc, f32x4 scale, f32x4 bias) {
#ifndef AZG_H2_MED3_HSWISH
    if (ACT == ACT_HSWISH) {
        const f32x4 x6 = acc * (scale * 6.f) + bias * 6.f;           // (the two constant products are hoisted out of the token loops)
        const f32x2 a = h2_hswish6_pk(f32x2{x6[0], x6[1]}), b = h2_hswish6_pk(f32x2{x6[2], x6[3]});
        return f32x4{a[0], a[1], b[0], b[1]};
    }
#endif
    return h2_act6(acc * scale + bias, ACT);
}

// phase-E operands of one block for this lane (column tile nt of the expand GEMM): requested one phase ahead -- during the
// previous block's project GEMM -- so that a block never starts by waiting for its first weights
struct H2EW {
    uint4 weh[2], wel[2];
    f32x4 be4, sd4, bd4;
    float wdv;                                            // the 7x7 token mix: element `lane` (rows move to SGPRs as they are used)
};
__device__ __forceinline__ void h2_load_ew(H2EW& e, const H2BlockW& W, int nt, int g, int lane) {
#pragma unroll
    for (int c = 0; c < 2; c++) { e.weh[c] = H2FRAG(W.We, 2, nt, c, 0); e.wel[c] = H2FRAG(W.We, 2, nt, c, 1); }
    const int ch0 = nt * 16 + 4 * g;
    e.be4 = *H2G(f32x4, W.be + ch0); e.sd4 = *H2G(f32x4, W.sd + ch0); e.bd4 = *H2G(f32x4, W.bd + ch0);
    e.wdv = H2G(float, W.Wd)[lane < 49 ? lane : 0];
}

// LDS map (bytes).  X: the tile the three blocks read; O: a head block's output (and the int8 board tile before the first layer);
// H: the SE-scaled expanded activations = the project operand (the head tails put their small buffers over it afterwards)
constexpr int H2_RSX = 128, H2_RSH = 384;
constexpr int H2_XH = 0, H2_XL = 14336, H2_OH = 28672, H2_OL = 43008, H2_HH = 57344, H2_HL = 100352, H2_PLH = 143360, H2_PLL = 149504,
              H2_SHH = 155648, H2_SHL = 157696, H2_LDS = 159744;
constexpr int H2_PDX = H2_XL - H2_XH, H2_PDH = H2_HL - H2_HH, H2_PDP = H2_PLL - H2_PLH, H2_PDS = H2_SHL - H2_SHH;
// head tails (inside H): RED partial sums, HID planes (row stride 384), LG logits
constexpr int H2_RED = H2_HH, H2_HIDH = H2_HH + 12288, H2_HIDL = H2_HIDH + 6144, H2_LG = H2_HIDL + 6144, H2_LS = 100;

#ifdef AZG_NN_PHASE_TIMES
static __device__ long long g_h2_phase[4][16];
#define H2_PH(k) do { if (blockIdx.x == 7 && threadIdx.x == 0) g_h2_phase[MODE][k] = clock64(); } while (0)
#else
#define H2_PH(k)
#endif

// One InvertedResidual1d block (SplendorNNet.py:189-202) on the tile in the X planes.  MODE 1: trunk (output replaces X);
// MODE 2 / 3: policy / value head block (output -> O planes, X stays for the other head) followed by the head's tail.
// NW = waves of the workgroup: 12 (the stand-alone kernel, 168 VGPRs, 3 waves per SIMD) or 16 (128 VGPRs, 4 waves per SIMD -- the budget
// of a workgroup that also runs 16 tree descents, kernels.hip.h k_rounds): with 16 waves the weights of a later phase are requested
// later (LEAN: fewer fragment registers in flight), the project GEMM and the first layer run over four row-tile groups instead of three.
// IND (the asynchronous pipeline, azg_async.hip.h): sample s of the workgroup is tree sidx[s] (LDS; < 0 = no sample) -- `valid` is then the
// pipeline's leaf-record array (its valid BIT mask, kernels.hip.h AsyncLeaf<SplendorDev<2>>: stride 416, mask at byte 400), read past the L1,
// and pi / v rows are written WRITE-THROUGH at the tree's index: the reader is a descent wave on another CU, inside the same launch.
constexpr int H2_AL_STRIDE = 416, H2_AL_MASK = 400;
constexpr int H2_IND_MASK = 52;               // IND: the samples' valid bit masks (u64 [16][2]) sit 52 ints behind sidx in LDS -- fetched with the
                                              // board tile at the start of the forward, so that the softmax does not wait for HBM
template <int ACT, int POOLMAX, int MODE, int NW, bool IND = false>
__device__ __forceinline__ void h2_block(uint8_t* lds, const H2BlockW& W, const H2NetW& N, int B, int P,
                                         const uint8_t* __restrict__ valid, float* __restrict__ pi_out, float* __restrict__ v_out,
                                         H2EW& ew /* in: this block's phase-E operands; out: the next block's */,
                                         const H2BlockW& Wnext, int wg, const int tid, const int* sidx = nullptr) {
    constexpr int NS = 16, A = 81;
    constexpr bool LEAN = NW > 12;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int b0 = wg * NS;
    uint8_t* const XH = lds + H2_XH;
    uint8_t* const OH = lds + H2_OH;
    uint8_t* const HH = lds + H2_HH;
    uint8_t* const PLH = lds + H2_PLH;
    uint8_t* const SHH = lds + H2_SHH;
    H2_PH(0);

    // ---- weights of the SE phases: requested now, behind the phase-E operands `ew` that are already on their way ----
    const int nt = wave < 11 ? wave : 0;
    uint4 w2h[2], w2l[2];
    const int ch0 = nt * 16 + 4 * g;                          // this lane's 4 expanded channels
    const uint4 weh0 = ew.weh[0], weh1 = ew.weh[1], wel0 = ew.wel[0], wel1 = ew.wel[1];
    const f32x4 be4 = ew.be4, sd4 = ew.sd4, bd4 = ew.bd4;
    const float wdv = ew.wdv;
    // SE fc1 (K = 192 = 6 chunks, 3 column tiles) = 9 units (column tile u % 3, chunk pair u / 3) whose partial sums are added up through
    // LDS.  12-wave workgroup: waves 0..8 take one unit each (a 3-wave fc1 would hold 48 registers of fragments per wave through phase
    // E).  16-wave workgroup (128 VGPRs): the five waves that sit out phase E (11..15) take units w - 11 and w - 6 -- the waves that
    // carry the token-mix results through the SE phases then never hold fc1 fragments (the two roles are separate code paths with the
    // same number of barriers, so the register allocator never sees their fragments live together).
    const int cs1 = tid / 12, cc1 = 4 * (tid - cs1 * 12);     // the fc1 combine step: (sample, 4 hidden units) of thread tid < 192
    f32x4 b24, b14;
    // the H planes' pad columns 176..191 (chunks 22, 23) must read as zeros; a head tail may have left its buffers there
    if (tid < 448) {
        const int row = tid >> 2, q = 22 + (tid & 1), pl = (tid >> 1) & 1;
        *(uint4*)(HH + pl * H2_PDH + h2_off(row, q, H2_RSH)) = make_uint4(0u, 0u, 0u, 0u);
    }
    float* const RED1 = (float*)(lds + H2_OH);                // [3 K groups][16][48] f32: the O planes are free until phase P
    constexpr int PG = NW / 4;                                // row-tile groups of the project GEMM: 4 column tiles x PG waves (1: every weight
                                                              // fragment enters the CU once; the four waves sit on the four SIMDs)
    const int ntp = wave & 3, rt0 = wave >> 2;
    uint4 wph[6], wpl[6];
    f32x4 bp4;
    f32x4 dw[7];

    // ---- E: expand GEMM (+BN+act) -> depthwise token mix (+BN+act) -> squeeze, all in this wave's registers (waves 0..10) ----
    auto phase_e = [&](uint4 (&w1h)[2], uint4 (&w1l)[2], int nt1, int kg1) {
        f32x4 in[7];
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int row = t * 16 + r;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                const int o0 = h2_off(row, g, H2_RSX), o1 = h2_off(row, 4 + g, H2_RSX);
                acc = h2_mma(weh0, wel0, *(const uint4*)(XH + o0), *(const uint4*)(XH + H2_PDX + o0), acc);
                acc = h2_mma(weh1, wel1, *(const uint4*)(XH + o1), *(const uint4*)(XH + H2_PDX + o1), acc);
            }
            in[t] = h2_scale_bias_act6<ACT>(acc, f32x4{W.se, W.se, W.se, W.se}, be4);
#ifndef AZG_H2_NO_EPRIO
            // the three expand waves of a SIMD share its VALU issue (the phase lasts as long as their instructions take one after the other, and
            // the oldest wave used to run ahead: done after 2.8 k cycles of a 7.2 k phase): a wave's priority falls as it advances through the
            // phase, so that the three stay in step and cover each other's MFMA -> VALU and LDS latencies: 75.6 -> 76.3 k env-steps/s
            if (t == 0) __builtin_amdgcn_s_setprio(3);
            if (t == 3) __builtin_amdgcn_s_setprio(2);
            if (t == 6) __builtin_amdgcn_s_setprio(1);
#endif
            if (t == 0) H2_PH(10);
        }
        H2_PH(11);
        // the SE weights: requested behind the expand GEMM's operands (the vector-memory pipe of the CU takes ~30 cycles per 1 KiB
        // wave request when every wave is asking), they land during the token mix
        if (!LEAN) {
#pragma unroll
            for (int c = 0; c < 2; c++) { w1h[c] = H2FRAG(W.W1, 6, nt1, 2 * kg1 + c, 0); w1l[c] = H2FRAG(W.W1, 6, nt1, 2 * kg1 + c, 1); }
#pragma unroll
            for (int c = 0; c < 2; c++) { w2h[c] = H2FRAG(W.W2, 2, nt, c, 0); w2l[c] = H2FRAG(W.W2, 2, nt, c, 1); }
            b24 = *H2G(f32x4, W.b2 + ch0);
            b14 = *H2G(f32x4, W.b1 + (tid < 192 ? cc1 : 0));
        }
        f32x4 pool = POOLMAX ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef AZG_H2_WD_READLANE
        // the 7 x 7 token-mix matrix through the scalar data cache: row m + 1 is requested (7 dwords, constant address space) before
        // row m's 28 packed FMAs are issued; no v_readlane (AZG_H2_WD_READLANE: the matrix in a VGPR, element m * 7 + l read with
        // v_readlane -- 147 more VALU instructions per wave, the same time within 0.3 %)
        typedef const float __attribute__((address_space(4))) * cfp;
        float wdn[7];
        {
            const float* p0 = W.Wd;
            asm volatile("" : "+s"(p0));
            const cfp q0 = (cfp)(uintptr_t)p0;
#pragma unroll
            for (int l = 0; l < 7; l++) wdn[l] = q0[l];
        }
#endif
#pragma unroll
        for (int m = 0; m < 7; m++) {
            float wd[7];
#ifndef AZG_H2_WD_READLANE
#pragma unroll
            for (int l = 0; l < 7; l++) wd[l] = wdn[l];
            if (m < 6) {
                const float* p1 = W.Wd + 7 * (m + 1);
                asm volatile("" : "+s"(p1));
                const cfp q1 = (cfp)(uintptr_t)p1;
#pragma unroll
                for (int l = 0; l < 7; l++) wdn[l] = q1[l];
            }
#else
#ifndef AZG_H2_WD_HOIST
            // (opaque per output token: hoisted above the expand GEMM, the 49 v_readlane results of a block do not fit the scalar registers
            // beside the kernel's 116 dwords of arguments -- the compiler then spilled them and 36 argument registers to VGPR lanes and
            // read everything back: 205 lane instructions per block and wave instead of 49)
            float wdm = wdv;
            asm volatile("" : "+v"(wdm));
#else
            const float wdm = wdv;
#endif
#pragma unroll
            for (int l = 0; l < 7; l++) wd[l] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wdm), m * 7 + l));
#endif
            f32x4 a = wd[0] * in[0];
#pragma unroll
            for (int l = 1; l < 7; l++) a += wd[l] * in[l];
            a = h2_scale_bias_act6<ACT>(a, sd4, bd4);
#ifndef AZG_H2_NO_EPRIO
            if (m == 3) __builtin_amdgcn_s_setprio(0);
#endif
            dw[m] = a;
            if (POOLMAX) pool = f32x4{fmaxf(pool[0], a[0]), fmaxf(pool[1], a[1]), fmaxf(pool[2], a[2]), fmaxf(pool[3], a[3])};
            else pool += a;
        }
        H2_PH(12);
        if (LEAN) {          // 128-VGPR budget: the fc2 weights only once the token mix has released its 28 input registers
#pragma unroll
            for (int c = 0; c < 2; c++) { w2h[c] = H2FRAG(W.W2, 2, nt, c, 0); w2l[c] = H2FRAG(W.W2, 2, nt, c, 1); }
            b24 = *H2G(f32x4, W.b2 + ch0);
        }
        constexpr float ACT_DIV = ACT == ACT_HSWISH ? 6.f : 1.f;                 // dw holds 6 * Hardswish(.)
        h2_store4(PLH, H2_PDP, H2_RSH, r, ch0, pool, H2_AS / ACT_DIV / (POOLMAX ? 1.f : 7.f));
    };
    auto load_wp = [&]() {
#pragma unroll
        for (int c = 0; c < 6; c++) { wph[c] = H2FRAG(W.Wp, 6, ntp, c, 0); wpl[c] = H2FRAG(W.Wp, 6, ntp, c, 1); }
        bp4 = *H2G(f32x4, W.bp + ntp * 16 + 4 * g);
    };
    // one fc1 unit (column tile nt1, chunk pair kg1): partial sums -> RED1
    auto s1_unit = [&](const uint4 (&w1h)[2], const uint4 (&w1l)[2], int nt1, int kg1) {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const int o0 = h2_off(r, 8 * kg1 + g, H2_RSH), o1 = h2_off(r, 8 * kg1 + 4 + g, H2_RSH);
        a0 = h2_mma(w1h[0], w1l[0], *(const uint4*)(PLH + o0), *(const uint4*)(PLH + H2_PDP + o0), a0);
        a1 = h2_mma(w1h[1], w1l[1], *(const uint4*)(PLH + o1), *(const uint4*)(PLH + H2_PDP + o1), a1);
        *(f32x4*)(RED1 + (kg1 * 16 + r) * 48 + nt1 * 16 + 4 * g) = a0 + a1;
    };
    auto s1_combine = [&]() {                                 // bias + ReLU -> SH (threads 0..191)
        const int s = cs1, col = cc1;
        const f32x4 hv = (*(const f32x4*)(RED1 + s * 48 + col) + *(const f32x4*)(RED1 + (16 + s) * 48 + col) + *(const f32x4*)(RED1 + (32 + s) * 48 + col)) * W.s1 + b14;
        h2_store4(SHH, H2_PDS, H2_RSX, s, col, f32x4{fmaxf(hv[0], 0.f), fmaxf(hv[1], 0.f), fmaxf(hv[2], 0.f), fmaxf(hv[3], 0.f)});
    };
    // S2: SE fc2 + Hardsigmoid: the scale of (sample r, this lane's 4 channels) lands in this lane -> H = 64 * dw * scale
    auto phase_s2 = [&]() {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int off = h2_off(r, 4 * c + g, H2_RSX);
            acc = h2_mma(w2h[c], w2l[c], *(const uint4*)(SHH + off), *(const uint4*)(SHH + H2_PDS + off), acc);
        }
        const f32x4 y = acc * W.s2 + b24;
        const f32x4 sc = f32x4{hardsigmoid(y[0]), hardsigmoid(y[1]), hardsigmoid(y[2]), hardsigmoid(y[3])};
        const f32x4 scm = sc * (H2_AS / (ACT == ACT_HSWISH ? 6.f : 1.f));                   // (the plane scale goes into the SE scale once)
#pragma unroll
        for (int t = 0; t < 7; t++) {
#ifndef AZG_H2_NO_EPRIO          /* (the same ladder through the seven split-and-store steps: +0.2 %) */
            if (t == 0) __builtin_amdgcn_s_setprio(2);
            if (t == 3) __builtin_amdgcn_s_setprio(1);
            if (t == 6) __builtin_amdgcn_s_setprio(0);
#endif
            h2_store4(HH, H2_PDH, H2_RSH, t * 16 + r, ch0, dw[t] * scm, 1.f);
        }
    };

    if (!LEAN) {
        uint4 w1h[2], w1l[2];
        const int nt1 = wave % 3, kg1 = wave < 9 ? wave / 3 : 0;
        if (wave < 11) phase_e(w1h, w1l, nt1, kg1);
        // project weights: requested when the phase-E arithmetic is done, they land during the SE phases
        if (wave < 4 * PG) load_wp();
        else bp4 = *H2G(f32x4, W.bp + ntp * 16 + 4 * g);
        __syncthreads();
        H2_PH(1);
        if (wave < 9) s1_unit(w1h, w1l, nt1, kg1);           // ---- S1: SE fc1 partial sums (9 waves) -> RED1, then bias + ReLU -> SH ----
        H2_PH(13);
        H2_PH(14);
        __syncthreads();
        H2_PH(15);
        if (tid < 192) s1_combine();
        __syncthreads();
        H2_PH(2);
        if (wave < 11) phase_s2();
        __syncthreads();
        H2_PH(3);
    } else if (wave < 11) {
        // role A (16-wave workgroup): the eleven column-tile waves -- E, [S1 runs elsewhere], combine (waves 0..2), S2
        uint4 none_h[2], none_l[2];
        if (tid < 192) b14 = *H2G(f32x4, W.b1 + cc1);
        phase_e(none_h, none_l, 0, 0);
        __syncthreads();
        H2_PH(1);
        __syncthreads();
        H2_PH(15);
        if (tid < 192) s1_combine();
        __syncthreads();
        H2_PH(2);
        load_wp();                                            // lands during S2 (the fc2 MFMAs and the seven split-and-store steps)
        phase_s2();
        __syncthreads();
        H2_PH(3);
    } else {
        // role B: the five waves that sit out phase E hold the fc1 fragments (units w - 11 and w - 6 of the nine) and run S1
        uint4 w1h[2][2], w1l[2][2];
        const int u0 = wave - 11, u1 = wave - 6;              // u1 < 9 for waves 11..14
#pragma unroll
        for (int c = 0; c < 2; c++) { w1h[0][c] = H2FRAG(W.W1, 6, u0 % 3, 2 * (u0 / 3) + c, 0); w1l[0][c] = H2FRAG(W.W1, 6, u0 % 3, 2 * (u0 / 3) + c, 1); }
        if (u1 < 9) {
#pragma unroll
            for (int c = 0; c < 2; c++) { w1h[1][c] = H2FRAG(W.W1, 6, u1 % 3, 2 * (u1 / 3) + c, 0); w1l[1][c] = H2FRAG(W.W1, 6, u1 % 3, 2 * (u1 / 3) + c, 1); }
        }
        __syncthreads();
        s1_unit(w1h[0], w1l[0], u0 % 3, u0 / 3);
        if (u1 < 9) s1_unit(w1h[1], w1l[1], u1 % 3, u1 / 3);
        __syncthreads();
        __syncthreads();
        load_wp();
        __syncthreads();
    }

    // the next block's phase-E operands and the head tail's first fragments stream in while the project GEMM runs
    if (MODE != 2) h2_load_ew(ew, Wnext, nt, g, lane);        // (policy block: after the tail's first GEMM, whose 56 fragment registers come first)
    uint4 wfh[7], wfl[7], wgh[3], wgl[3];
    const int ht_nt = wave % 6, ht_half = wave / 6;
    f32x4 bt1 = f32x4{0.f, 0.f, 0.f, 0.f}, bt2 = bt1;         // the tails' biases: requested with the fragments
    const int ts = tid / 24, tcol = 4 * (tid - ts * 24);      // policy tail combine step: (sample, 4 hidden units) of thread tid < 384
    constexpr int PRE = LEAN ? 2 : 4;                         // fragments of the first policy Linear requested before the project GEMM
    const bool tailw = wave < 12;                             // the policy tail's first GEMM: 6 column tiles x 2 K halves = 12 waves
    if (MODE == 2) {
        bt1 = *H2G(f32x4, N.bpi1 + (tid < 384 ? tcol : 0));
        bt2 = *H2G(f32x4, N.bpi2 + (wave < 6 ? wave : 0) * 16 + 4 * g);
        if (tailw) {
#pragma unroll
            for (int cc = 0; cc < PRE; cc++) { wfh[cc] = H2FRAG(N.Wpi1, 14, ht_nt, ht_half * 7 + cc, 0); wfl[cc] = H2FRAG(N.Wpi1, 14, ht_nt, ht_half * 7 + cc, 1); }
        }
    }
    if (MODE == 3 && tailw) {                                 // (twelve K groups whatever NW is: the summation order is part of the result)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const int c = wave + 12 * cc;
            wfh[cc] = c < 14 ? H2FRAG(N.Wv1, 14, 0, c, 0) : make_uint4(0u, 0u, 0u, 0u);
            wfl[cc] = c < 14 ? H2FRAG(N.Wv1, 14, 0, c, 1) : make_uint4(0u, 0u, 0u, 0u);
        }
    }

    // ---- P: project GEMM + BN + residual -> X (trunk, in place: a lane rewrites the cells it read) or O (heads) ----
    if (wave < 4 * PG) {
        // (seven row tiles over PG = 3 waves of a SIMD; raising the issue priority of the wave that has three of them: 74.5 -> 74.3 k,
        // a priority that falls tile by tile: 76.3 -> 75.6 k -- both dropped)
#pragma unroll 1
        for (int rt = rt0; rt < 7; rt += PG) {
            const int row = rt * 16 + r;
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int c = 0; c < 6; c += 2) {
                const int o0 = h2_off(row, 4 * c + g, H2_RSH), o1 = h2_off(row, 4 * c + 4 + g, H2_RSH);
                a0 = h2_mma(wph[c], wpl[c], *(const uint4*)(HH + o0), *(const uint4*)(HH + H2_PDH + o0), a0);
                a1 = h2_mma(wph[c + 1], wpl[c + 1], *(const uint4*)(HH + o1), *(const uint4*)(HH + H2_PDH + o1), a1);
            }
            const int col0 = ntp * 16 + 4 * g;
            // (in the planes' units: 64 * sp, 64 * bias, the residual as stored -- the bits of ((a0 + a1) * sp + bias + x) * 64)
            const f32x4 o4 = (a0 + a1) * (W.sp * H2_AS) + bp4 * H2_AS + h2_load4(XH, H2_PDX, H2_RSX, row, col0, 1.f);
            h2_store4(MODE == 1 ? XH : OH, H2_PDX, H2_RSX, row, col0, o4, 1.f);  // (columns 56..63: zero weights + zero bias + zero x)
        }
    }
    if (MODE == 2 && tailw) {                                 // the rest of the first policy Linear's fragments (the project fragments are dead)
#pragma unroll
        for (int cc = PRE; cc < 7; cc++) { wfh[cc] = H2FRAG(N.Wpi1, 14, ht_nt, ht_half * 7 + cc, 0); wfl[cc] = H2FRAG(N.Wpi1, 14, ht_nt, ht_half * 7 + cc, 1); }
    }
    if (MODE == 2 && wave < 6) {                              // second policy Linear
#pragma unroll
        for (int c = 0; c < 3; c++) { wgh[c] = H2FRAG(N.Wpi2, 3, wave, c, 0); wgl[c] = H2FRAG(N.Wpi2, 3, wave, c, 1); }
    }
    __syncthreads();
    H2_PH(4);

    if (MODE == 2) {
        // ---- policy tail: Flatten -> Linear(392, 81) + ReLU -> Linear(81, 81) -> masked softmax ----
        float* RED = (float*)(lds + H2_RED);                  // [2 K halves][16][96]
        float* LG = (float*)(lds + H2_LG);                    // [16][H2_LS]
        uint8_t* const HIDH = lds + H2_HIDH;
        if (tailw) {
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int cc = 0; cc < 7; cc++) {
                const int c = ht_half * 7 + cc;               // K chunk of 32 = half a token row of the O planes
                const int off = h2_off((c >> 1) * 16 + r, 4 * (c & 1) + g, H2_RSX);
                if (cc & 1) a1 = h2_mma(wfh[cc], wfl[cc], *(const uint4*)(OH + off), *(const uint4*)(OH + H2_PDX + off), a1);
                else a0 = h2_mma(wfh[cc], wfl[cc], *(const uint4*)(OH + off), *(const uint4*)(OH + H2_PDX + off), a0);
            }
            *(f32x4*)(RED + (ht_half * 16 + r) * 96 + ht_nt * 16 + 4 * g) = a0 + a1;
        }
        h2_load_ew(ew, Wnext, nt, g, lane);
        __syncthreads();
        if (tid < 16 * 24) {
            const int s = ts, col = tcol;
            const f32x4 p = (*(const f32x4*)(RED + s * 96 + col) + *(const f32x4*)(RED + (16 + s) * 96 + col)) * N.spi1 + bt1;
            h2_store4(HIDH, H2_HIDL - H2_HIDH, H2_RSH, s, col, f32x4{fmaxf(p[0], 0.f), fmaxf(p[1], 0.f), fmaxf(p[2], 0.f), fmaxf(p[3], 0.f)});
        }
        __syncthreads();
        if (wave < 6) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int off = h2_off(r, 4 * c + g, H2_RSH);
                acc = h2_mma(wgh[c], wgl[c], *(const uint4*)(HIDH + off), *(const uint4*)(HIDH + (H2_HIDL - H2_HIDH) + off), acc);
            }
            *(f32x4*)(LG + r * H2_LS + wave * 16 + 4 * g) = acc * N.spi2 + bt2;
        }
        __syncthreads();
        H2_PH(5);
        // masked softmax == exp(log_softmax(where(valid, logits, -1e8))) (GenericNNetWrapper.py:105-107), one wave per sample
        for (int s = wave; s < NS; s += NW) {
            const int b = IND ? sidx[s] : b0 + s;
            if (IND ? b < 0 : b >= B) continue;
            const int a1i = lane + 64;
            bool ok0, ok1;
            if constexpr (IND) {
                const unsigned long long* mp = (const unsigned long long*)(sidx + H2_IND_MASK) + 2 * s;
                const unsigned long long m0 = mp[0], m1 = mp[1];
                ok0 = (m0 >> lane) & 1ull; ok1 = (m1 >> lane) & 1ull;
            } else {
                ok0 = valid[(size_t)b * A + lane] != 0; ok1 = a1i < A && valid[(size_t)b * A + a1i] != 0;
            }
            float x0 = ok0 ? LG[s * H2_LS + lane] : -1e8f;
            float x1 = a1i < A ? (ok1 ? LG[s * H2_LS + a1i] : -1e8f) : -INFINITY;
            const float mx = nn_wave_max(fmaxf(x0, x1));
            x0 = expf(x0 - mx);
            x1 = a1i < A ? expf(x1 - mx) : 0.f;
            const float sum = nn_wave_sum(x0 + x1);
            if constexpr (IND) {
                __hip_atomic_store((uint32_t*)pi_out + (size_t)b * A + lane, __float_as_uint(x0 / sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a1i < A) __hip_atomic_store((uint32_t*)pi_out + (size_t)b * A + a1i, __float_as_uint(x1 / sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                pi_out[(size_t)b * A + lane] = x0 / sum;
                if (a1i < A) pi_out[(size_t)b * A + a1i] = x1 / sum;
            }
        }
        __syncthreads();                                      // the value block re-zeroes H's pad columns under LG / HID
        H2_PH(6);
    }

    if (MODE == 3) {
        // ---- value tail: Flatten -> Linear(392, P) + ReLU -> Linear(P, P) -> tanh ----
        float* RED = (float*)(lds + H2_RED);                  // [12 waves][16][16]
        if (tailw) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const int c = wave + 12 * cc;
                if (c < 14) {
                    const int off = h2_off((c >> 1) * 16 + r, 4 * (c & 1) + g, H2_RSX);
                    acc = h2_mma(wfh[cc], wfl[cc], *(const uint4*)(OH + off), *(const uint4*)(OH + H2_PDX + off), acc);
                }
            }
            *(f32x4*)(RED + (wave * 16 + r) * 16 + 4 * g) = acc;
        }
        __syncthreads();
        if (tid < NS * P) {
            const int s = tid / P, p = tid - s * P, b = IND ? sidx[s] : b0 + s;
            if (IND ? b >= 0 : b < B) {
                float a = H2G(float, N.bv2)[p];
                for (int j = 0; j < P; j++) {
                    float h = 0.f;
                    for (int w = 0; w < 12; w++) h += RED[(w * 16 + s) * 16 + j];
                    a += fmaxf(h * N.sv1 + H2G(float, N.bv1)[j], 0.f) * H2G(float, N.Wv2)[j * P + p];
                }
                if constexpr (IND) __hip_atomic_store((uint32_t*)v_out + (size_t)b * P + p, __float_as_uint(tanhf(a)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v_out[(size_t)b * P + p] = tanhf(a);
            }
        }
        H2_PH(5);
    }
}

// the whole forward of workgroup `wg` (samples 16 * wg ..); the caller's workgroup has NW waves and H2_LDS bytes of LDS at `lds`
// IND: see h2_block -- `boards` and `valid` are both the pipeline's leaf-record array, sidx[16] (LDS) names the workgroup's trees
template <int NW, bool IND = false>
__device__ __forceinline__ void h2_net_body(uint8_t* lds, H2WeightsC Wc, const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid,
                                            int B, int P, float* __restrict__ pi_out, float* __restrict__ v_out, int wg, const int* sidx = nullptr) {
    constexpr int NS = 16, C = 56, NT = NW * 64, KB = (NS * (7 * C / 4) + NT - 1) / NT;
    h2_fp16_saturate_mode();
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));      // (opaque: inside the round loop of k_rounds_v80 nothing derived from the thread id may be hoisted out of the loop)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int b0 = wg * NS;
    const int nb = min(NS, B - b0);
    uint8_t* const XH = lds + H2_XH;
    uint8_t* const X0 = lds + H2_OH;                          // the board tile as ONE f16 plane (64 * int8 is exact): O is free until the heads

    // ---- board tile int8 [s][c][l] -> X0[l*16 + s][c] = 64 * board, requested before any weight ----
    const uint32_t* bsrc = (const uint32_t*)(boards + (size_t)b0 * (7 * C));
    uint32_t bv[KB];
    unsigned long long ind_mask = 0ull;
    if constexpr (IND) {
        if (tid < 32) {
            const int b = sidx[tid >> 1];
            if (b >= 0) ind_mask = __hip_atomic_load((const unsigned long long*)(valid + (size_t)b * H2_AL_STRIDE + H2_AL_MASK) + (tid & 1), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int k = 0; k < KB; k++) {
        const int i = tid + NT * k;
        if constexpr (IND) {
            const int s = i / (7 * C / 4), d = i - s * (7 * C / 4);
            const int b = i < NS * (7 * C / 4) ? sidx[s] : -1;
            bv[k] = b >= 0 ? __hip_atomic_load((const uint32_t*)(boards + (size_t)b * H2_AL_STRIDE) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        } else
            bv[k] = (i < NS * (7 * C / 4) && i / (7 * C / 4) < nb) ? bsrc[i] : 0u;
    }
    constexpr int RG0 = NW / 4;                               // row-tile groups of the first layer
    const int ntp = wave & 3, rt0 = wave >> 2;
    uint4 w0h[2], w0l[2];
    float s0;
    f32x4 b04;
    H2EW ew;
    {
        const H2WeightsC c0 = h2_opaque(Wc);
        const H2NetW N = h2_load_const(&c0->N);
        const H2BlockW Wt = h2_load_const(&c0->Wt);
#pragma unroll
        for (int c = 0; c < 2; c++) { w0h[c] = H2FRAG(N.W0, 2, ntp, c, 0); w0l[c] = H2FRAG(N.W0, 2, ntp, c, 1); }
        b04 = *H2G(f32x4, N.b0 + ntp * 16 + 4 * g);
        s0 = N.s0;
        h2_load_ew(ew, Wt, wave < 11 ? wave : 0, g, lane);   // the trunk block's first operands, behind the board tile and W0
    }
    // zero what is read but never written: the pooled / SE-hidden planes (pad columns) and X0's columns 56..63
    if (tid < 768) *(uint4*)(lds + H2_PLH + tid * 16) = make_uint4(0u, 0u, 0u, 0u);       // 768 x 16 B = both PL planes
    if (tid < 256) *(uint4*)(lds + H2_SHH + tid * 16) = make_uint4(0u, 0u, 0u, 0u);       // both SH planes
    if (tid < 112) *(uint4*)(X0 + h2_off(tid, 7, H2_RSX)) = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (IND) { if (tid < 32) ((unsigned long long*)(sidx + H2_IND_MASK))[tid] = ind_mask; }
    __syncthreads();                                           // (the zeroing of X0's last chunk precedes the scatter below: none overlap, but PL/SH need it anyway)
#pragma unroll
    for (int k = 0; k < KB; k++) {
        const int i = tid + NT * k;
        if (i < NS * (7 * C / 4)) {
            const int s = i / (7 * C / 4), d = i - s * (7 * C / 4);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = 4 * d + q, c = idx / 7, l = idx - c * 7;
                const _Float16 hv = (_Float16)(float)((int)(int8_t)(bv[k] >> (8 * q)) * 64);
                *(_Float16*)(X0 + h2_off(l * 16 + s, c >> 3, H2_RSX) + ((c & 7) << 1)) = hv;
            }
        }
    }
    __syncthreads();
    // ---- first layer Linear(56, 56) + BN (SplendorNNet.py:397-401) -> X planes; the operand has no lo part ----
#pragma unroll 1
    for (int rt = rt0; rt < 7; rt += RG0) {
        const int row = rt * 16 + r;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const f16x8 a = __builtin_bit_cast(f16x8, *(const uint4*)(X0 + h2_off(row, 4 * c + g, H2_RSX)));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w0l[c]), a, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w0h[c]), a, acc, 0, 0, 0);
        }
        h2_store4(XH, H2_PDX, H2_RSX, row, ntp * 16 + 4 * g, acc * s0 + b04);
    }
    __syncthreads();
    // V80 geometry (SplendorNNet.py:262-283): trunk ReLU + mean squeeze, both heads Hardswish + max squeeze
    {
        const H2WeightsC c1 = h2_opaque(Wc);
        const H2BlockW Wt = h2_load_const(&c1->Wt), Wp = h2_load_const(&c1->Wp);
        const H2NetW N = h2_load_const(&c1->N);
        h2_block<1, 0, 1, NW, IND>(lds, Wt, N, B, P, valid, pi_out, v_out, ew, Wp, wg, tid, sidx);
    }
    {
        const H2WeightsC c2 = h2_opaque(Wc);
        const H2BlockW Wp = h2_load_const(&c2->Wp), Wv = h2_load_const(&c2->Wv);
        const H2NetW N = h2_load_const(&c2->N);
        h2_block<2, 1, 2, NW, IND>(lds, Wp, N, B, P, valid, pi_out, v_out, ew, Wv, wg, tid, sidx);
    }
    {
        const H2WeightsC c3 = h2_opaque(Wc);
        const H2BlockW Wv = h2_load_const(&c3->Wv);
        const H2NetW N = h2_load_const(&c3->N);
        h2_block<2, 1, 3, NW, IND>(lds, Wv, N, B, P, valid, pi_out, v_out, ew, Wv, wg, tid, sidx);   // (the last prefetch is unused)
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_v80_net_h2(H2Weights W /* first argument: offset 0 of the kernel argument segment; never read
                                                                        by name (see H2WeightsC) */,
                                                        const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid, int B, int P,
                                                        float* __restrict__ pi_out, float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const H2WeightsC Wc = (H2WeightsC)__builtin_amdgcn_kernarg_segment_ptr();
    h2_net_body<NW>(lds, Wc, boards, valid, B, P, pi_out, v_out, (int)blockIdx.x);
}

}  // namespace azg

#pragma clang fp contract(off)
