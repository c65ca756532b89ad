// nn_conv5x5.hip.h -- the Santorini ResNet (santorini/SantoriniNNet.py nn_version 88/89: first 3x3 conv 2 -> 64 + BN + ReLU,
// NB SimpleResBlocks :71-84 of two 3x3 convs 64 -> 64, SimpleHead policy / value heads :17-40) as ONE launch per leaf
// batch.  A workgroup owns 8 samples = 200 board cells; the two activation tiles [200][64] f32 live in LDS, each 3x3
// convolution is an implicit GEMM out[cell][co] = sum_{tap, ci} in[cell + tap][ci] * W[tap*64 + ci][co] on
// v_mfma_f32_16x16x4_f32 (weights = A operand in fragment order, activations = B operand read as float4 from LDS; a tap
// that leaves the 5x5 board contributes zeros).  12 waves = 4 output-channel tiles x 3 groups of cell tiles; a wave keeps
// the weight fragments of one kernel row (3 taps x 64 channels = 48 VGPRs) in registers while it walks its cell tiles, and
// the accumulators of its (at most 5) cell tiles across the three kernel rows.  The heads are a few thousand MACs per
// sample and run on the vector ALUs.  (MIOpen's Winograd path needs 11 launches of ~100 us for the same batch.)
#pragma once
#include <type_traits>
#include "nn_kernels.hip.h"
#include "nn_mb1d.hip.h"

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

// Row order of an activation tile.  Sample-major (CM = false, rounds 1-3): row = sample * 25 + cell -- a 16-row MFMA tile mixes up to 16
// different cells, so every tile needs all nine taps and a tap that leaves the board is an MFMA on the zero row.  CELL-major (CM = true,
// the f16 x 2 ResNet kernel since round 4): row = cell * NS + sample; with NS = 8 a tile is exactly two neighbouring cells, the taps that
// leave the board for BOTH of them are skipped for the whole tile (wave-uniform): 99 (tile, tap) pairs per convolution instead of 117,
// i.e. 15 % fewer MFMAs and LDS operand reads.  A neighbour cell is NS * (dy * 5 + dx) rows away.
template <int NS, bool CM> __device__ __forceinline__ int c5_cell(int r) { return CM ? r / NS : r % 25; }
template <int NS, bool CM> __device__ __forceinline__ int c5_sample(int r) { return CM ? r % NS : r / 25; }
template <int NS, bool CM> __device__ __forceinline__ int c5_step(int dcell) { return CM ? NS * dcell : dcell; }
__host__ __device__ constexpr uint32_t c5_cell_taps(int cell) {        // bit t = tap t (ky * 3 + kx) stays on the 5 x 5 board
    const int y = cell / 5, x = cell - 5 * y;
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < 5 && xx >= 0 && xx < 5) m |= 1u << t;
    }
    return m;
}

// Row tile of slot i of row group rg (12 waves = 4 output-channel tiles x 3 row groups).  Sample-major: rt = rg + 3 i (group 0 takes the odd
// 13th tile).  Cell-major: the 13 tiles need 6 6 8 9 9 9 9 9 9 9 6 6 4 taps (99), dealt out so that every group walks 33 (tile, tap) pairs
// -- {3 4 5 0}, {6 7 8 1}, {9 2 10 11 12} -- instead of 37 / 30 / 32 with the strided deal (the convolution ends with its slowest group).
template <bool CM> __host__ __device__ constexpr int c5_tile_of(int rg, int i) {
    if (!CM) return rg + 3 * i;
    constexpr int T[3][5] = {{3, 4, 5, 0, 99}, {6, 7, 8, 1, 99}, {9, 2, 10, 11, 12}};
    return T[rg][i];
}
template <bool CM> __host__ __device__ constexpr int c5_last_group() { return CM ? 2 : 0; }     // the group with five tiles

// The (kernel row ky, fragment k6 = kx * 2 + K chunk, tile slot i) steps of one convolution for the waves of row group RGV (row tiles
// rt = RGV + 3 i, i < NT), kernel row by kernel row, fragment by fragment, in the order the software pipeline walks them.  CM: a step is
// kept only if its tap stays on the board for at least one of the tile's two cells; last[s] marks a fragment's final step of its kernel row.
template <int NT, int RGV, bool CM>
struct C5Steps {
    int n = 0;
    int8_t ky[18 * NT] = {}, k6[18 * NT] = {}, ti[18 * NT] = {};
    int8_t kind[18 * NT] = {};          // CM: which of the tile's two cells the tap stays on the board for -- 0 both, 1 the first only, 2 the second only
    bool last[18 * NT] = {};
    // The epilogue of a tile (bias, residual, ReLU, split, store) is straight-line code inside the step loop's instance, issued EPI_DELAY
    // steps after the tile's final MFMA: epi[s] = bit i: tile slot i's epilogue goes after step s; tail_order = the tiles left for after
    // the loop, in completion order.  bias_at = the step before which the bias is requested.  AZG_C5_TILE_MAJOR 1 walks the LAST kernel
    // row tile by tile, so that the accumulators complete one after the other and the epilogues sit between the MFMAs of the tiles that
    // follow.  Measured (round 4): no gain over all epilogues after the loop (179 us either way) -- what a convolution loses is not the
    // epilogue but its tail, the last wave of a SIMD finishing alone (AZG_C5_PRIO below); the default keeps the fragment-major order.
#ifndef AZG_C5_EPI_DELAY
#define AZG_C5_EPI_DELAY 100   /* 2 with AZG_C5_TILE_MAJOR 1: measured the same 179 us as the epilogues after the loop, which is the default */
#endif
#ifndef AZG_C5_TILE_MAJOR
#define AZG_C5_TILE_MAJOR 0
#endif
    static constexpr int EPI_DELAY = AZG_C5_EPI_DELAY;
    uint8_t epi[18 * NT] = {};
    int8_t fin[NT] = {};                // final step of tile slot i
    int8_t tail_order[NT] = {};
    int n_tail = 0, bias_at = 0;
    constexpr void add(int y, int k, int i) {
        const int rt = c5_tile_of<CM>(RGV, i), c0 = 2 * rt, c1 = 2 * rt + 1;
        const uint32_t ta = CM ? (c0 < 25 ? c5_cell_taps(c0) : 0u) : 0x1FFu, tb = CM ? (c1 < 25 ? c5_cell_taps(c1) : 0u) : 0x1FFu;
        const bool on_a = (ta >> (y * 3 + (k >> 1))) & 1u, on_b = (tb >> (y * 3 + (k >> 1))) & 1u;
        if (on_a || on_b) {
            ky[n] = (int8_t)y; k6[n] = (int8_t)k; ti[n] = (int8_t)i; kind[n] = (int8_t)(on_a && on_b ? 0 : on_a ? 1 : 2); last[n] = false; n++;
        }
    }
    constexpr C5Steps() {
        for (int y = 0; y < 3; y++) {
            const int row0 = n;
            if (CM && y == 2 && AZG_C5_TILE_MAJOR) {
                for (int i = 0; i < NT; i++)
                    for (int k = 0; k < 6; k++) add(y, k, i);
            } else {
                for (int k = 0; k < 6; k++)
                    for (int i = 0; i < NT; i++) add(y, k, i);
            }
            // a fragment's final step of this kernel row (where its registers are refilled with the next row's fragment)
            // (every fragment keeps at least one tile in every row group: the board has five rows and columns, a row group of at
            // least four tiles always holds a cell that is neither on the top / bottom nor on the left / right edge)
            for (int k = 0; k < 6; k++) {
                int l = -1;
                for (int s = row0; s < n; s++) if (k6[s] == k) l = s;
                if (l >= 0) last[l] = true;
            }
        }
        int first_fin = n;
        for (int i = 0; i < NT; i++) {
            int f = 0;
            for (int s = 0; s < n; s++) if (ti[s] == i) f = s;
            fin[i] = (int8_t)f;
            if (f < first_fin) first_fin = f;
        }
        bias_at = first_fin > 4 ? first_fin - 4 : 0;
        // tiles in completion order
        bool done[NT] = {};
        for (int r = 0; r < NT; r++) {
            int best = -1;
            for (int i = 0; i < NT; i++) if (!done[i] && (best < 0 || fin[i] < fin[best])) best = i;
            done[best] = true;
            if (fin[best] + EPI_DELAY < n) epi[fin[best] + EPI_DELAY] |= (uint8_t)(1u << best);
            else tail_order[n_tail++] = (int8_t)best;
        }
    }
};

template <int NT, int RGV, bool CM> struct C5StepList { static constexpr C5Steps<NT, RGV, CM> value{}; };
// compile-time loop: f(integral_constant<int, I>) for I = LO .. HI - 1 (indices into register arrays stay constants whatever the body's size)
template <int LO, int HI, class F>
__device__ __forceinline__ void c5_static_for(F&& f) {
    if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); c5_static_for<LO + 1, HI>(f); }
}

// one 3x3 convolution over the workgroup's tile.  KC = input-channel chunks of 16 per tap.
//   IN [ROWS][CS] -> OUT [ROWS][CS] = relu(conv(IN) + bias (+ RES)); OUT may alias RES (same lane reads and writes an element)
template <int KC, int NS, bool RELU = true>
__device__ __forceinline__ void conv3x3_tile(const float* __restrict__ Wfrag, const float* __restrict__ bias,
                                             const float* IN, float* OUT, const float* RES) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, CS = 68, KCH = 9 * KC, RG = 3, MAXT = (RT + RG - 1) / RG;
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
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
        *(float4*)dst = RELU ? make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)) : o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-precision variant of the trunk (the MI355X matrix cores run f32 inputs at 1/16 of the bf16 rate).  Every f32 value x is
// carried as three bf16 numbers x = hi + mid + lo (each rounded to nearest-even from what the previous ones left over: 24
// significant bits in all, x - (hi + mid + lo) <= 2^-25 |x|), and a product a*b is accumulated in f32 as the six partial
// products of at least 2^-24 relative weight: lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi -- bf16 x bf16 is exact in f32,
// so the result differs from the f32 MFMA's only by the dropped 2^-24 terms and the accumulation order (checked <= 1e-5 on the
// net's outputs against the reference model like the f32 path).  Six v_mfma_f32_16x16x32_bf16 cover K = 32 in ~100 cycles where
// eight v_mfma_f32_16x16x4_f32 need 256.
//   activations in LDS: three planes [ROWS + 2][64] bf16 per tile (row = 128 B; the 16-byte chunk q of a row sits at
//   q ^ (row & 7), which makes the ds_read_b128 of 16 consecutive rows conflict-free; rows ROWS, ROWS + 1 stay zero: they are what a
//   tap that leaves the 5x5 board reads, so no select is needed on the operand registers -- pl_off_z: the lane reads the zeros from
//   the 16-byte bank column ITS on-board address would have used, so the other lanes of its ds_read_b128 group never meet it on a bank;
//   rounds 2-3 had ONE zero row and every lane off the board read its chunk q: a second address on the bank column of the group's
//   (row & 7) == 0 lane -- SQ_LDS_BANK_CONFLICT 64 % of the kernel's LDS cycles); written split by the producing epilogue;
//   weights: frag[ct 4][chunk 18][plane 3][lane 64][8] bf16 = W_plane[32*chunk + 8*(lane>>4) + j][16*ct + (lane&15)], K = tap*64 + ci.
// (bf16x8, split3x2, pl_off, store_split4, load_split4: nn_kernels.hip.h)
constexpr int C5_LDS_LEAD = 6 * 1024;                    // cell-major f16 x 2 kernel: the tiles start six cells (of 8 rows x 128 B) into the LDS
constexpr uint32_t C5_LDS_NOWHERE = 0x40000000u;         // an LDS address outside every allocation: reads return zeros
template <int ROWS> __device__ __forceinline__ int pl_off_z(int r, int q, bool on) {
    static_assert(ROWS % 2 == 0, "the zero rows start on a 256-byte bank row");
    const int o = pl_off(r, q);
    return on ? o : ROWS * 128 + (o & 255);
}

// the first convolution (2 board planes, K = 9 x 16): f32 MFMA from the f32 staging tile, output written split
template <int NS, bool RELU = true, int NPL = 3>
__device__ __forceinline__ void conv3x3_first_split(const float* __restrict__ Wfrag, const float* __restrict__ bias,
                                                    const float* IN, uint8_t* OUT) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, CS = 68, RG = 3, MAXT = (RT + RG - 1) / RG, PB = (ROWS + 2) * 128;
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
    const int ct = wave & 3, rg = wave >> 2;
    float4 w[9];
#pragma unroll
    for (int c = 0; c < 9; c++) w[c] = FRAG(Wfrag, 9, ct, c);
    const float4 b = *(const float4*)(bias + ct * 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int rt = rg + RG * i, r = rt * 16 + r16;
        if (rt >= RT) continue;
        const int cell = r % 25, y = cell / 5, x = cell - 5 * y;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool on = r < ROWS && yy >= 0 && yy < 5 && xx >= 0 && xx < 5;
            float4 a = *(const float4*)(IN + (on ? r + (t / 3 - 1) * 5 + (t % 3 - 1) : 0) * CS + 4 * g);
            if (!on) a = make_float4(0.f, 0.f, 0.f, 0.f);
            // component j of the fragments covers K = j, 4 + j, 8 + j, 12 + j of the tap: the two board planes are K = 0 and 1,
            // the other fourteen channels of the staging tile are zero padding -> components z and w contribute nothing
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t].y, a.y, acc, 0, 0, 0);
        }
        if (r < ROWS) {
            float4 o = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
            if (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            if (NPL == 2) h2_store4(OUT, PB, 128, r, ct * 16 + 4 * g, f32x4{o.x, o.y, o.z, o.w});
            else store_split4(OUT, PB, r, ct * 16 + 4 * g, o);
        }
    }
}

// The first convolution on one f16 MFMA chunk: K = 9 taps x 2 board planes = 18 of the 32 (k = 2 * tap + plane).  The board values are
// small integers (exact in f16, no lo half), the weights split hi + lo on the fly from the f32 fragments: two v_mfma_f32_16x16x32_f16 per
// row tile instead of eighteen v_mfma_f32_16x16x4_f32 (a sixteenth of the rate each).  Output written as f16 x 2 planes.
#ifdef AZG_NN_PHASE_TIMES
static __device__ long long g_c5_first[8];
#define C5F_PH(k) do { if (blockIdx.x == 7 && threadIdx.x == 0) g_c5_first[k] = clock64(); } while (0)
#else
#define C5F_PH(k) do { } while (0)
#endif
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };
// `prefetch` is called once this convolution's own weights are requested: whatever it asks for travels behind them and lands during
// the tile loop (which only touches LDS) instead of delaying the first MFMAs
// WLDS: Wfrag points to the compact LDS copy [4 ct][9 taps][16 output channels][2 board planes] made at kernel start (and `bias` to LDS
// as well): the gather from the padded f32 fragment array in global memory (8 dword loads per lane, 128 cache lines per wave) and the bias
// behind the prefetch made this convolution as long as a trunk convolution (4.2 k cycles until the weights had arrived, 4.9 k in the tile
// loop waiting for 16 bytes of bias)
template <int NS, bool RELU = true, class Prefetch = NoPrefetch, bool CM = false, bool WLDS = false>
__device__ __forceinline__ void conv3x3_first_h2(const float* __restrict__ Wfrag, const float* __restrict__ bias,
                                                 const float* IN, uint8_t* OUT, Prefetch prefetch = Prefetch()) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, CS = 68, RG = 3, MAXT = (RT + RG - 1) / RG, PB = (ROWS + 2) * 128;
    constexpr float WS = 256.f;                             // weight scale (|w| < 255 keeps the hi half finite)
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
    const int ct = wave & 3, rg = wave >> 2;
    uint4 wh, wl;
    C5F_PH(0);
    {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int tap = 4 * g + (j >> 1);
            if (WLDS) wv[j] = tap < 9 ? Wfrag[(((ct * 9 + tap) * 16 + r16) << 1) + (j & 1)] : 0.f;
            else wv[j] = tap < 9 ? Wfrag[((((size_t)ct * 9 + tap) * 64 + r16) << 2) + (j & 1)] : 0.f;
        }
        prefetch();
#pragma unroll
        for (int j = 0; j < 8; j++) wv[j] *= WS;
        h2_split2(wv[0], wv[1], wh.x, wl.x); h2_split2(wv[2], wv[3], wh.y, wl.y);
        h2_split2(wv[4], wv[5], wh.z, wl.z); h2_split2(wv[6], wv[7], wh.w, wl.w);
    }
    C5F_PH(1);
    const float4 b = *(const float4*)(bias + ct * 16 + 4 * g);
    __builtin_amdgcn_sched_barrier(0);                      // (keeps the prefetch requests behind this convolution's own)
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int rt = rg + RG * i, r = rt * 16 + r16;
        if (rt >= RT) continue;
        const int cell = r < ROWS ? c5_cell<NS, CM>(r) : 0, y = cell / 5, x = cell - 5 * y;
        uint32_t bv[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int tap = 4 * g + jj, dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
            const bool on = r < ROWS && tap < 9 && y + dy >= 0 && y + dy < 5 && x + dx >= 0 && x + dx < 5;
            const float2 v = *(const float2*)(IN + (on ? r + c5_step<NS, CM>(dy * 5 + dx) : 0) * CS);
            bv[jj] = on ? __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v.x, v.y}, f16x2_t)) : 0u;
        }
        const f16x8 bh = __builtin_bit_cast(f16x8, make_uint4(bv[0], bv[1], bv[2], bv[3]));
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), bh, acc, 0, 0, 0);
        if (r < ROWS) {
            f32x4 o = acc * (1.f / WS) + f32x4{b.x, b.y, b.z, b.w};
            if (RELU) o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
            h2_store4(OUT, PB, 128, r, ct * 16 + 4 * g, o);
        }
    }
    C5F_PH(2);
}

// The two 1x1-convolution heads of the SimpleHead pair (policy: 64 -> 2 channels, value: 64 -> 1; + folded BN + ReLU) on the trunk's
// f16 x 2 planes: one MFMA column tile (channels 0, 1 = policy, 2 = value, the rest zero), weights split on the fly from the f32
// matrices Wp [64][2] / Wv [64][1]; wave w owns row tile w (wave 0 the 13th as well).  HP [NS][2 * 25] (channel-major), HV [NS][25].
// HPT: the policy features feature-major, HP[(c * 25 + cell) * NS + sample] (the policy FC then reads the NS samples of a feature as two
// float4), instead of HP[sample][c * 25 + cell]
template <int NS, bool CM = false, bool HPT = false>
__device__ __forceinline__ void heads1x1_h2(const float* __restrict__ Wp, const float* __restrict__ bp, const float* __restrict__ Wv,
                                            const float* __restrict__ bv, const uint8_t* IN, float* HP, float* HV) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, PB = (ROWS + 2) * 128;
    constexpr float WS = 256.f;
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
    uint4 wh[2], wl[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = 32 * c + 8 * g + j;
            wv[j] = (r16 < 2 ? Wp[k * 2 + r16] : r16 == 2 ? Wv[k] : 0.f) * WS;
        }
        h2_split2(wv[0], wv[1], wh[c].x, wl[c].x); h2_split2(wv[2], wv[3], wh[c].y, wl[c].y);
        h2_split2(wv[4], wv[5], wh[c].z, wl[c].z); h2_split2(wv[6], wv[7], wh[c].w, wl[c].w);
    }
    const float b0 = bp[0], b1 = bp[1], b2 = bv[0];
    for (int rt = wave; rt < RT; rt += 12) {
        const int r = rt * 16 + r16;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint8_t* src = IN + pl_off_z<ROWS>(r, 4 * c + g, r < ROWS);
            acc = h2_mma(wh[c], wl[c], *(const uint4*)src, *(const uint4*)(src + PB), acc);
        }
        if (g == 0 && r < ROWS) {                           // lanes of g = 0 hold output channels 0..3 of row r
            const float ds = 1.f / (WS * H2_AS);
            const int smp = c5_sample<NS, CM>(r), cell = c5_cell<NS, CM>(r);
            HP[HPT ? cell * NS + smp : smp * 50 + cell] = fmaxf(acc[0] * ds + b0, 0.f);
            HP[HPT ? (25 + cell) * NS + smp : smp * 50 + 25 + cell] = fmaxf(acc[1] * ds + b1, 0.f);
            HV[smp * 25 + cell] = fmaxf(acc[2] * ds + b2, 0.f);
        }
    }
}

#ifdef AZG_NN_PHASE_TIMES
static __device__ long long g_c5_phase[32];
#define C5_PH(k) do { if (blockIdx.x == 7 && threadIdx.x == 0) g_c5_phase[k] = clock64(); } while (0)
#else
#define C5_PH(k) do { } while (0)
#endif
struct SplitFrag { uint4 h, m, l; };                        // operand fragments of one (tile, tap, K chunk of 32)
#define AZG_BF(x) __builtin_bit_cast(bf16x8, x)

// one 64 -> 64 3x3 convolution on split activations: OUT = relu(conv(IN) + bias (+ RES)); OUT may alias RES
// NPL = 3: bf16 x 3 (hi + mid + lo, six MFMAs per product).  NPL = 2: f16 x 2 (hi + lo: 22 significant bits, three
// v_mfma_f32_16x16x32_f16 per product -- lo*hi, hi*lo, hi*hi; nn_v80_h2.hip.h): two planes per tile, the activation planes hold
// 64 * x, the weight fragments W * 2^k, `descale` = 2^-k / 64 brings the accumulator back
template <int NS, int NPL = 3, bool PRELOADED = false, bool CM = false>
__device__ __forceinline__ void conv3x3_split(const uint4* __restrict__ Wfrag, const float* __restrict__ bias, const uint8_t* IN,
                                              uint8_t* OUT, const uint8_t* RES, float descale = 1.f,
                                              const uint4* __restrict__ WNEXT = nullptr, uint4 (*wio)[6][3] = nullptr) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, RG = 3, MAXT = (RT + RG - 1) / RG, PB = (ROWS + 2) * 128, KCH = 18;
    static_assert(MAXT == 5 && RT - RG * (MAXT - 1) == 1, "step schedule: two tile pairs per wave + one odd tile in the first row group");
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
    const int ct = wave & 3, rg = wave >> 2;
    C5_PH(24);
    static_assert(!CM || NS == 8, "cell-major tiles: two cells of eight samples per 16-row tile");
    int row[MAXT];
    uint32_t tapmask[MAXT];
    int r16v = r16;
    if (CM) asm volatile("" : "+v"(r16v));      // opaque per call: with every step a compile-time constant the ~90 operand addresses of a
                                                // convolution are loop-invariant in the caller's block loop, and hoisted out of it they
                                                // cost 118 spilled registers
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int rt = c5_tile_of<CM>(rg, i), r = rt * 16 + r16v;
        row[i] = r;
        tapmask[i] = (rt < RT && r < ROWS) ? c5_cell_taps(c5_cell<NS, CM>(r)) : 0u;
    }
    f32x4 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool last_slot = rg == c5_last_group<CM>();       // wave-uniform: this wave has a tile in slot MAXT - 1
    float4 b;                                               // the bias of this lane's four output channels (CM: from LDS, requested a few steps before the end)
    auto load = [&](int i, int t, int c) {
        const bool on = (tapmask[i] >> t) & 1u;
        const int r = row[i] + c5_step<NS, CM>((t / 3 - 1) * 5 + (t % 3 - 1));
        const uint8_t* src = IN + pl_off_z<ROWS>(r, 4 * c + g, on);                                   // off the board: the zero rows
        return SplitFrag{*(const uint4*)src, *(const uint4*)(src + PB), NPL == 3 ? *(const uint4*)(src + 2 * PB) : make_uint4(0u, 0u, 0u, 0u)};
    };
    // one (tile, tap, K chunk of 32): three 16-byte operand reads, six MFMAs into the tile's accumulator
    auto step = [&](int i, int t, int c, bf16x8 wh, bf16x8 wm, bf16x8 wl) {
        const SplitFrag a = load(i, t, c);
        if (NPL == 2) {                                      // planes: h = hi, m = lo (f16)
            acc[i] = h2_mma(__builtin_bit_cast(uint4, wh), __builtin_bit_cast(uint4, wm), a.h, a.m, acc[i]);
            return;
        }
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, AZG_BF(a.h), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, AZG_BF(a.l), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, AZG_BF(a.m), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, AZG_BF(a.h), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, AZG_BF(a.m), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, AZG_BF(a.h), acc[i], 0, 0, 0);
    };
    // the weight fragments of one kernel row, [kx * 2 + c][plane].  NPL == 2: a fragment's registers are refilled with the next
    // kernel row's fragment as soon as its tiles are issued (rolling prefetch: the L2 round trip of a row's weights hides behind
    // the five other fragments' MFMAs instead of stalling all twelve waves at the top of every kernel row); WNEXT = the next
    // convolution's fragments, whose first row is requested during the last row of this one and handed over in `w`
    uint4 w[6][3];
    auto wfrag = [&](const uint4* __restrict__ W, int ky, int c6, int p) { return W[(((size_t)ct * KCH + ky * 6 + c6) * NPL + p) * 64 + lane]; };
    if (!PRELOADED) {
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++)
#pragma unroll
            for (int p = 0; p < 3; p++) w[c6][p] = p < NPL ? wfrag(Wfrag, 0, c6, p) : make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++)
#pragma unroll
            for (int p = 0; p < 3; p++) w[c6][p] = (*wio)[c6][p];
    }
    C5_PH(25);
    if constexpr (NPL == 2) {
        // software pipeline: the operands of step s + AH are requested before the MFMAs of step s are issued (AH = 1, 2, 3 measure the
        // same 220-230 us per 4096 leaves, 4 spills; interleaving the MFMAs of two tiles so that consecutive ones never share an
        // accumulator is 4 % SLOWER -- back-to-back MFMAs on one accumulator are the cheap case) (hipcc emits
        // ds_read pair -> s_waitcnt -> three MFMAs per step otherwise, i.e. every step waits out an LDS round trip), across the
        // kernel rows as well; a step = (fragment k6 = s / NT: tap ky * 3 + k6 / 2, K chunk k6 & 1; tile s % NT).  The waves with
        // the odd 13th row tile run the NT = MAXT instance, the others NT = MAXT - 1 (wave-uniform branch)
        // CM (cell-major tiles): the steps whose tap leaves the board for both cells of the tile are dropped AT COMPILE TIME -- the row
        // group rg of the wave is a template argument of the loop body (three instances), every kernel row is unrolled, and the list of
        // the remaining steps (C5Steps) drives the same software pipeline; the code stays straight-line (a first version skipped the
        // steps with wave-uniform branches: the branches broke the pipelining, 226 -> 265 us per 4096 leaves).
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef u32x4_t __attribute__((address_space(3))) lds_u4;
        const uint32_t in0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)IN;
        uint32_t abase[MAXT];
#pragma unroll
        for (int i = 0; i < MAXT; i++) abase[i] = in0 + (uint32_t)pl_off(row[i], g) - 6u * 1024u;
        const bool in_a = (r16v & 8) == 0;                   // this lane's row belongs to the first of its tile's two cells
        // the epilogue of tile slot i of row group RGV: straight-line but for the one tile that carries the eight pad rows (a compile-time fact)
        const uint32_t out0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)OUT;
        const uint32_t res0 = RES ? (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)RES : 0u;
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        typedef u32x2_t __attribute__((address_space(3))) lds_u2;
        auto epilogue_cm = [&](auto I, auto rg_tag) {
            constexpr int i = decltype(I)::value, rt = c5_tile_of<CM>(decltype(rg_tag)::value, i);
            const uint32_t off = (uint32_t)pl_off(row[i], 2 * ct + (g >> 1)) + (uint32_t)((g & 1) << 3);
            // (everything in units of the planes' scale: `bias` holds 64 b, the accumulator is brought back by 64 * descale, the residual
            // is taken as stored -- scaling by 2^6 commutes with every rounding here, the bits are those of relu(acc * descale + b + res) * 64)
            f32x4 o = acc[i] * (descale * H2_AS) + f32x4{b.x, b.y, b.z, b.w};
            if (RES) {
                typedef uint64_t __attribute__((address_space(3))) lds_u64;
                const uint64_t hv = *(const lds_u64*)(uintptr_t)(res0 + off), lv = *(const lds_u64*)(uintptr_t)(res0 + off + (uint32_t)PB);
                const f32x2 ra = h2_join2((uint32_t)hv, (uint32_t)lv), rb = h2_join2((uint32_t)(hv >> 32), (uint32_t)(lv >> 32));
                o += f32x4{ra.x, ra.y, rb.x, rb.y};                                                  // == 64 * h2_load4
            }
            o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
            uint32_t h0, l0, h1, l1;
            h2_split2(o[0], o[1], h0, l0);
            h2_split2(o[2], o[3], h1, l1);
            const uint32_t dst = out0 + off;
            if (2 * rt + 1 < 25 || in_a) {                                                         // (rows >= ROWS of the 13th tile: no store)
                *(lds_u2*)(uintptr_t)dst = u32x2_t{h0, h1};
                *(lds_u2*)(uintptr_t)(dst + (uint32_t)PB) = u32x2_t{l0, l1};
            }
        };
        auto run = [&](auto nt_tag, auto rg_tag) {
            constexpr int NT = decltype(nt_tag)::value, RGV = decltype(rg_tag)::value;
            using SL = C5StepList<NT, RGV, CM>;
            constexpr int NSTEP = SL::value.n;
            // Operand addresses without arithmetic (round 4, later).  Cell-major rows put a neighbour cell 8 * (dy * 5 + dx) rows = a multiple
            // of 1 KiB away, which leaves the row's chunk swizzle (row & 7) alone: the address of (tile i, tap) is abase[i] + a COMPILE-TIME
            // constant that goes into the ds_read's offset field (abase is taken 6 cells back so that the constant is never negative -- the
            // tiles start C5_LDS_LEAD bytes into the workgroup's LDS).  A lane whose tap leaves the board gets an address far outside the
            // allocation instead: an out-of-range LDS read returns zeros (the architected behaviour of the DS unit), touches no bank and
            // needs no zero row; which half of the tile's lanes that is (first cell, second cell) is a compile-time property of the step,
            // so the select is one v_cndmask on a wave mask computed once per convolution, and only on the ~40 % of the steps with a mixed
            // tile.  Before: bit test + compare + select + swizzle, 7.5 VALU instructions per step beside its three MFMAs (the VALU issue
            // slots of a SIMD were 60 % taken, the MFMA pipe 47 % busy).
            auto ld = [&](auto S) {
                constexpr int sx = decltype(S)::value;
                constexpr int ky = SL::value.ky[sx], k6 = SL::value.k6[sx], i = SL::value.ti[sx], kind = SL::value.kind[sx];
                constexpr int kx = k6 >> 1, c = k6 & 1;
                constexpr uint32_t IMM = 1024u * (uint32_t)((ky - 1) * 5 + (kx - 1) + 6);
                uint32_t a = abase[i];
                if constexpr (kind == 1) a = in_a ? a : C5_LDS_NOWHERE;
                if constexpr (kind == 2) a = in_a ? C5_LDS_NOWHERE : a;
                if constexpr (c == 1) a ^= 64u;                                  // chunk 4 + g: bit 2 of the swizzled chunk index
                if constexpr (kind != 0 || c == 1) asm volatile("" : "+v"(a));   // (recomputed per step: kept as common subexpressions the
                                                                                 // up to six variants per tile are 30 live registers -> spills,
                                                                                 // and a spill reload waits with vmcnt(0) for the weight prefetch)
                const lds_u4* src = (const lds_u4*)(uintptr_t)(a + IMM);
                const lds_u4* src1 = (const lds_u4*)(uintptr_t)(a + IMM + (uint32_t)PB);
                const u32x4_t v0 = *src, v1 = *src1;
                return SplitFrag{__builtin_bit_cast(uint4, v0), __builtin_bit_cast(uint4, v1), make_uint4(0u, 0u, 0u, 0u)};
            };
#ifndef AZG_C5_AHEAD
#define AZG_C5_AHEAD 1
#endif
            constexpr int AH = AZG_C5_AHEAD;
            static_assert(AH == 1 || AH == 2, "the compile-time step loop carries one or two look-ahead operands");
            SplitFrag f0 = ld(std::integral_constant<int, 0>{});
            SplitFrag f1 = f0;                               // AH == 2: the operands of the step after the next
            if constexpr (AH == 2 && NSTEP > 1) f1 = ld(std::integral_constant<int, (NSTEP > 1 ? 1 : 0)>{});
            c5_static_for<0, NSTEP>([&](auto I) {
                constexpr int sidx = decltype(I)::value;
                constexpr int ky = SL::value.ky[sidx], k6 = SL::value.k6[sidx], i = SL::value.ti[sidx];
                SplitFrag fn = AH == 2 ? f1 : f0;
                if constexpr (AH == 1 && sidx + 1 < NSTEP) fn = ld(std::integral_constant<int, sidx + 1 < NSTEP ? sidx + 1 : 0>{});
                if constexpr (AH == 2 && sidx + 2 < NSTEP) f1 = ld(std::integral_constant<int, sidx + 2 < NSTEP ? sidx + 2 : 0>{});
#ifndef AZG_C5_PRIO
#define AZG_C5_PRIO 4
#endif
#if AZG_C5_PRIO
                // the three waves of a SIMD share its MFMA pipe: a wave's issue priority falls as it advances through the convolution, so
                // that none runs ahead and leaves the last one to finish alone (a lone wave waits out every LDS round trip): 4 levels over
                // the steps of a convolution, 319.7 k -> 307.7 k cycles per launch; two look-ahead operands (AZG_C5_AHEAD 2) change nothing
                if constexpr (sidx == 0 || (sidx * AZG_C5_PRIO) / NSTEP != ((sidx - 1) * AZG_C5_PRIO) / NSTEP)
                    __builtin_amdgcn_s_setprio(3 - (sidx * AZG_C5_PRIO) / NSTEP * 3 / (AZG_C5_PRIO - 1 > 0 ? AZG_C5_PRIO - 1 : 1));
#endif
                if constexpr (sidx == SL::value.bias_at) b = *(const float4*)(bias + ct * 16 + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
                acc[i] = h2_mma(w[k6][0], w[k6][1], f0.h, f0.m, acc[i]);
                if constexpr (SL::value.last[sidx]) {          // the fragment's last step of this kernel row: refill with the next row's
                    const uint4* __restrict__ Wn = ky < 2 ? Wfrag : (WNEXT ? WNEXT : Wfrag);      // (the trunk's last convolution re-requests its
                    constexpr int kyn = ky < 2 ? ky + 1 : 0;                                      // own first row: skipping that with a uniform
                    w[k6][0] = wfrag(Wn, kyn, k6, 0); w[k6][1] = wfrag(Wn, kyn, k6, 1);           // branch measured 307.9 k -> 310.8 k cycles)
                }
                __builtin_amdgcn_sched_barrier(0);
                f0 = fn;
                if constexpr (CM && SL::value.epi[sidx] != 0)
                    c5_static_for<0, NT>([&](auto J) { if constexpr ((SL::value.epi[sidx] >> decltype(J)::value) & 1) epilogue_cm(J, rg_tag); });
            });
            if constexpr (CM)
                c5_static_for<0, SL::value.n_tail>([&](auto J) { epilogue_cm(std::integral_constant<int, SL::value.tail_order[decltype(J)::value]>{}, rg_tag); });
        };
        if (rg == 0) run(std::integral_constant<int, CM ? MAXT - 1 : MAXT>{}, std::integral_constant<int, 0>{});
        else if (rg == 1) run(std::integral_constant<int, MAXT - 1>{}, std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, CM ? MAXT : MAXT - 1>{}, std::integral_constant<int, 2>{});
    } else
#pragma unroll 1
    for (int ky = 0; ky < 3; ky++) {
        if (NPL == 3 && ky > 0) {
#pragma unroll
            for (int c6 = 0; c6 < 6; c6++)
#pragma unroll
                for (int p = 0; p < 3; p++) w[c6][p] = wfrag(Wfrag, ky, c6, p);
        }
        const uint4* __restrict__ Wn = ky < 2 ? Wfrag : WNEXT;
        const int kyn = ky < 2 ? ky + 1 : 0;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const bf16x8 wh = AZG_BF(w[kx * 2 + c][0]), wm = AZG_BF(w[kx * 2 + c][1]), wl = AZG_BF(w[kx * 2 + c][2]);
#pragma unroll
                for (int i = 0; i < MAXT - 1; i++) step(i, ky * 3 + kx, c, wh, wm, wl);      // straight-line: consecutive tiles
                if (last_slot) step(MAXT - 1, ky * 3 + kx, c, wh, wm, wl);                    // use different accumulators
                if (NPL == 2 && Wn) {
                    w[kx * 2 + c][0] = wfrag(Wn, kyn, kx * 2 + c, 0);
                    w[kx * 2 + c][1] = wfrag(Wn, kyn, kx * 2 + c, 1);
                }
            }
        }
    }
    C5_PH(26);
    if (wio) {
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++)
#pragma unroll
            for (int p = 0; p < 3; p++) (*wio)[c6][p] = w[c6][p];
    }
    if (!(NPL == 2 && CM)) b = *(const float4*)(bias + ct * 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        if (NPL == 2 && CM) continue;                         // (done inside the step loop)
        if (c5_tile_of<CM>(rg, i) >= RT || row[i] >= ROWS) continue;
        if (NPL == 2) {
            f32x4 o = acc[i] * descale + f32x4{b.x, b.y, b.z, b.w};
            if (RES) o += h2_load4(RES, PB, 128, row[i], ct * 16 + 4 * g);
            h2_store4(OUT, PB, 128, row[i], ct * 16 + 4 * g, f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)});
            continue;
        }
        float4 o = make_float4(acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w);
        if (RES) {
            const float4 r = load_split4(RES, PB, row[i], ct * 16 + 4 * g);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        store_split4(OUT, PB, row[i], ct * 16 + 4 * g, make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)));
    }
    C5_PH(27);
}

// One 64 x 64 GEMM phase on split activations (a 1x1 convolution over 64 input channels): acc[i] += IN[tile i] * W for the
// wave's row tiles rt = rg + 3 i and its output-channel tile ct (12 waves = 4 ct x 3 rg, as conv3x3_split).
// Wfrag: [4 ct][2 K chunks of 32][3 planes][64 lanes] uint4; rows >= ROWS read the zero row.
// (the weight fragments w[chunk * 3 + plane] are requested by gemm64_wload a phase ahead)
// (NPL = 2: f16 x 2 operands, [4 ct][2 chunks][2 planes]; w[c * 3 + plane], the third slot unused)
template <int NPL = 3>
__device__ __forceinline__ void gemm64_wload(const uint4* __restrict__ Wfrag, uint4 (&w)[6]) {
    const int tidx_ = nn_tid(), lane = tidx_ & 63, ct = (tidx_ >> 6) & 3;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int p = 0; p < 3; p++) w[c * 3 + p] = p < NPL ? Wfrag[(((size_t)ct * 2 + c) * NPL + p) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
}
template <int NS, int NPL = 3>
__device__ __forceinline__ void gemm64_split(const uint4 (&w)[6], const uint8_t* IN, f32x4 (&acc)[(NS * 25 + 15) / 16 / 3 + 1]) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, RG = 3, MAXT = (RT + RG - 1) / RG, PB = (ROWS + 2) * 128;
    static_assert(MAXT == RT / 3 + 1 && RT - RG * (MAXT - 1) == 1, "tile schedule: MAXT - 1 tiles per wave + one odd tile in the first row group");
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
    const int rg = wave >> 2;
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        if (i == MAXT - 1 && rg != 0) continue;             // (wave-uniform)
        const int r = (rg + RG * i) * 16 + r16;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint8_t* src = IN + pl_off_z<ROWS>(r, 4 * c + g, r < ROWS);
            if (NPL == 2) {
                acc[i] = h2_mma(w[c * 3], w[c * 3 + 1], *(const uint4*)src, *(const uint4*)(src + PB), acc[i]);
                continue;
            }
            const bf16x8 ah = AZG_BF(*(const uint4*)src), am = AZG_BF(*(const uint4*)(src + PB)), al = AZG_BF(*(const uint4*)(src + 2 * PB));
            const bf16x8 wh = AZG_BF(w[c * 3]), wm = AZG_BF(w[c * 3 + 1]), wl = AZG_BF(w[c * 3 + 2]);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, acc[i], 0, 0, 0);
        }
    }
}
#undef AZG_BF

// The same GEMM phase with TWO column tiles per wave (f16 x 2 only; the with-gods trunk, round 6): 12 waves = 2 column-tile pairs x 6 row
// groups, a wave's row tiles rt = rg + 6 i.  A wave reads each activation tile once for its two column tiles: 104 instead of 208 16-byte
// operand reads per GEMM over the workgroup -- the phase was bound by them (1.9 k cycles of LDS bandwidth against 1.4 k of the MFMA pipe).
// w[ctl * 4 + c * 2 + plane]; acc[i * 2 + ctl].
__device__ __forceinline__ void gemm64_wload2(const uint4* __restrict__ Wfrag, uint4 (&w)[8]) {
    const int tidx_ = nn_tid(), lane = tidx_ & 63, ctp = (tidx_ >> 6) & 1;
#pragma unroll
    for (int ctl = 0; ctl < 2; ctl++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int p = 0; p < 2; p++) w[ctl * 4 + c * 2 + p] = Wfrag[(((size_t)(2 * ctp + ctl) * 2 + c) * 2 + p) * 64 + lane];
}
template <int NS>
__device__ __forceinline__ void gemm64_split2(const uint4 (&w)[8], const uint8_t* IN, f32x4 (&acc)[((NS * 25 + 15) / 16 / 6 + 1) * 2]) {
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, RG = 6, MAXT = (RT + RG - 1) / RG, PB = (ROWS + 2) * 128;
    static_assert(MAXT == RT / RG + 1 && RT - RG * (MAXT - 1) == 1, "tile schedule: MAXT - 1 tiles per wave + one odd tile in the first row group");
    const int tidx_ = nn_tid(), lane = tidx_ & 63, wave = tidx_ >> 6, g = lane >> 4, r16 = lane & 15;
    const int rg = wave >> 1;
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        if (i == MAXT - 1 && rg != 0) continue;             // (wave-uniform)
        const int r = (rg + RG * i) * 16 + r16;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint8_t* src = IN + pl_off_z<ROWS>(r, 4 * c + g, r < ROWS);
            const uint4 ah = *(const uint4*)src, al = *(const uint4*)(src + PB);
            acc[i * 2] = h2_mma(w[c * 2], w[c * 2 + 1], ah, al, acc[i * 2]);
            acc[i * 2 + 1] = h2_mma(w[4 + c * 2], w[4 + c * 2 + 1], ah, al, acc[i * 2 + 1]);
        }
    }
}

// IND (the asynchronous pipeline, azg_async.hip.h): sample s of the workgroup is tree sidx[s] (LDS; < 0 = no sample); `boards` and `valid`
// are then both the pipeline's leaf-record array (kernels.hip.h AsyncLeaf<SantoriniDev<1>>: int8 state [80] + valid bit mask u64[3], stride
// 112), read past the L1; the samples' masks are fetched with the boards into `smask` (LDS u64 [8][3]); pi / v rows are written
// WRITE-THROUGH at the tree's index -- the reader is a descent wave on another CU, inside the same launch.
constexpr int C5_AL_STRIDE = 112, C5_AL_MASK = 80;
// The weight pointers are read through the CONSTANT address space wherever they are used (the stand-alone kernel: its own argument segment;
// the pipeline: its argument block in device memory) -- as a by-value struct inside the pipeline's persistent loop the 14 pointers stayed
// live in 28 scalar registers through the whole forward, and the scalar spills they caused became 48 spilled vector registers.
typedef const Conv5NetW __attribute__((address_space(4))) * Conv5NetWC;
#define N (*Np)
template <int NB, int A, int P, int SPLIT, bool IND>
__device__ __forceinline__ void conv5_net_body(float* smem, const Conv5NetWC Np, const int8_t* __restrict__ boards,
                                               const uint8_t* __restrict__ valid, int B, float* __restrict__ pi_out,
                                               float* __restrict__ v_out, float descale, const int wg, const int* sidx = nullptr,
                                               unsigned long long* smask = nullptr) {
    constexpr int NPL = SPLIT ? SPLIT : 3;                  // planes per tile: 3 = bf16 x 3, 2 = f16 x 2
    constexpr int NS = 8, ROWS = NS * 25, CS = 68, CP2 = 2, AS = (A + 3) / 4 * 4 + 4, PLANE_B = (ROWS + 2) * 128, TILE_B = NPL * PLANE_B;
    constexpr int AWI = (A + 63) / 64;
    if (NPL == 2) h2_fp16_saturate_mode();
    constexpr int LEAD = (NPL == 2 && SPLIT == 2) ? C5_LDS_LEAD : 0;      // (conv3x3_split: operand addresses reach six cells back)
    float* X = smem + LEAD / 4;             // [ROWS][CS]   (SPLIT: three bf16 planes, TILE_B bytes)
    float* Y = SPLIT ? (float*)((uint8_t*)X + TILE_B) : X + ROWS * CS;               // [ROWS][CS]
    int tid_ = threadIdx.x;
    if (IND) asm volatile("" : "+v"(tid_));            // (opaque inside the pipeline's persistent loop)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int b0 = wg * NS, nb = IND ? NS : min(NS, B - b0);
    bool heads_done = false;
    unsigned long long ind_mask = 0ull;
    if constexpr (IND) {
        if (tid < NS * AWI) {
            const int b = sidx[tid / AWI];
            if (b >= 0) ind_mask = __hip_atomic_load((const unsigned long long*)(valid + (size_t)b * C5_AL_STRIDE + C5_AL_MASK) + tid % AWI, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // f16 x 2 kernel: LDS copies of Wfp [50][A], Wf1 [25][64], Wp [64][2], Wv [64] behind the two tiles (the second is 64 KB)
    constexpr int WST_FP = CP2 * 25 * A, WST_F1 = WST_FP + 25 * 64, WST_P = WST_F1 + 64 * CP2, WST_N = WST_P + 64;
    float* const WST = (float*)((uint8_t*)X + TILE_B + 65536);
    C5_PH(0);
    // ---- board int8 [s][y][x][3] -> Y[s*25 + cell][plane 0..1], channels 2..15 zero (the first conv reads 16) ----
    for (int i = tid; i < ROWS * 4; i += 768) *(float4*)(Y + (i >> 2) * CS + 4 * (i & 3)) = make_float4(0.f, 0.f, 0.f, 0.f);
    // f16 x 2 kernel: the trunk's 2 NB bias vectors live in the lead bytes of the LDS (no operand address ever points there): a
    // convolution's epilogue reads its bias from LDS instead of waiting out an L2 round trip behind its last MFMA (the compiler kept the
    // global load below the main loop's scheduling barriers: 1-2 k cycles of every convolution)
    float* const BL = smem;
    static_assert(LEAD == 0 || (2 * NB + 1) * 64 * 4 <= LEAD, "bias vectors in the LDS lead");
    if (LEAD && tid < 2 * NB * 64) BL[tid] = N.bc[tid] * H2_AS;          // (64 b: the epilogue works in the planes' units)
    // ... and the first convolution's operands: its bias behind the trunk's, the 1152 weights that are not zero padding (of the 9216 of
    // the f32 fragment array) compactly behind the f32 staging tile -- requested here, they arrive while the boards are staged
    float* const W0L = Y + ROWS * CS;                        // [4 ct][9 taps][16][2]
    static_assert(!LEAD || (ROWS * CS + 4 * 9 * 16 * 2) * 4 <= 65536, "compact first-convolution weights behind the staging tile");
    if (LEAD && tid < 4 * 9 * 16) *(float2*)(W0L + 2 * tid) = *(const float2*)(N.W0 + (((size_t)(tid >> 4) * 64 + (tid & 15)) << 2));
    if (LEAD && tid >= 704 && tid < 768) BL[2 * NB * 64 + tid - 704] = N.b0[tid - 704];
    // (the small operands of the last phases -- head / FC biases, the value head's second matrix, the valid masks -- copied here as well:
    // 307.9 k -> 312.3 k cycles per launch, dropped: their global loads are not what the 1x1 heads / FC / softmax phases wait for)
    if constexpr (IND) { if (tid < NS * AWI) smask[tid] = ind_mask; }
    __syncthreads();
    constexpr bool CM = NPL == 2 && SPLIT == 2;                // cell-major tiles (row = cell * NS + sample) in the f16 x 2 kernel
    for (int i = tid; i < nb * 25 * 2; i += 768) {
        const int sc = i >> 1, pl = i & 1, smp = sc / 25, cell = sc - 25 * smp;        // (sample, cell) of the board byte
        if constexpr (IND) {
            const int b = sidx[smp];
            Y[(CM ? cell * NS + smp : sc) * CS + pl] =
                b >= 0 ? (float)(int8_t)__hip_atomic_load((const uint8_t*)boards + (size_t)b * C5_AL_STRIDE + cell * 3 + pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        } else
        Y[(CM ? cell * NS + smp : sc) * CS + pl] = (float)boards[(size_t)b0 * 75 + sc * 3 + pl];
    }
    __syncthreads();
    C5_PH(1);
    if (SPLIT) {
        uint8_t* XP = (uint8_t*)X;
        uint8_t* YP = (uint8_t*)Y;
        if (tid < NPL * 64) ((uint32_t*)(XP + (tid >> 6) * PLANE_B + ROWS * 128))[tid & 63] = 0u;    // X's zero rows
        constexpr size_t CONV_U4 = (size_t)4 * 18 * NPL * 64;   // uint4 per convolution
        uint4 wreg[6][3];                                        // NPL == 2: the trunk's weight fragments travel from one convolution
        if (NPL == 2) {                                          // to the next in registers (conv3x3_split, rolling prefetch)
            // requested behind the first convolution's own weights, landing during its tile loop: the first kernel row of the trunk's
            // weights, and the matrices of the heads and FCs (39 KB, kept in LDS behind the tiles: the phases at the end of the kernel
            // then never wait for a first touch of global memory)
            auto pf_lambda = [&]() {
                const int ct = wave & 3;
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) {
                    wreg[c6][0] = ((const uint4*)N.Wc)[(((size_t)ct * 18 + c6) * 2 + 0) * 64 + lane];
                    wreg[c6][1] = ((const uint4*)N.Wc)[(((size_t)ct * 18 + c6) * 2 + 1) * 64 + lane];
                    wreg[c6][2] = make_uint4(0u, 0u, 0u, 0u);
                }
                // global -> LDS DMA, 1 KiB per wave instruction (destination = uniform base + 16 * lane), no registers: waited for
                // (vmcnt) before the head phase only
                auto dma = [&](const float* src, int dst, int bytes) {
                    for (int c = wave; c * 1024 < bytes; c += 12) {
                        const int off = c * 1024 + lane * 16;
                        if (off < bytes)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const uint8_t*)src + off),
                                                             (__attribute__((address_space(3))) void*)((uint8_t*)(WST + dst) + c * 1024), 16, 0, 0);
                    }
                };
                dma(N.Wfp, 0, WST_FP * 4); dma(N.Wf1, WST_FP, (WST_F1 - WST_FP) * 4); dma(N.Wp, WST_F1, (WST_P - WST_F1) * 4);
                dma(N.Wv, WST_P, (WST_N - WST_P) * 4);
            };
            conv3x3_first_h2<NS, true, decltype(pf_lambda), CM, true>(W0L, BL + 2 * NB * 64, Y, XP, pf_lambda);
        } else conv3x3_first_split<NS, true, NPL>(N.W0, N.b0, Y, XP);   // (Y still holds the f32 board staging tile)
        __syncthreads();
        C5_PH(2);
        if (tid < NPL * 64) ((uint32_t*)(YP + (tid >> 6) * PLANE_B + ROWS * 128))[tid & 63] = 0u;    // Y's (the staging tile is dead)
#pragma unroll 1
        for (int blk = 0; blk < NB; blk++) {
            const uint4* W1 = (const uint4*)N.Wc + (size_t)(2 * blk) * CONV_U4;
            if (NPL == 2) {
                conv3x3_split<NS, NPL, true, CM>(W1, BL + (2 * blk) * 64, XP, YP, nullptr, descale, W1 + CONV_U4, &wreg);
                __syncthreads();
                conv3x3_split<NS, NPL, true, CM>(W1 + CONV_U4, BL + (2 * blk + 1) * 64, YP, XP, XP, descale,
                                             blk + 1 < NB ? W1 + 2 * CONV_U4 : nullptr, &wreg);
                __syncthreads();
                C5_PH(3 + blk);
                continue;
            }
            conv3x3_split<NS, NPL>(W1, N.bc + (2 * blk) * 64, XP, YP, nullptr, descale);
            __syncthreads();
            conv3x3_split<NS, NPL>(W1 + CONV_U4, N.bc + (2 * blk + 1) * 64, YP, XP, XP, descale);
            __syncthreads();
            C5_PH(3 + blk);
        }
        if (NPL == 2) {
            // f16 x 2: the 1x1 head convolutions run on the MFMAs straight from the planes (the Y tile is dead: head buffers go there)
            C5_PH(20);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the LDS copies of the head / FC matrices (DMA issued at the start)
            __syncthreads();
            heads1x1_h2<NS, CM, true>(WST + WST_F1, N.bp, WST + WST_P, N.bv, XP, Y, Y + NS * CP2 * 25);
            __syncthreads();
            heads_done = true;
        } else {
            // the heads read f32: rebuild the trunk output as [ROWS][CS] f32 at the start of the Y tile
            for (int i = tid; i < ROWS * 16; i += 768) {
                const int r = i >> 4, c4 = (i & 15) * 4;
                *(float4*)(Y + r * CS + c4) = load_split4(XP, PLANE_B, r, c4);
            }
            __syncthreads();
            C5_PH(20);
            X = Y;
            Y = Y + ROWS * CS;
        }
    } else {
        conv3x3_tile<1, NS>(N.W0, N.b0, Y, X, nullptr);
        __syncthreads();
#pragma unroll 1
        for (int blk = 0; blk < NB; blk++) {
            conv3x3_tile<4, NS>(N.Wc + (size_t)(2 * blk) * (9 * 64 * 64), N.bc + (2 * blk) * 64, X, Y, nullptr);
            __syncthreads();
            conv3x3_tile<4, NS>(N.Wc + (size_t)(2 * blk + 1) * (9 * 64 * 64), N.bc + (2 * blk + 1) * 64, Y, X, X);
            __syncthreads();
        }
    }
    // ---- heads (SimpleHead): 1x1 conv + BN + ReLU -> flatten (channel-major) -> FC ----
    float* HP = Y;                          // [NS][CP2*25]   policy head features
    float* HV = HP + NS * CP2 * 25;         // [NS][25]       value head features
    float* LG = HV + NS * 25;               // [NS][AS]       logits
    float* H1 = LG + NS * AS;               // [NS][64]       value fc1
    if (!heads_done)
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
    C5_PH(21);
    auto fcs = [&](const float* __restrict__ Wfp, const float* __restrict__ Wf1) {
        for (int i = tid; i < NS * A; i += 768) {
            const int s = i / A, a = i - s * A;
            float acc = N.bfp[a];
#pragma unroll 10
            for (int k = 0; k < CP2 * 25; k++) acc += HP[s * (CP2 * 25) + k] * Wfp[k * A + a];
            LG[s * AS + a] = acc;
        }
        for (int i = tid; i < NS * 64; i += 768) {
            const int s = i >> 6, j = i & 63;
            float acc = N.bf1[j];
            for (int k = 0; k < 25; k++) acc += HV[s * 25 + k] * Wf1[k * 64 + j];
            H1[s * 64 + j] = fmaxf(acc, 0.f);
        }
    };
    if constexpr (NPL == 2 && SPLIT == 2) {
        // f16 x 2 kernel: the policy FC with the 50 features split over four thread groups (thread = (K quarter, action), all NS samples of
        // a feature from two float4 reads of the feature-major HP): 39 LDS reads per thread instead of 200, partial sums combined after a barrier
        static_assert(NS == 8, "two float4 of samples per feature");
        constexpr int KF = CP2 * 25, KQ = (KF + 3) / 4;
        float* const RED = H1 + NS * 64;                     // [4][NS][AS]
        const float* const Wfp = WST;
        if (tid < 4 * A) {
            const int q = tid / A, a = tid - q * A, k1 = min(KF, (q + 1) * KQ);
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
            for (int k = q * KQ; k < k1; k++) {
                const float w = Wfp[k * A + a];
                a0 += *(const f32x4*)(HP + k * NS) * w;
                a1 += *(const f32x4*)(HP + k * NS + 4) * w;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) { RED[(q * NS + s4) * AS + a] = a0[s4]; RED[(q * NS + 4 + s4) * AS + a] = a1[s4]; }
        }
        const float* const Wf1 = WST + WST_FP;
        for (int i = tid; i < NS * 64; i += 768) {
            const int s = i >> 6, j = i & 63;
            float acc = N.bf1[j];
            for (int k = 0; k < 25; k++) acc += HV[s * 25 + k] * Wf1[k * 64 + j];
            H1[s * 64 + j] = fmaxf(acc, 0.f);
        }
        __syncthreads();
        for (int i = tid; i < NS * A; i += 768) {
            const int s = i / A, a = i - s * A;
            LG[s * AS + a] = N.bfp[a] + ((RED[s * AS + a] + RED[(NS + s) * AS + a]) + (RED[(2 * NS + s) * AS + a] + RED[(3 * NS + s) * AS + a]));
        }
    } else fcs(N.Wfp, N.Wf1);
    __syncthreads();
    C5_PH(22);
    // masked softmax == exp(log_softmax(where(valid, logits, -1e8))), one wave per sample
    for (int s = wave; s < nb; s += 12) {
        const int b = IND ? sidx[s] : b0 + s;
        if (IND && b < 0) continue;
        float x[(A + 63) / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < (A + 63) / 64; k++) {
            const int a = lane + 64 * k;
            x[k] = -INFINITY;
            if constexpr (IND) { if (a < A) x[k] = ((smask[s * AWI + k] >> lane) & 1ull) ? LG[s * AS + a] : -1e8f; }
            else if (a < A) x[k] = valid[(size_t)b * A + a] ? LG[s * AS + a] : -1e8f;
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
    if (tid < nb * P) {
        const int s = tid / P, p = tid - s * P;
        const int b = IND ? sidx[s] : b0 + s;
        if (!IND || b >= 0) {
            float acc = N.bf2[p];
            for (int j = 0; j < 64; j++) acc += H1[s * 64 + j] * N.Wf2[j * P + p];
            if constexpr (IND) __hip_atomic_store((uint32_t*)v_out + (size_t)b * P + p, __float_as_uint(tanhf(acc)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v_out[(size_t)b * P + p] = tanhf(acc);
        }
    }
    C5_PH(23);
}
#undef N

// SPLIT: the trunk on bf16 x 3 operands (above; N.Wc then points to the split fragments); LDS = 2 tiles x 3 planes x (ROWS + 1) x 128 B
template <int NB, int A, int P, int SPLIT = 0>
__global__ __launch_bounds__(768) void k_conv5_net(Conv5NetW N /* first argument: offset 0 of the kernel argument segment, read through it */,
                                                   const int8_t* __restrict__ boards,
                                                   const uint8_t* __restrict__ valid, int B, float* __restrict__ pi_out,
                                                   float* __restrict__ v_out, float descale) {
    extern __shared__ __attribute__((aligned(256))) float smem[];
    (void)N;
    conv5_net_body<NB, A, P, SPLIT, false>(smem, (Conv5NetWC)__builtin_amdgcn_kernarg_segment_ptr(), boards, valid, B, pi_out, v_out, descale, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The Santorini-with-gods net (nn_version 78, SantoriniNNet.py:167-192,264-271; HeadWithMeta :42-69): conv3x3(2->64, no BN,
// no activation) -> NB torchvision InvertedResidual blocks (1x1 expand 64->192 + BN + ReLU, depthwise 3x3 + BN + ReLU,
// 1x1 project 192->64 + BN, residual) -> 1x1-conv heads whose flattened features are concatenated with a 32-wide embedding
// of the gods / metadata plane (Linear(25,32) + ReLU) -> FC.  One launch; a workgroup owns 4 samples = 100 board cells:
// X [112][68] and the expanded tile H [112][196] live in LDS, the two 1x1 convolutions are MFMA GEMMs (mb_gemm), the
// depthwise convolution runs in place -- one thread owns the 5x5 plane of one (sample, channel) -- and the small heads run on
// the vector ALUs.  The 132 x 1782 policy FC is a second launch (k_s78_policy): with four samples per workgroup every
// workgroup streamed the 940 KB matrix for 0.9 MFLOP of work (20 % of the forward); the trunk leaves the 132 policy features
// of a sample at the start of its pi row, and k_s78_policy multiplies 16 samples at a time on the MFMAs, logits in LDS, masked
// softmax in place.  (As PyTorch ops MIOpen falls back to its naive kernel for the depthwise convolutions: 3 ms per batch of 1024.)
struct S78NetW {
    const float *W0;                  // first conv [9*16][64] fragment order, no bias
    const float *We, *be;             // NB x [64][192] fragment order, bias [NB][192]
    const float *Wd, *bd;             // NB x [192][9] plain (channel, tap = ky*3 + kx), bias [NB][192]
    const float *Wp, *bp;             // NB x [192][64] fragment order, bias [NB][64]
    const float *Wm, *bm;             // meta Linear [25][32], bias [32]
    const float *Whp, *bhp;           // policy 1x1 conv [64][4], bias [4]
    const float *Wfp, *bfp;           // policy FC [144][1792] zero padded, MFMA fragment order (rows: c*25 + cell, then the 32 meta features), bias [1792]
    const float *Whv, *bhv;           // value 1x1 conv [64][2], bias [2]
    const float *Wf1, *bf1;           // value fc1 [82][64], bias [64]
    const float *Wf2, *bf2;           // value fc2 [64][P], bias [P]
};

// The heads of the with-gods net on the f32 trunk output X [NS * 25][CS] (HeadWithMeta :42-69): 1x1 convolutions + ReLU, the
// flattened features joined by the 32 metadata features; the policy features go to the head of each sample's pi row
// (k_s78_policy), the value head (fc1 + ReLU, fc2, tanh) finishes here.  SCR: NS * (132 + 82 + 64) floats of scratch.
template <int NS, int A, int P>
__device__ __forceinline__ void s78_heads(const S78NetW& N, const float* X, const float* META, float* SCR, int b0, int nb,
                                          float* __restrict__ pi_out, float* __restrict__ v_out) {
    constexpr int CS = 68, CPI = 4, CV = 2, FP = CPI * 25 + 32, FV = CV * 25 + 32;
    const int tid = threadIdx.x;
    float* FEAT_P = SCR;                    // [NS][FP]   policy features: 4 x 25 conv outputs (channel-major) + meta
    float* FEAT_V = FEAT_P + NS * FP;       // [NS][FV]
    float* H1 = FEAT_V + NS * FV;           // [NS][64]
    for (int i = tid; i < NS * 25 * (CPI + CV); i += 768) {
        const int r = i / (CPI + CV), c = i - r * (CPI + CV);
        const float* xr = X + r * CS;
        float a = c < CPI ? N.bhp[c] : N.bhv[c - CPI];
#pragma unroll 8
        for (int k = 0; k < 64; k++) a += xr[k] * (c < CPI ? N.Whp[k * CPI + c] : N.Whv[k * CV + (c - CPI)]);
        a = fmaxf(a, 0.f);
        const int s = r / 25, cell = r - 25 * s;
        if (c < CPI) FEAT_P[s * FP + c * 25 + cell] = a; else FEAT_V[s * FV + (c - CPI) * 25 + cell] = a;
    }
    for (int i = tid; i < NS * 32; i += 768) {
        const int s = i >> 5, j = i & 31;
        FEAT_P[s * FP + CPI * 25 + j] = META[i];
        FEAT_V[s * FV + CV * 25 + j] = META[i];
    }
    __syncthreads();
    for (int i = tid; i < nb * FP; i += 768) {                 // the policy features go to the head of the sample's pi row (k_s78_policy)
        const int s = i / FP, k = i - s * FP;
        pi_out[(size_t)(b0 + s) * A + k] = FEAT_P[s * FP + k];
    }
    for (int i = tid; i < NS * 64; i += 768) {
        const int s = i >> 6, j = i & 63;
        float acc = N.bf1[j];
        for (int k = 0; k < FV; k++) acc += FEAT_V[s * FV + k] * N.Wf1[k * 64 + j];
        H1[s * 64 + j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (tid < nb * P) {
        const int s = tid / P, p = tid - s * P;
        float acc = N.bf2[p];
        for (int j = 0; j < 64; j++) acc += H1[s * 64 + j] * N.Wf2[j * P + p];
        v_out[(size_t)(b0 + s) * P + p] = tanhf(acc);
    }
}

// The same heads for the 8-sample kernels (round 6): the form above reads every weight from global memory inside its dot products (the
// 1x1 convolutions two loads per term, fc1 one) and ran 31 k of the trunk kernel's 360 k cycles (tools/dbg_nn_phases_s78.py).  Here the
// head weights are staged ONCE in LDS (WL: the X planes' region, dead after the f32 rebuild -- the caller has a barrier between), a
// thread of the 1x1 convolutions owns (row, two output channels: uniform per wave) and reads its weights as broadcast float2, fc1 keeps
// four partial sums, fc2 splits its K over four lanes; the weights are requested before the rebuild (s78_heads_prefetch).  WL: (64 * 6 + 8 + 82 * 64 + 64 + 64 * P + P) floats.
struct S78HeadPf { float wf1[7], w6, b6, bf1, wf2, bf2; };      // a thread's share of the head weights, requested ahead (s78_heads_prefetch)
template <int P>
__device__ __forceinline__ void s78_heads_prefetch(const S78NetW& N, S78HeadPf& pf) {
    constexpr int CPI = 4, CV = 2, FV = CV * 25 + 32;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 7; q++) pf.wf1[q] = tid + 768 * q < FV * 64 ? N.Wf1[tid + 768 * q] : 0.f;
    const int k = tid / 6, c = tid - 6 * k;
    pf.w6 = tid < 64 * 6 ? (c < CPI ? N.Whp[k * CPI + c] : N.Whv[k * CV + (c - CPI)]) : 0.f;
    pf.b6 = tid < 6 ? (tid < CPI ? N.bhp[tid] : N.bhv[tid - CPI]) : 0.f;
    pf.bf1 = tid < 64 ? N.bf1[tid] : 0.f;
    pf.wf2 = tid < 64 * P ? N.Wf2[tid] : 0.f;
    pf.bf2 = tid < P ? N.bf2[tid] : 0.f;
}
template <int NS, int A, int P>
__device__ __forceinline__ void s78_heads_lds(const S78HeadPf& pf, const float* X, const float* META, float* SCR, float* WL, int b0, int nb,
                                              float* __restrict__ pi_out, float* __restrict__ v_out) {
    constexpr int CS = 68, CPI = 4, CV = 2, FP = CPI * 25 + 32, FV = CV * 25 + 32, ROWS = NS * 25;
    static_assert(ROWS <= 256 && NS * 64 <= 768 && NS * P * 4 <= 64 && FV * 64 <= 7 * 768, "thread maps");
    const int tid = threadIdx.x;
    float* FEAT_P = SCR;                    // [NS][FP]
    float* FEAT_V = FEAT_P + NS * FP;       // [NS][FV]
    float* H1 = FEAT_V + NS * FV;           // [NS][64]
    float* W6 = WL;                         // [64][6]: policy channels 0..3, value channels 0..1
    float* B6 = W6 + 64 * 6;                // [6] (+ 2)
    float* WF1 = B6 + 8;                    // [FV][64]
    float* BF1 = WF1 + FV * 64;             // [64]
    float* WF2 = BF1 + 64;                  // [64][P]
    float* BF2 = WF2 + 64 * P;              // [P]
#pragma unroll
    for (int q = 0; q < 7; q++)
        if (tid + 768 * q < FV * 64) WF1[tid + 768 * q] = pf.wf1[q];
    if (tid < 64 * 6) W6[tid] = pf.w6;
    if (tid < 6) B6[tid] = pf.b6;
    if (tid < 64) BF1[tid] = pf.bf1;
    if (tid < 64 * P) WF2[tid] = pf.wf2;
    if (tid < P) BF2[tid] = pf.bf2;
    for (int i = tid; i < NS * 32; i += 768) {
        const int s = i >> 5, j = i & 31;
        FEAT_P[s * FP + CPI * 25 + j] = META[i];
        FEAT_V[s * FV + CV * 25 + j] = META[i];
    }
    C5_PH(14);
    __syncthreads();
    C5_PH(15);
    {                                       // 1x1 convolutions + ReLU: a wave's output pair is uniform (waves 0-3 / 4-7 / 8-11), a lane = a row
        const int og = __builtin_amdgcn_readfirstlane(tid >> 8), r = tid & 255;
        if (r < ROWS) {
            const float* xr = X + r * CS;
            const float* wg = W6 + 2 * og;  // (the same address in every lane: broadcast reads, requested in one batch)
            float a0 = B6[2 * og], a1 = B6[2 * og + 1], c0 = 0.f, c1 = 0.f;
#pragma unroll 1
            for (int kc = 0; kc < 64; kc += 16) {       // (16 terms per batch of reads: all 64 weight pairs at once were 128 registers -- spilled)
                float2 w[16];
                float4 x[4];
#pragma unroll
                for (int k = 0; k < 16; k++) w[k] = *(const float2*)(wg + 6 * (kc + k));
#pragma unroll
                for (int k = 0; k < 4; k++) x[k] = *(const float4*)(xr + kc + 4 * k);
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    a0 += x[k].x * w[4 * k].x; a1 += x[k].x * w[4 * k].y; c0 += x[k + 1].x * w[4 * k + 4].x; c1 += x[k + 1].x * w[4 * k + 4].y;
                    a0 += x[k].y * w[4 * k + 1].x; a1 += x[k].y * w[4 * k + 1].y; c0 += x[k + 1].y * w[4 * k + 5].x; c1 += x[k + 1].y * w[4 * k + 5].y;
                    a0 += x[k].z * w[4 * k + 2].x; a1 += x[k].z * w[4 * k + 2].y; c0 += x[k + 1].z * w[4 * k + 6].x; c1 += x[k + 1].z * w[4 * k + 6].y;
                    a0 += x[k].w * w[4 * k + 3].x; a1 += x[k].w * w[4 * k + 3].y; c0 += x[k + 1].w * w[4 * k + 7].x; c1 += x[k + 1].w * w[4 * k + 7].y;
                }
            }
            a0 = fmaxf(a0 + c0, 0.f); a1 = fmaxf(a1 + c1, 0.f);
            const int s = r / 25, cell = r - 25 * s;
            if (og < 2) { FEAT_P[s * FP + (2 * og) * 25 + cell] = a0; FEAT_P[s * FP + (2 * og + 1) * 25 + cell] = a1; }
            else { FEAT_V[s * FV + cell] = a0; FEAT_V[s * FV + 25 + cell] = a1; }
        }
    }
    C5_PH(16);
    __syncthreads();
    C5_PH(17);
    for (int i = tid; i < nb * FP; i += 768) {                 // the policy features go to the head of the sample's pi row (k_s78_policy)
        const int s = i / FP, k = i - s * FP;
        pi_out[(size_t)(b0 + s) * A + k] = FEAT_P[s * FP + k];
    }
    if (tid < NS * 64) {
        const int s = tid >> 6, j = tid & 63;
        const float* fv = FEAT_V + s * FV;
        float acc[4] = {BF1[j], 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < FV; k++) acc[k & 3] += fv[k] * WF1[k * 64 + j];
        H1[s * 64 + j] = fmaxf((acc[0] + acc[1]) + (acc[2] + acc[3]), 0.f);
    }
    C5_PH(18);
    __syncthreads();
    C5_PH(19);
    if (tid < NS * P * 4) {                 // fc2 + tanh: (sample, player) x four lanes over K
        const int q = tid & 3, sp = tid >> 2, s = sp / P, p = sp - s * P;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) acc += H1[s * 64 + 16 * q + j] * WF2[(16 * q + j) * P + p];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (q == 0 && s < nb) v_out[(size_t)(b0 + s) * P + p] = tanhf(acc + BF2[p]);
    }
}

template <int NB, int A, int P>
__global__ __launch_bounds__(768) void k_s78_net(S78NetW N, const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid,
                                                 int B, float* __restrict__ pi_out, float* __restrict__ v_out) {
    constexpr int NS = 4, ROWS = NS * 25, RT = (ROWS + 15) / 16, ROWSP = RT * 16, CS = 68, HS = 196, E = 192, NW = 12;
    constexpr int CPI = 4, CV = 2, FP = CPI * 25 + 32, FV = CV * 25 + 32, AS = (A + 3) / 4 * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                        // [ROWSP][CS]
    float* H = X + ROWSP * CS;              // [ROWSP][HS]
    float* META = H + ROWSP * HS;           // [NS][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int b0 = blockIdx.x * NS, nb = min(NS, B - b0);
    for (int i = tid; i < (ROWSP * CS + ROWSP * HS + NS * 32) / 4; i += 768) ((float4*)smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- planes 0, 1 -> H[cell][0..1] (row stride CS inside the H region, channels 2..15 stay zero); plane 2 -> meta ----
    float* IN0 = H;
    for (int i = tid; i < nb * 25 * 2; i += 768) {
        const int r = i >> 1, pl = i & 1;
        IN0[r * CS + pl] = (float)boards[(size_t)b0 * 75 + r * 3 + pl];
    }
    for (int i = tid; i < nb * 32; i += 768) {
        const int s = i >> 5, j = i & 31;
        float acc = N.bm[j];
        for (int k = 0; k < 25; k++) acc += (float)boards[(size_t)(b0 + s) * 75 + k * 3 + 2] * N.Wm[k * 32 + j];
        META[s * 32 + j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    {
        __shared__ float zero_bias[64];
        if (tid < 64) zero_bias[tid] = 0.f;
        __syncthreads();
        conv3x3_tile<1, NS, false>(N.W0, zero_bias, IN0, X, nullptr);
    }
    __syncthreads();
    for (int i = tid; i < ROWS * 4; i += 768) *(float4*)(IN0 + (i >> 2) * CS + 4 * (i & 3)) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
#pragma unroll 1
    for (int blk = 0; blk < NB; blk++) {
        const float* be = N.be + blk * E;
        // ---- 1x1 expand + BN + ReLU -> H ----
        mb_gemm<4, E / 16, RT, NW>(
            N.We + (size_t)blk * (64 * E), [&](int rt, int c) { return *(const float4*)(X + (rt * 16 + r16) * CS + 16 * c + 4 * g); },
            [&](int ct, int rt, f32x4 acc) {
                const float4 b = *(const float4*)(be + ct * 16 + 4 * g);
                *(float4*)(H + (rt * 16 + r16) * HS + ct * 16 + 4 * g) =
                    make_float4(fmaxf(acc[0] + b.x, 0.f), fmaxf(acc[1] + b.y, 0.f), fmaxf(acc[2] + b.z, 0.f), fmaxf(acc[3] + b.w, 0.f));
            });
        __syncthreads();
        // ---- depthwise 3x3 + BN + ReLU, in place: one thread = the 5x5 plane of one (sample, channel) ----
        for (int i = tid; i < NS * E; i += 768) {
            const int s = i / E, c = i - s * E;
            float* base = H + (s * 25) * HS + c;
            float in[25], w[9];
#pragma unroll
            for (int k = 0; k < 25; k++) in[k] = base[k * HS];
#pragma unroll
            for (int k = 0; k < 9; k++) w[k] = N.Wd[((size_t)blk * E + c) * 9 + k];
            const float bias = N.bd[blk * E + c];
#pragma unroll
            for (int y = 0; y < 5; y++)
#pragma unroll
                for (int x = 0; x < 5; x++) {
                    float a = bias;
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const int yy = y + ky - 1, xx = x + kx - 1;
                            if (yy >= 0 && yy < 5 && xx >= 0 && xx < 5) a += w[ky * 3 + kx] * in[yy * 5 + xx];
                        }
                    base[(y * 5 + x) * HS] = fmaxf(a, 0.f);
                }
        }
        __syncthreads();
        // ---- 1x1 project + BN + residual -> X (in place: a lane reads and writes its own elements) ----
        const float* bp = N.bp + blk * 64;
        mb_gemm<E / 16, 4, RT, NW>(
            N.Wp + (size_t)blk * (E * 64), [&](int rt, int c) { return *(const float4*)(H + (rt * 16 + r16) * HS + 16 * c + 4 * g); },
            [&](int ct, int rt, f32x4 acc) {
                float* xp = X + (rt * 16 + r16) * CS + ct * 16 + 4 * g;
                const float4 b = *(const float4*)(bp + ct * 16 + 4 * g), x = *(const float4*)xp;
                *(float4*)xp = make_float4(acc[0] + b.x + x.x, acc[1] + b.y + x.y, acc[2] + b.z + x.z, acc[3] + b.w + x.w);
            });
        __syncthreads();
    }
    s78_heads<NS, A, P>(N, X, META, H, b0, nb, pi_out, v_out);
}

// The same trunk on split-precision operands (bf16 x 3, see conv3x3_split): a workgroup owns 8 samples = 200 cells; X [200][64] and
// one THIRD of the expanded tile (64 of the 192 channels) live in LDS as three bf16 planes each (2 x 77 KB, the layout of
// k_conv5_net<SPLIT>).  InvertedResidual has no squeeze-excite, so a block runs as three passes
//   expand 64 -> 64 (third t of We) + BN + ReLU -> H;  depthwise 3x3 + BN + ReLU on H in place;  project acc += H * Wp[third t]
// with the project accumulators kept in registers across the passes; every GEMM is the 64 x 64 gemm64_split (six bf16 MFMAs per
// product).  N.We / N.Wp point to the split fragments [NB][3 thirds][4 ct][2 chunks][3 planes][64 lanes][8] bf16.
// NPL = 2: f16 x 2 operands (hi + lo, three MFMAs per product, two planes per tile holding 64 * x; N.We / N.Wp then hold
// [NB][3 thirds][4 ct][2 chunks][2 planes][64 lanes][8] f16 of W * 2^k, ds_e / ds_p = 2^-k / 64 of the two matrix families).
// NPL = 2 since round 6: LDS = X planes (52 KB) | the expanded third as an F32 tile [200][64], 16-byte chunks swizzled by the row (51 KB) |
// the two f16 planes of the depthwise pass's output (52 KB) | meta features.  A pass: expand GEMM (two column tiles per wave,
// gemm64_split2) -> f32 tile; barrier; depthwise 3x3 (a thread = a channel pair x a group of output rows; reads the f32 tile, writes the
// planes); barrier; project GEMM from the planes -- and straight on into the next pass's expand, which touches neither: 7 barriers per block.
#ifndef AZG_S78_W2
#define AZG_S78_W2 1                        /* 0: one column tile per wave (4 x 3), the form of rounds 2-5 */
#endif
template <int NB, int A, int P, int NPL = 3, int NS = 8>
__global__ __launch_bounds__(768) void k_s78_net_split(S78NetW N, const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid,
                                                       int B, float* __restrict__ pi_out, float* __restrict__ v_out, float ds_e, float ds_p) {
    constexpr bool W2 = NPL == 2 && AZG_S78_W2;            // two column tiles per wave (gemm64_split2): 2 column-tile pairs x 6 row groups
    constexpr int ROWS = NS * 25, RT = (ROWS + 15) / 16, RG = W2 ? 6 : 3, MAXT = (RT + RG - 1) / RG, CTW = W2 ? 2 : 1, CS = 68, E = 192;
    // (the f32 staging / head tile [ROWS][CS] + head buffers live in the H region: it keeps the size of three planes)
    // (NPL = 2: the H region holds the f32 expanded tile AND, behind it, the two f16 planes the depthwise pass writes for the project GEMM)
    constexpr int PLANE_B = (ROWS + 2) * 128, TILE_B = NPL * PLANE_B, HREG_B = NPL == 2 ? 2 * TILE_B : 3 * PLANE_B;
    constexpr size_t G64_U4 = (size_t)4 * 2 * NPL * 64;        // uint4 per 64 x 64 matrix
    extern __shared__ __attribute__((aligned(256))) float smem[];
    if (NPL == 2) h2_fp16_saturate_mode();
    uint8_t* XP = (uint8_t*)smem;                               // X: three bf16 planes [ROWS + 1][64]
    uint8_t* HP = XP + TILE_B;                                  // one third of the expanded tile, same layout
    float* META = (float*)(HP + HREG_B);                        // [NS][32]
    float* STG = (float*)HP;                                    // f32 [ROWS][CS]: the board staging tile, later the trunk output for the heads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int ct = W2 ? 2 * (wave & 1) : (wave & 3), rg = W2 ? (wave >> 1) : (wave >> 2);        // (W2: the first of the wave's two column tiles)
    const int b0 = blockIdx.x * NS, nb = min(NS, B - b0);
    C5_PH(8);
    for (int i = tid; i < ROWS * 4; i += 768) *(float4*)(STG + (i >> 2) * CS + 4 * (i & 3)) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < NPL * 64) ((uint32_t*)(XP + (tid >> 6) * PLANE_B + ROWS * 128))[tid & 63] = 0u;      // X's zero row
    __syncthreads();
    for (int i = tid; i < nb * 25 * 2; i += 768) {
        const int r = i >> 1, pl = i & 1;
        STG[r * CS + pl] = (float)boards[(size_t)b0 * 75 + r * 3 + pl];
    }
    for (int i = tid; i < NS * 32; i += 768) {
        const int s = i >> 5, j = i & 31;
        float acc = N.bm[j];
        if (s < nb)
            for (int k = 0; k < 25; k++) acc += (float)boards[(size_t)(b0 + s) * 75 + k * 3 + 2] * N.Wm[k * 32 + j];
        META[s * 32 + j] = fmaxf(acc, 0.f);
    }
    __shared__ float zero_bias[64];
    if (tid < 64) zero_bias[tid] = 0.f;
    __syncthreads();
    C5_PH(9);
    if (NPL == 2) conv3x3_first_h2<NS, false>(N.W0, zero_bias, STG, XP);
    else conv3x3_first_split<NS, false, NPL>(N.W0, zero_bias, STG, XP);
    __syncthreads();
    C5_PH(10);
    uint8_t* const HPL = NPL == 2 ? HP + TILE_B : HP;           // the planes the project GEMM reads (NPL = 2: a buffer of their own)
    if (tid < NPL * 64) ((uint32_t*)(HPL + (tid >> 6) * PLANE_B + ROWS * 128))[tid & 63] = 0u;      // H's zero row (the staging tile is dead)
    // The phases of a pass (third t of block blk), as functions of the H buffer they work on:
    f32x2 dw_w[9], dw_b;                    // (NPL = 2) the depthwise weights of this thread's channel pair, requested a phase ahead
    f32x4 pacc[MAXT * CTW];
#define S78_PH(k) do { if (blk == 5 && t == 1) C5_PH(k); } while (0)
    // ---- 1x1 expand (channels 64 t .. 64 t + 63) + BN + ReLU -> Hc ----
    auto expand = [&](int blk, int t, uint8_t* Hc) {
        f32x4 e[MAXT * CTW];
#pragma unroll
        for (int i = 0; i < MAXT * CTW; i++) e[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (W2) {
            uint4 we[8];
            gemm64_wload2((const uint4*)N.We + (size_t)(blk * 3 + t) * G64_U4, we);
            gemm64_split2<NS>(we, XP, e);
        } else {
            uint4 we[6];
            gemm64_wload<NPL>((const uint4*)N.We + (size_t)(blk * 3 + t) * G64_U4, we);
            gemm64_split<NS, NPL>(we, XP, e);
        }
        // (measured and dropped, round 6: the GEMMs' weight fragments requested a phase ahead -- the project's under the depthwise
        // convolution: wave 0's project phase 1.7 k -> 1.0 k cycles, the forward 367 -> 379 us; the expand's under the project
        // GEMM of the pass before: nothing, that phase is bound by the MFMA pipe, its LDS operand reads and its epilogue)
        if constexpr (NPL == 2) {           // the depthwise weights: requested here, used behind the barrier
            const int ec = blk * E + t * 64 + 2 * (tid & 31);
            const float* wd = N.Wd + (size_t)ec * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) dw_w[k] = f32x2{wd[k], wd[9 + k]};
            dw_b = f32x2{N.bd[ec], N.bd[ec + 1]};
        }
        const float4 b = *(const float4*)(N.be + blk * E + t * 64 + ct * 16 + 4 * g);
        float4 b2 = b;
        if (W2) b2 = *(const float4*)(N.be + blk * E + t * 64 + (ct + 1) * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < MAXT; i++) {
            const int r = (rg + RG * i) * 16 + r16;
            if (rg + RG * i >= RT || r >= ROWS) continue;
            if (NPL == 2) {
                // (the expanded tile goes to the depthwise pass as F32 -- [row][64 channels], 16-byte chunks swizzled by the row, in the planes'
                // units 64 * x -- over the same bytes the two f16 planes take afterwards: the depthwise pass reads all of it before it writes
                // the planes.  Splitting here and joining there again was 10 of this epilogue's 14 and 40 of that pass's 165 vector instructions.)
                const float se = ds_e * H2_AS;
                const f32x4 o = e[i * CTW] * se + f32x4{b.x, b.y, b.z, b.w} * H2_AS;
                *(float4*)(Hc + r * 256 + (((4 * ct + g) ^ (r & 7)) << 4)) = make_float4(fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f));
                if (W2) {
                    const f32x4 o2 = e[i * CTW + CTW - 1] * se + f32x4{b2.x, b2.y, b2.z, b2.w} * H2_AS;
                    *(float4*)(Hc + r * 256 + (((4 * (ct + 1) + g) ^ (r & 7)) << 4)) =
                        make_float4(fmaxf(o2[0], 0.f), fmaxf(o2[1], 0.f), fmaxf(o2[2], 0.f), fmaxf(o2[3], 0.f));
                }
                continue;
            }
            store_split4(Hc, PLANE_B, r, ct * 16 + 4 * g, make_float4(fmaxf(e[i][0] + b.x, 0.f), fmaxf(e[i][1] + b.y, 0.f),
                                                                       fmaxf(e[i][2] + b.z, 0.f), fmaxf(e[i][3] + b.w, 0.f)));
        }
    };
    auto depthwise = [&](int blk, int t, uint8_t* Hc, uint8_t* Hd) {       // (Hd: where the planes go; NPL = 3: Hd == Hc, in place)
        (void)blk; (void)t; (void)Hd;
        // ---- depthwise 3x3 + BN + ReLU in place: one thread = the 5x5 plane of one (sample, channel) ----
        // (round 4, f16 x 2 kernel, measured and dropped: one thread per channel PAIR -- dword accesses to the planes, packed f32
        // multiply-adds, 256 threads instead of 512 -- 415 -> 419 us per 4096 leaves; the GEMM phases' weight fragments and biases
        // requested one phase ahead -- 411-419 -> 408-415 us, 48 B of spills: neither the 16-bit LDS accesses nor the 60 weight
        // round trips are what this kernel waits for)
        if constexpr (NPL == 2) {
            // One thread = a PAIR of channels (one dword of a plane row: packed f32 multiply-adds, half the LDS accesses) x a group of
            // output rows (0-1 / 2-3 / 4; wave-uniform), holding the <= 4 input rows it needs: 3 x NS x 32 threads, i.e. every wave of an
            // 8-sample workgroup -- the form with one thread per plane ran 650 vector instructions on 8 of the 12 waves (5.1 k of a
            // pass's 11 k cycles, tools/dbg_nn_phases_s78.py).  The planes hold 64 x: with the bias scaled the sums come out as 64 x too.
            // Row r0 + cell of sample s sits in swizzle class (s + cell) & 7 (25 = 1 mod 8): eight base addresses per thread, the cell
            // itself is an immediate offset.  In place: every read is done before the first write (the barrier in the middle).
            constexpr int PT = NS * 32;
            const int part = tid / PT, s_ = (tid >> 5) % NS, pr = tid & 31;
            int fbase[8];                    // f32 tile (read) and planes (written, below): both swizzled by the row class (s + cell) & 7
#pragma unroll
            for (int j = 0; j < 8; j++) fbase[j] = s_ * 25 * 256 + (((pr >> 1) ^ ((s_ + j) & 7)) << 4) + ((pr & 1) << 3);
            f32x2 out[10];
            const f32x2 bias = dw_b * H2_AS;
            const f32x2 (&w)[9] = dw_w;
            auto rows = [&](auto y0c, auto nyc) {
                constexpr int Y0 = decltype(y0c)::value, NY = decltype(nyc)::value, I0 = Y0 > 0 ? Y0 - 1 : 0, I1 = Y0 + NY < 5 ? Y0 + NY : 4;
                f32x2 in[(I1 - I0 + 1) * 5];
#pragma unroll
                for (int k = I0 * 5; k < (I1 + 1) * 5; k++) {
                    in[k - I0 * 5] = *(const f32x2*)(Hc + fbase[k & 7] + k * 256);
                }
                // (taps outside, outputs inside: ten independent accumulators between two uses of one -- a dependent packed
                // multiply-add needs a wait state, and output by output the compiler emitted one s_nop per multiply-add)
#pragma unroll
                for (int o = 0; o < NY * 5; o++) out[o] = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
#pragma unroll
                        for (int y = Y0; y < Y0 + NY; y++)
#pragma unroll
                            for (int x = 0; x < 5; x++) {
                                const int yy = y + ky - 1, xx = x + kx - 1;
                                if (yy >= 0 && yy < 5 && xx >= 0 && xx < 5) out[(y - Y0) * 5 + x] += w[ky * 3 + kx] * in[(yy - I0) * 5 + xx];
                            }
#pragma unroll
                for (int o = 0; o < NY * 5; o++) out[o] = f32x2{fmaxf(out[o].x, 0.f), fmaxf(out[o].y, 0.f)};
            };
            if (part == 0) rows(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            else if (part == 1) rows(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
            else if (part == 2) rows(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
            int base[8];                     // (the planes are a buffer of their own: no barrier between the reads above and the writes below)
#pragma unroll
            for (int j = 0; j < 8; j++) base[j] = s_ * 25 * 128 + (((pr >> 2) ^ ((s_ + j) & 7)) << 4) + ((pr & 3) << 2);
            auto put = [&](auto y0c, auto nyc) {
                constexpr int Y0 = decltype(y0c)::value, NY = decltype(nyc)::value;
#pragma unroll
                for (int k = 0; k < NY * 5; k++) {
                    uint32_t h, l;
                    h2_split2(out[k].x, out[k].y, h, l);
                    uint8_t* dst = Hd + base[(Y0 * 5 + k) & 7] + (Y0 * 5 + k) * 128;
                    *(uint32_t*)dst = h; *(uint32_t*)(dst + PLANE_B) = l;
                }
            };
            if (part == 0) put(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            else if (part == 1) put(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
            else if (part == 2) put(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
        } else if (tid < NS * 64) {
            const int s = tid >> 6, c = tid & 63, ec = blk * E + t * 64 + c;
            float in[25], w[9];
            const int q = c >> 3, cb = (c & 7) << 1, r0 = s * 25;
            auto off = [&](int k) { return (r0 + k) * 128 + ((q ^ ((r0 + k) & 7)) << 4) + cb; };
#pragma unroll
            for (int k = 0; k < 25; k++) {
                const uint8_t* src = Hc + off(k);
                if (NPL == 2) in[k] = ((float)*(const _Float16*)src + (float)*(const _Float16*)(src + PLANE_B)) * H2_IAS;
                else in[k] = (bf16_lo_f32(*(const uint16_t*)src) + bf16_lo_f32(*(const uint16_t*)(src + PLANE_B))) +
                             bf16_lo_f32(*(const uint16_t*)(src + 2 * PLANE_B));
            }
#pragma unroll
            for (int k = 0; k < 9; k++) w[k] = N.Wd[(size_t)ec * 9 + k];
            const float bias = N.bd[ec];
            float out[26];
#pragma unroll
            for (int y = 0; y < 5; y++)
#pragma unroll
                for (int x = 0; x < 5; x++) {
                    float a = bias;
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const int yy = y + ky - 1, xx = x + kx - 1;
                            if (yy >= 0 && yy < 5 && xx >= 0 && xx < 5) a += w[ky * 3 + kx] * in[yy * 5 + xx];
                        }
                    out[y * 5 + x] = fmaxf(a, 0.f);
                }
            out[25] = 0.f;
#pragma unroll
            for (int k = 0; k < 26; k += 2) {                  // two cells per conversion
                uint32_t h, m, l;
                if (NPL == 2) {
                    h2_split2(out[k] * H2_AS, out[k + 1] * H2_AS, h, m);
                    uint8_t* e0 = Hc + off(k);
                    *(uint16_t*)e0 = (uint16_t)h; *(uint16_t*)(e0 + PLANE_B) = (uint16_t)m;
                    if (k + 1 < 25) {
                        uint8_t* e1 = Hc + off(k + 1);
                        *(uint16_t*)e1 = (uint16_t)(h >> 16); *(uint16_t*)(e1 + PLANE_B) = (uint16_t)(m >> 16);
                    }
                    continue;
                }
                split3x2(out[k], out[k + 1], h, m, l);
                uint8_t* d0 = Hc + off(k);
                *(uint16_t*)d0 = (uint16_t)h; *(uint16_t*)(d0 + PLANE_B) = (uint16_t)m; *(uint16_t*)(d0 + 2 * PLANE_B) = (uint16_t)l;
                if (k + 1 < 25) {
                    uint8_t* d1 = Hc + off(k + 1);
                    *(uint16_t*)d1 = (uint16_t)(h >> 16); *(uint16_t*)(d1 + PLANE_B) = (uint16_t)(m >> 16);
                    *(uint16_t*)(d1 + 2 * PLANE_B) = (uint16_t)(l >> 16);
                }
            }
        }
    };
    // ---- 1x1 project, K = this third of the expanded channels ----
    auto project = [&](int blk, int t, const uint8_t* Hc) {
        if constexpr (W2) {
            uint4 wp[8];
            gemm64_wload2((const uint4*)N.Wp + (size_t)(blk * 3 + t) * G64_U4, wp);
            gemm64_split2<NS>(wp, Hc, pacc);
        } else {
            uint4 wp[6];
            gemm64_wload<NPL>((const uint4*)N.Wp + (size_t)(blk * 3 + t) * G64_U4, wp);
            gemm64_split<NS, NPL>(wp, Hc, pacc);
        }
    };
    auto residual = [&](int blk) {
    // ---- + BN bias + residual -> X, in place (a lane reads and writes its own elements) ----
    const float4 b = *(const float4*)(N.bp + blk * 64 + ct * 16 + 4 * g);
    float4 b2 = b;
    if (W2) b2 = *(const float4*)(N.bp + blk * 64 + (ct + 1) * 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int r = (rg + RG * i) * 16 + r16;
        if (rg + RG * i >= RT || r >= ROWS) continue;
        if (NPL == 2) {
            h2_store4(XP, PLANE_B, 128, r, ct * 16 + 4 * g, pacc[i * CTW] * ds_p + f32x4{b.x, b.y, b.z, b.w} + h2_load4(XP, PLANE_B, 128, r, ct * 16 + 4 * g));
            h2_store4(XP, PLANE_B, 128, r, (ct + 1) * 16 + 4 * g,
                      pacc[i * CTW + CTW - 1] * ds_p + f32x4{b2.x, b2.y, b2.z, b2.w} + h2_load4(XP, PLANE_B, 128, r, (ct + 1) * 16 + 4 * g));
            continue;
        }
        const float4 x = load_split4(XP, PLANE_B, r, ct * 16 + 4 * g);
        store_split4(XP, PLANE_B, r, ct * 16 + 4 * g, make_float4(pacc[i][0] + b.x + x.x, pacc[i][1] + b.y + x.y,
                                                                   pacc[i][2] + b.z + x.z, pacc[i][3] + b.w + x.w));
    }
    };
#pragma unroll 1
    for (int blk = 0; blk < NB; blk++) {
#pragma unroll
        for (int i = 0; i < MAXT * CTW; i++) pacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (measured and dropped, round 6: two H buffers -- the project GEMM of third t and the expand GEMM of third t + 1 as ONE phase, seven
        // barriers per block instead of ten: 333 -> 341-345 us per 4096 leaves on the same box, 17 spilled registers)
        {
#pragma unroll 1
            for (int t = 0; t < 3; t++) {
                S78_PH(0);
                expand(blk, t, HP);
                S78_PH(1);
                __syncthreads();
                S78_PH(2);
                depthwise(blk, t, HP, HPL);
                S78_PH(3);
                __syncthreads();
                S78_PH(4);
                project(blk, t, HPL);
                S78_PH(5);
                // (NPL = 2: no barrier here -- the next expand writes the f32 tile and reads X, the project GEMM read the planes; the next
                // depthwise pass, which overwrites the planes, starts behind the barrier that follows that expand)
                if (NPL != 2) __syncthreads();
                S78_PH(6);
            }
            // (X is rewritten below, each lane its own elements: the block's last expand, which read all of X, ran before the depthwise barrier)
            residual(blk);
            __syncthreads();
        }
    }
    C5_PH(11);
    S78HeadPf hpf;
    s78_heads_prefetch<P>(N, hpf);           // (lands under the rebuild and its barrier)
    // ---- the heads read f32: rebuild the trunk output as [ROWS][CS] f32 over the H region ----
    for (int i = tid; i < ROWS * 16; i += 768) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        if (NPL == 2) { const f32x4 o = h2_load4(XP, PLANE_B, 128, r, c4); *(float4*)(STG + r * CS + c4) = make_float4(o[0], o[1], o[2], o[3]); }
        else *(float4*)(STG + r * CS + c4) = load_split4(XP, PLANE_B, r, c4);
    }
    __syncthreads();
    C5_PH(12);
    static_assert((size_t)(64 * 6 + 8 + 82 * 64 + 64 + 64 * P + P) * 4 <= (size_t)TILE_B, "the head weights fit the X planes' region");
    s78_heads_lds<NS, A, P>(hpf, STG, META, STG + ROWS * CS, (float*)XP, b0, nb, pi_out, v_out);
    C5_PH(13);
}

// Policy FC + masked softmax of the with-gods net for 16 samples per workgroup: logits[s][a] = bfp[a] + sum_k feat[s][k] Wfp[k][a]
// (HeadWithMeta :62-69), pi = exp(log_softmax(where(valid, logits, -1e8))).  feat = the first FP floats of each pi row (written by
// k_s78_net), K padded to 144.  The activation fragments of the 16 samples stay in registers, the 12 waves stream the weight
// fragments of their column tiles (1 KB per K chunk), logits live in LDS [16][LS].
template <int A, int FP>
__global__ __launch_bounds__(768) void k_s78_policy(const float* __restrict__ Wfrag, const float* __restrict__ bias,
                                                    const uint8_t* __restrict__ valid, int B, float* __restrict__ pi) {
    constexpr int KCH = (FP + 15) / 16, KP = KCH * 16, FS = KP + 4, NT = (A + 15) / 16, LS = NT * 16 + 4, NW = 12;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* FEAT = smem;                     // [16][FS]
    float* LG = FEAT + 16 * FS;             // [16][LS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int b0 = blockIdx.x * 16, nb = min(16, B - b0);
    for (int i = tid; i < 16 * FS; i += 768) {
        const int s = i / FS, k = i - s * FS;
        FEAT[i] = (s < nb && k < FP) ? pi[(size_t)(b0 + s) * A + k] : 0.f;
    }
    __syncthreads();
    float4 a[KCH];
#pragma unroll
    for (int c = 0; c < KCH; c++) a[c] = *(const float4*)(FEAT + r16 * FS + 16 * c + 4 * g);
    // a wave's column tiles ct = wave, wave + NW, ...: the fragments of the next tile are requested before the MFMAs of this one
    auto wload = [&](int ct, float4* w) {
#pragma unroll
        for (int c = 0; c < KCH; c++) w[c] = FRAG(Wfrag, KCH, ct < NT ? ct : 0, c);
    };
    auto tile = [&](int ct, const float4* w) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KCH; c++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, a[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].y, a[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].z, a[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].w, a[c].w, acc, 0, 0, 0);
        }
        const float4 b = *(const float4*)(bias + ct * 16 + 4 * g);
        *(float4*)(LG + r16 * LS + ct * 16 + 4 * g) = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
    };
    float4 w0[KCH], w1[KCH];
    wload(wave, w0);
#pragma unroll 1
    for (int ct = wave; ct < NT; ct += 2 * NW) {
        wload(ct + NW, w1);
        tile(ct, w0);
        if (ct + NW < NT) {
            wload(ct + 2 * NW, w0);
            tile(ct + NW, w1);
        }
    }
    __syncthreads();
    for (int s = wave; s < nb; s += NW) {                      // masked softmax, one wave per sample
        const int b = b0 + s;
        constexpr int NK = (A + 63) / 64;
        float x[NK];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int ai = lane + 64 * k;
            x[k] = -INFINITY;
            if (ai < A) x[k] = valid[(size_t)b * A + ai] ? LG[s * LS + ai] : -1e8f;
            mx = fmaxf(mx, x[k]);
        }
        mx = nn_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NK; k++) { x[k] = (lane + 64 * k < A) ? expf(x[k] - mx) : 0.f; sum += x[k]; }
        sum = nn_wave_sum(sum);
#pragma unroll
        for (int k = 0; k < NK; k++)
            if (lane + 64 * k < A) pi[(size_t)b * A + lane + 64 * k] = x[k] / sum;
    }
}

// The same FC on f16 x 2 split-precision operands (azg_nn_s78_forward_h2, round 6): the f32-input MFMA runs at 1/16 of the 16-bit rate
// and the FC was bound by it (112 column tiles x 36 v_mfma_f32_16x16x4f32 per 16 samples); here a column tile is 5 K chunks of 32 x three
// v_mfma_f32_16x16x32_f16 (lo*hi, hi*lo, hi*hi: h2_mma).  Wfrag: [112 ct][5 chunks][2 planes hi, lo][64 lanes][8] f16 of Wfp * 2^k
// (K 132 -> 160, N 1782 -> 1792, zero padded; element = W_plane[32*chunk + 8*(lane>>4) + j][16*ct + (lane&15)]) followed by ONE float:
// the descale 2^-k / 64 (the feature operand holds 64 * x like every activation plane).  The 16 samples' feature fragments stay in
// registers, a wave streams the fragments of its column tiles two tiles ahead of the MFMAs.
template <int A, int FP>
__global__ __launch_bounds__(768) void k_s78_policy_h2(const uint4* __restrict__ Wfrag, const float* __restrict__ bias,
                                                       const uint8_t* __restrict__ valid, int B, float* __restrict__ pi) {
    constexpr int KCH = (FP + 31) / 32, KP = KCH * 32, FS = KP + 4, NT = (A + 15) / 16, LS = NT * 16 + 4, NW = 12;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* FEAT = smem;                     // [16][FS]
    float* LG = FEAT + 16 * FS;             // [16][LS]
    h2_fp16_saturate_mode();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int b0 = blockIdx.x * 16, nb = min(16, B - b0);
    const float descale = *(const float*)(Wfrag + (size_t)NT * KCH * 2 * 64);
    auto wload = [&](int ct, uint4* w) {
#pragma unroll
        for (int c = 0; c < 2 * KCH; c++) w[c] = Wfrag[((size_t)(ct < NT ? ct : 0) * (2 * KCH) + c) * 64 + lane];
    };
    uint4 w0[2 * KCH], w1[2 * KCH];
    wload(wave, w0);                        // (under the feature staging)
    wload(wave + NW, w1);
    for (int i = tid; i < 16 * FS; i += 768) {
        const int s = i / FS, k = i - s * FS;
        FEAT[i] = (s < nb && k < FP) ? pi[(size_t)(b0 + s) * A + k] : 0.f;
    }
    __syncthreads();
    uint4 ah[KCH], al[KCH];                 // sample r16, k = 32 c + 8 g .. + 7
#pragma unroll
    for (int c = 0; c < KCH; c++) {
        const float4 x0 = *(const float4*)(FEAT + r16 * FS + 32 * c + 8 * g), x1 = *(const float4*)(FEAT + r16 * FS + 32 * c + 8 * g + 4);
        h2_split2(x0.x * H2_AS, x0.y * H2_AS, ah[c].x, al[c].x);
        h2_split2(x0.z * H2_AS, x0.w * H2_AS, ah[c].y, al[c].y);
        h2_split2(x1.x * H2_AS, x1.y * H2_AS, ah[c].z, al[c].z);
        h2_split2(x1.z * H2_AS, x1.w * H2_AS, ah[c].w, al[c].w);
    }
    auto tile = [&](int ct, const uint4* w) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KCH; c++) acc = h2_mma(w[2 * c], w[2 * c + 1], ah[c], al[c], acc);
        const float4 b = *(const float4*)(bias + ct * 16 + 4 * g);
        *(float4*)(LG + r16 * LS + ct * 16 + 4 * g) =
            make_float4(acc[0] * descale + b.x, acc[1] * descale + b.y, acc[2] * descale + b.z, acc[3] * descale + b.w);
    };
#pragma unroll 1
    for (int ct = wave; ct < NT; ct += 2 * NW) {
        tile(ct, w0);
        wload(ct + 2 * NW, w0);
        if (ct + NW < NT) {
            tile(ct + NW, w1);
            wload(ct + 3 * NW, w1);
        }
    }
    __syncthreads();
    for (int s = wave; s < nb; s += NW) {                      // masked softmax, one wave per sample (as k_s78_policy)
        const int b = b0 + s;
        constexpr int NK = (A + 63) / 64;
        float x[NK];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int ai = lane + 64 * k;
            x[k] = -INFINITY;
            if (ai < A) x[k] = valid[(size_t)b * A + ai] ? LG[s * LS + ai] : -1e8f;
            mx = fmaxf(mx, x[k]);
        }
        mx = nn_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NK; k++) { x[k] = (lane + 64 * k < A) ? expf(x[k] - mx) : 0.f; sum += x[k]; }
        sum = nn_wave_sum(sum);
#pragma unroll
        for (int k = 0; k < NK; k++)
            if (lane + 64 * k < A) pi[(size_t)b * A + lane + 64 * k] = x[k] / sum;
    }
}

// ... and in two launches (round 6, the default of azg_nn_s78_forward_h2): k_s78_policy_h2 streams the FC's 1.15 MB of weight fragments
// once per 16 samples -- 294 MB per 4096 leaves, 8.9 TB/s out of the L2s: that stream is what bounds it (34 us).  Here a workgroup owns
// 64 samples x a QUARTER of the column tiles (28 of 112): every fragment it loads serves four sample groups (74 MB per 4096 leaves), the
// features of its 64 samples sit in LDS as f16 hi / lo planes (row stride 336 B: conflict-free 16-byte operand reads), raw logits go to a
// workspace [B][1792] (the pi rows still hold the FEATURES the other three quarters of the same samples read), and k_s78_policy_softmax
// normalises them into pi, one wave per sample.
template <int A, int FP>
__global__ __launch_bounds__(768) void k_s78_policy_gemm_h2(const uint4* __restrict__ Wfrag, const float* __restrict__ bias, int B,
                                                            const float* __restrict__ pi_feat, float* __restrict__ logits) {
    constexpr int KCH = (FP + 31) / 32, KP = KCH * 32, NT = (A + 15) / 16, NQ = 4, TQ = NT / NQ, LROW = NT * 16, NW = 12, NSG = 4, RS = KP * 2 + 16;
    static_assert(NT % NQ == 0, "column tiles divide into quarters");
    extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
    uint8_t* FH = fsm;                      // [64][RS] f16 hi plane of 64 * x
    uint8_t* FL = FH + 64 * RS;             // lo plane
    h2_fp16_saturate_mode();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int b0 = (int)blockIdx.x * 64, nb = min(64, B - b0), cq = (int)blockIdx.y;
    const float descale = *(const float*)(Wfrag + (size_t)NT * KCH * 2 * 64);
    auto wload = [&](int ctl, uint4* w) {                         // column tile cq * TQ + ctl of this quarter
        const int ct = cq * TQ + (ctl < TQ ? ctl : 0);
#pragma unroll
        for (int c = 0; c < 2 * KCH; c++) w[c] = Wfrag[((size_t)ct * (2 * KCH) + c) * 64 + lane];
    };
    uint4 w0[2 * KCH], w1[2 * KCH];
    wload(wave, w0);                        // (under the feature staging)
    wload(wave + NW, w1);
    for (int i = tid; i < 64 * (KP / 2); i += 768) {
        const int s = i / (KP / 2), k = 2 * (i - s * (KP / 2));
        const float x0 = (s < nb && k < FP) ? pi_feat[(size_t)(b0 + s) * A + k] : 0.f;
        const float x1 = (s < nb && k + 1 < FP) ? pi_feat[(size_t)(b0 + s) * A + k + 1] : 0.f;
        uint32_t h, l;
        h2_split2(x0 * H2_AS, x1 * H2_AS, h, l);
        *(uint32_t*)(FH + s * RS + 2 * k) = h;
        *(uint32_t*)(FL + s * RS + 2 * k) = l;
    }
    __syncthreads();
    auto tile = [&](int ctl, const uint4* w) {
        const int col = (cq * TQ + ctl) * 16 + 4 * g;
        const float4 b = *(const float4*)(bias + col);
#pragma unroll 2                             /* (fully unrolled, the four groups' 40 operand reads were hoisted together: 118 spilled registers) */
        for (int sg = 0; sg < NSG; sg++) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KCH; c++) {
                const int off = (16 * sg + r16) * RS + 64 * c + 16 * g;
                acc = h2_mma(w[2 * c], w[2 * c + 1], *(const uint4*)(FH + off), *(const uint4*)(FL + off), acc);
            }
            const int s = 16 * sg + r16;
            if (s < nb)
                *(float4*)(logits + (size_t)(b0 + s) * LROW + col) =
                    make_float4(acc[0] * descale + b.x, acc[1] * descale + b.y, acc[2] * descale + b.z, acc[3] * descale + b.w);
        }
    };
#pragma unroll 1
    for (int ctl = wave; ctl < TQ; ctl += 2 * NW) {
        tile(ctl, w0);
        wload(ctl + 2 * NW, w0);
        if (ctl + NW < TQ) {
            tile(ctl + NW, w1);
            wload(ctl + 3 * NW, w1);
        }
    }
}
template <int A>
__global__ __launch_bounds__(256) void k_s78_policy_softmax(const float* __restrict__ logits, const uint8_t* __restrict__ valid, int B,
                                                            float* __restrict__ pi) {
    // one wave per sample, a lane owns PAIRS of neighbouring actions (A is even: the valid bytes of a pair are one aligned 16-bit load,
    // logits and probabilities 8-byte accesses)
    static_assert(A % 2 == 0, "pairs of actions");
    constexpr int LROW = (A + 15) / 16 * 16, NK = (A / 2 + 63) / 64;
    const int lane = threadIdx.x & 63, b = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (b >= B) return;
    float2 x[NK];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int ai = 2 * (lane + 64 * k);
        x[k] = make_float2(-INFINITY, -INFINITY);
        if (ai < A) {
            const uint32_t vb = *(const uint16_t*)(valid + (size_t)b * A + ai);
            const float2 lg = *(const float2*)(logits + (size_t)b * LROW + ai);
            x[k] = make_float2((vb & 0xFFu) ? lg.x : -1e8f, (vb >> 8) ? lg.y : -1e8f);
        }
        mx = fmaxf(mx, fmaxf(x[k].x, x[k].y));
    }
    mx = nn_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const bool on = 2 * (lane + 64 * k) < A;
        x[k] = on ? make_float2(expf(x[k].x - mx), expf(x[k].y - mx)) : make_float2(0.f, 0.f);
        sum += x[k].x; sum += x[k].y;
    }
    sum = nn_wave_sum(sum);
#pragma unroll
    for (int k = 0; k < NK; k++)
        if (2 * (lane + 64 * k) < A) *(float2*)(pi + (size_t)b * A + 2 * (lane + 64 * k)) = make_float2(x[k].x / sum, x[k].y / sum);
}

#pragma clang fp contract(off)

}  // namespace azg
