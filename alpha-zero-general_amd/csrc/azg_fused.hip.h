// azg_fused.hip.h -- (included at the end of azg_nn.hip, the translation unit of the net kernels) the per-CU round kernel: `rounds` lock-step rounds of
//     select (MCTS.search descent + the previous leaf's expansion / backup, MCTS.py:105-184)  ->  NeuralNet.predict of the 16 leaves
// for SIXTEEN trees per workgroup, inside ONE launch, with no grid-wide boundary between the rounds.
//
// Why: as two kernels a round is select + net back to back and BOTH are latency chains -- a select launch lasts as
// long as the slowest of its 4096 trees (median wave 20 us, last wave 33 us), the net launch as long as one workgroup's 16 leaves --
// and neither can share a CU with the other (the net workgroup takes 504 of a SIMD's 512 VGPRs).  Here a workgroup of 16 waves owns
// 16 trees AND their 16 leaves: in the select phase every wave runs one tree's simulation (select_tree, kernels.hip.h -- the same
// code as k_select), in the net phase the same 16 waves run the V80 forward of the workgroup's 16 leaves (h2_net_body<16>,
// nn_v80_h2.hip.h -- the same code as k_v80_net_h2<16>, 128 VGPRs), separated by workgroup barriers only.  A round of a CU waits for
// the slowest of ITS 16 trees (mean of that maximum: 25.7 us where the launch-wide maximum is 33.5 us), launches and their cold starts
// disappear, and the workgroups drift apart, so the memory system sees descents and net phases mixed.
// Results are those of the two-kernel rounds bit for bit: per tree the same sequence of select_tree calls on the same pi / v.
//
// The descent's per-wave LDS (state, scratch, valid mask, path, dense policy: 3.2 KB) lies over the net's H planes, which are dead
// outside the net phase.  Leaf state / valid mask / pi / v still travel through their global buffers (written and read by the same CU).
#pragma once
#ifndef AZG_WAVE_LOCAL_SYNC
#error "the including translation unit must define AZG_WAVE_LOCAL_SYNC before azg_common.hip.h: wave_sync() = wavefront fence (16 independent tree waves per workgroup)"
#endif
#include "game_splendor.hip.h"
#include "kernels.hip.h"
#include "nn_v80_h2.hip.h"
#include <string.h>
#include <vector>

namespace azg {

template <class G>
struct RoundLds {
    using Smem = typename Forest<G>::Smem;
    static constexpr int DENSE_OFF = (int)((sizeof(Smem) + 15) / 16 * 16);
    static constexpr int STRIDE = (DENSE_OFF + G::A * 4 + 255) / 256 * 256;          // per tree wave
    static_assert(16 * STRIDE <= H2_PLH - H2_HH, "the sixteen descent blocks must fit over the net's H planes");
};

// Everything a round needs, in ONE device buffer read through the constant address space: as kernel arguments the forest descriptor
// and the net's 43 pointers (~900 bytes) were loaded at kernel entry and kept live across both phases of every round -- 536 scalar
// registers spilled into vector lanes, 349 vector registers into scratch, a round twice as long as the two kernels.  Each phase now
// copies what it needs from the buffer at ITS start (scalar loads; the pointer is laundered per round so nothing is hoisted out of the loop).
struct RoundArgs {
    ForestDev F;
    H2Weights W;
    int8_t* leaf_states; uint8_t* leaf_valid; uint8_t* needs_eval; float* pi; float* v;
    unsigned long long* prof;          // [workgroups][4] ticks of the 100 MHz wall clock: select phase, net phase, sum of the waves' own descent times, rounds
    int noise;
};
typedef const RoundArgs __attribute__((address_space(4))) * RoundArgsC;

template <class T>
__device__ __forceinline__ T load_const(const T __attribute__((address_space(4))) * p) {
    static_assert(sizeof(T) % 4 == 0, "word-sized struct");
    const uint32_t __attribute__((address_space(4))) * s = (const uint32_t __attribute__((address_space(4))) *)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = s[k];      // scalar loads; the words a phase does not use are dropped
    T out;
    __builtin_memcpy(&out, w, sizeof(T));          // (word array -> struct: the optimiser splits both into registers, as in load_uniform)
    return out;
}

template <class G>
__global__ __launch_bounds__(1024) void k_rounds_v80(const RoundArgs* args, int rounds) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using RL = RoundLds<G>;
    // phase times on the 100 MHz wall clock (s_memrealtime), kept by every wave in scalar registers and written once at the end:
    // how long the workgroup's select phase lasts (= its slowest tree), how long its net phase, and the waves' own descent times
    unsigned long long p_sel = 0, p_net = 0, p_own = 0;
#pragma unroll 1
    for (int r = 0; r < rounds; r++) {
        const unsigned long long c0 = wall_clock64();
        unsigned long long c_own;
        // ---- select phase: one tree per wave (the expansion + backup of the previous round's leaf rides in its prologue) ----
        {
            const RoundArgs* a = args;
            asm volatile("" : "+s"(a));
            int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wg = (int)blockIdx.x;
            asm volatile("" : "+s"(wave), "+s"(wg));          // (opaque per round, like the lane id: nothing of a phase is computed outside it)
            const int t = wg * 16 + wave;
            uint8_t* const mine = lds + H2_HH + wave * RL::STRIDE;
            typename RL::Smem& sm = *(typename RL::Smem*)mine;
            float* const dense = (float*)(mine + RL::DENSE_OFF);
            const RoundArgsC A = (RoundArgsC)(uintptr_t)a;
            const ForestDev F = load_const(&A->F);
            if (t < F.T) select_tree<G>(F, t, sm, dense, A->leaf_states, A->leaf_valid, A->needs_eval, A->noise, A->pi, A->v, A->noise);
            c_own = wall_clock64();
        }
        __syncthreads();               // the leaf states / masks of all 16 trees are written (workgroup scope: same CU, same L1)
        const unsigned long long c1 = wall_clock64();
        // ---- net phase: the V80 forward of this workgroup's 16 leaves on all 16 waves ----
        {
            const RoundArgs* a = args;
            asm volatile("" : "+s"(a));
            const RoundArgsC A = (RoundArgsC)(uintptr_t)a;
            int wg = (int)blockIdx.x;
            asm volatile("" : "+s"(wg));
            h2_net_body<16>(lds, &A->W, A->leaf_states, A->leaf_valid, A->F.T, G::P, A->pi, A->v, wg);
        }
        __syncthreads();               // pi / v of the 16 leaves are written; the H planes are free for the descents again
        const unsigned long long c2 = wall_clock64();
        p_sel += c1 - c0; p_net += c2 - c1; p_own += c_own - c0;
    }
    {
        const RoundArgs* a = args;
        asm volatile("" : "+s"(a));
        unsigned long long* prof = ((RoundArgsC)(uintptr_t)a)->prof;
        if (prof && (threadIdx.x & 63) == 0) {
            unsigned long long* mine = prof + (size_t)blockIdx.x * 4;
            __hip_atomic_fetch_add(mine + 2, p_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sixteen waves
            if (threadIdx.x == 0) { mine[0] += p_sel; mine[1] += p_net; mine[3] += (unsigned long long)rounds; }
        }
    }
}

}  // namespace azg

// the round kernel's argument block + profile counters: owned by the forest they belong to (azg_forest_attach), freed with it
#ifndef AZG_FUSED_DEVICE_ONLY      /* (azg_async.hip includes this file for RoundLds / load_const only) */
struct RoundSlot { RoundArgs host; RoundArgs* devbuf; unsigned long long* prof; int n_wg; };
static void round_slot_free(void* p) {
    RoundSlot* sl = (RoundSlot*)p;
    (void)hipFree(sl->devbuf); (void)hipFree(sl->prof);
    delete sl;
}

// include/azg.h: phase times of the round kernel since the last reset, averaged per round (microseconds)
extern "C" int azg_forest_rounds_profile(azg_forest* f, double* out /* [4] */, int reset) {
    if (!f || !out) return fail("azg_forest_rounds_profile: null argument");
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (RoundSlot* sl = (RoundSlot*)azg_forest_attached(f, "rounds_v80")) {
        HIPCHK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)4 * sl->n_wg);
        HIPCHK(hipMemcpy(h.data(), sl->prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        double sel = 0, net = 0, own = 0, rounds = 0;
        for (int g = 0; g < sl->n_wg; g++) { sel += (double)h[4 * g]; net += (double)h[4 * g + 1]; own += (double)h[4 * g + 2]; rounds += (double)h[4 * g + 3]; }
        if (rounds > 0) {
            out[0] = sel / rounds / 100.0;             // select phase of a workgroup (its slowest tree), us per round
            out[1] = net / rounds / 100.0;             // net phase, us per round
            out[2] = own / rounds / 16.0 / 100.0;      // a wave's own descent, us per round
            out[3] = rounds / sl->n_wg;                // rounds measured
        }
        if (reset) HIPCHK(hipMemset(sl->prof, 0, sizeof(unsigned long long) * h.size()));
    }
    return 0;
}

// include/azg.h: `rounds` self-play rounds of a Splendor-2p forest with the V80 net in one launch
extern "C" int azg_forest_rounds_v80_h2(azg_forest* f, int8_t* leaf_states, uint8_t* leaf_valid, uint8_t* needs_eval, float* pi, float* v,
                                        int noise_stride, const void* const* w, const float* descale, int rounds, void* stream) {
    if (!f || !leaf_states || !leaf_valid || !needs_eval || !pi || !v || !w || !descale) return fail("azg_forest_rounds_v80_h2: null argument");
    if (rounds <= 0) return 0;
    if (noise_stride != 0 && noise_stride != -2) return fail("azg_forest_rounds_v80_h2: noise_stride must be 0 or -2");
    int game = 0, variant = 0;
    double alpha = 0.0;
    const ForestDev* dev = azg_forest_dev_internal(f, &game, &variant, &alpha);
    if (game != AZG_SPLENDOR || variant != 2) return fail("azg_forest_rounds_v80_h2: Splendor 2 players only (the V80 geometry of nn_v80_h2.hip.h)");
    const int noise = (alpha != 0.0 && noise_stride == -2) ? 1 : 0;
    using G = SplendorDev<2>;
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute((const void*)k_rounds_v80<G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // the argument block lives in device memory, one per (forest, buffers, weights) combination; it is (re)written -- ordered on the
    // launch stream, from a host copy that stays alive -- whenever the combination changes (also under stream capture: the copy becomes a
    // graph node in front of the kernel node)
    RoundArgs want;
    memset(&want, 0, sizeof(want));
    want.F = *dev; want.W = h2_weights(w, descale);
    want.leaf_states = leaf_states; want.leaf_valid = leaf_valid; want.needs_eval = needs_eval; want.pi = pi; want.v = v; want.noise = noise;
    RoundSlot* sl = (RoundSlot*)azg_forest_attached(f, "rounds_v80");
    if (!sl) {
        sl = new RoundSlot();
        sl->n_wg = (dev->T + 15) / 16;
        sl->devbuf = nullptr; sl->prof = nullptr;
        memset(&sl->host, 0xFF, sizeof(sl->host));
        azg_forest_attach(f, "rounds_v80", sl, round_slot_free);
        HIPCHK(hipMalloc(&sl->devbuf, sizeof(RoundArgs)));
        HIPCHK(hipMalloc(&sl->prof, sizeof(unsigned long long) * 4 * sl->n_wg));
        HIPCHK(hipMemset(sl->prof, 0, sizeof(unsigned long long) * 4 * sl->n_wg));
    }
    want.prof = sl->prof;
    if (memcmp(&sl->host, &want, sizeof(want)) != 0) {
        sl->host = want;
        HIPCHK(hipMemcpyAsync(sl->devbuf, &sl->host, sizeof(RoundArgs), hipMemcpyHostToDevice, (hipStream_t)stream));
    }
    k_rounds_v80<G><<<dim3((dev->T + 15) / 16), dim3(1024), H2_LDS, (hipStream_t)stream>>>(sl->devbuf, rounds);
    HIPCHK(hipGetLastError());
    return 0;
}
#endif  // AZG_FUSED_DEVICE_ONLY
