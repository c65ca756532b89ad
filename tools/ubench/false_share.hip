// false_share.hip -- do 4- / 8-byte write-through (sc1) stores from workgroups on DIFFERENT XCDs to DIFFERENT words of ONE 128-byte line ever lose
// or resurrect a value?  (The pipeline's hand-over words -- ring slots, ready words, pi / v rows -- share lines between writers on different
// XCDs.)  Workgroup w < W owns word w of line (w / WPL): it stores 1, 2, 3, ... there; every workgroup also reads all words and checks that each
// one only ever grows.  A word that is seen to DECREASE is a resurrected old value.
//   hipcc --offload-arch=gfx950 -O2 -o false_share false_share.hip && ./false_share [iterations] [words per line] [mode 0 store / 1 atomic exchange]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
__global__ void k(uint32_t* words, int W, int wpl_stride, int iters, int mode, unsigned long long* viol, uint32_t* xcc) {
    const int w = blockIdx.x, l = threadIdx.x;
    uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(id));
    if (l == 0) xcc[w] = id;
    uint32_t last = 0;                         // lane l tracks word l (l < W)
    uint32_t* mine = words + (size_t)w * wpl_stride;
    for (int it = 1; it <= iters; it++) {
        if (l == 0) {
            if (mode == 0) __hip_atomic_store(mine, (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_exchange(mine, (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int j = l; j < W; j += 64) {
            const uint32_t v = __hip_atomic_load(words + (size_t)j * wpl_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (one tracker per lane: word l only when W <= 64)
            if (j == l) { if (v < last) atomicAdd(viol + 0, 1ull); else last = v; }
        }
        if ((it & 1023) == 0) __builtin_amdgcn_s_sleep(1);
    }
    // final: every owner checks its own word
    __syncthreads();
    if (l == 0) { const uint32_t v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (v != (uint32_t)iters) atomicAdd(viol + 1, 1ull); }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000000, wpl = argc > 2 ? atoi(argv[2]) : 32, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int W = 64;                                          // 64 writers, one per workgroup: with wpl = 32 they share two lines; wpl = 1: a line each
    const int stride = wpl >= 32 ? 1 : 32 / wpl;               // words between two writers' words
    uint32_t* words; unsigned long long* viol; uint32_t* xcc;
    hipMalloc(&words, sizeof(uint32_t) * W * 32); hipMemset(words, 0, sizeof(uint32_t) * W * 32);
    hipMalloc(&viol, 16); hipMemset(viol, 0, 16); hipMalloc(&xcc, 4 * W);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
    k<<<W, 64>>>(words, W, stride, iters, mode, viol, xcc);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, viol, 16, hipMemcpyDeviceToHost);
    uint32_t hx[64]; hipMemcpy(hx, xcc, 4 * W, hipMemcpyDeviceToHost);
    int nx[16] = {0}; for (int i = 0; i < W; i++) nx[hx[i] & 15]++;
    printf("mode %s, %d writers, words %d apart (%d per 128-B line), %d iterations, %.1f ms: words seen to DECREASE %llu, final values wrong %llu; writers per XCD:", mode ? "atomic exchange" : "sc1 store", W, stride, 32 / stride, iters, ms, h[0], h[1]);
    for (int i = 0; i < 8; i++) printf(" %d", nx[i]);
    printf("\n");
    return 0;
}
