// visibility.hip -- how OLD can the data be that an agent-scope (sc1) load returns while another workgroup keeps overwriting the word with
// write-through (sc1) stores?  Workgroup w < W writes the 100 MHz wall clock into word w as fast as it can; every workgroup polls the words of
// the OTHER workgroups and records the largest (now - value read) it ever sees, in ticks of 10 ns.  `busy` > 0: the other lanes of the polling
// wave stream through a big buffer (plain loads) to load the CU's memory queue and turn the L2 over; busy == 0: a quiet chip (the end of a launch).
//   hipcc --offload-arch=gfx950 -O2 -o visibility visibility.hip && ./visibility [ms] [words per line] [busy]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t wall32() { uint64_t t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (uint32_t)t; }
__global__ void k(uint32_t* words, int W, int stride, uint32_t run_ticks, int busy, const uint4* big, size_t big_n, uint32_t* maxage, unsigned long long* hist) {
    const int w = blockIdx.x, l = threadIdx.x;
    const uint32_t t0 = wall32();
    uint32_t* mine = words + (size_t)w * stride;
    uint32_t worst = 0, acc = 0;
    size_t pos = ((size_t)w * 977 + l) % big_n;
    for (;;) {
        const uint32_t now = wall32();
        if (now - t0 > run_ticks) break;
        if (l == 0) __hip_atomic_store(mine, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (l >= 1 && l <= 8) {                                   // eight pollers per workgroup, each watches one other writer
            const int j = (w + l * 7) % W;
            const uint32_t v = __hip_atomic_load(words + (size_t)j * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t n2 = wall32();
            if (v != 0) {
                const uint32_t age = n2 - v;
                if ((int32_t)age > (int32_t)worst) worst = age;
                const int b = age < 100 ? 0 : age < 300 ? 1 : age < 1000 ? 2 : age < 10000 ? 3 : age < 100000 ? 4 : age < 1000000 ? 5 : 6;   // <1us <3us <10us <100us <1ms <10ms more
                atomicAdd(hist + b, 1ull);
            }
        } else if (busy && l >= 16) {
            const uint4 x = big[pos]; acc += x.x ^ x.w; pos += 4099; if (pos >= big_n) pos -= big_n;
        }
    }
    if (acc == 0x12345678u) words[0] = acc;
    if (l >= 1 && l <= 8) atomicMax(maxage + w, worst);
}
int main(int argc, char** argv) {
    const int ms = argc > 1 ? atoi(argv[1]) : 2000, wpl = argc > 2 ? atoi(argv[2]) : 32, busy = argc > 3 ? atoi(argv[3]) : 0;
    const int W = 256, stride = wpl >= 32 ? 1 : 32 / wpl;
    uint32_t* words; uint32_t* maxage; unsigned long long* hist; uint4* big;
    const size_t big_n = (size_t)1 << 26;                      // 1 GiB of uint4
    hipMalloc(&words, sizeof(uint32_t) * W * 32); hipMemset(words, 0, sizeof(uint32_t) * W * 32);
    hipMalloc(&maxage, 4 * W); hipMemset(maxage, 0, 4 * W); hipMalloc(&hist, 64); hipMemset(hist, 0, 64);
    hipMalloc(&big, big_n * 16); hipMemset(big, 1, big_n * 16);
    k<<<W, 64>>>(words, W, stride, (uint32_t)ms * 100000u, busy, big, big_n, maxage, hist);
    hipDeviceSynchronize();
    uint32_t h[256]; hipMemcpy(h, maxage, 4 * W, hipMemcpyDeviceToHost);
    unsigned long long hh[8]; hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
    uint32_t mx = 0; for (int i = 0; i < W; i++) mx = h[i] > mx ? h[i] : mx;
    printf("%d ms, %d words per line, busy %d: oldest data ever returned %.2f us; ages <1us %llu <3us %llu <10us %llu <100us %llu <1ms %llu <10ms %llu more %llu\n", ms, 32 / stride, busy,
           mx / 100.0, hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6]);
    return 0;
}
