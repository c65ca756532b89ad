// Dependent-load latency vs footprint on one wavefront (and with many waves in flight): is the ~3.5k-cycle round trip the
// descent kernel sees a DRAM latency or a translation (TLB) effect?   hipcc --offload-arch=gfx950 -O3 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }
__global__ void chase(const uint32_t* buf, uint64_t n_dwords, int iters, unsigned long long* out_cycles, uint32_t* sink) {
    const uint64_t mask = n_dwords - 1;     // n_dwords is a power of two
    uint64_t idx = mix64(blockIdx.x * 977 + 13) & mask;
    uint32_t acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        uint32_t v = buf[(idx & ~63ull) + threadIdx.x];            // 64 consecutive dwords: one 256-B request per wave
        v = __builtin_amdgcn_readfirstlane(v);
        acc += v;
        idx = ((idx + v) * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL);   // cheap LCG (scalar unit), dependent on the load
        idx = (idx >> 13) & mask;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out_cycles[blockIdx.x] = (unsigned long long)(t1 - t0); sink[blockIdx.x] = acc; }
}
int main() {
    const double gbs[] = {0.0625, 1, 8, 32, 128};
    unsigned long long* cyc; uint32_t* sink;
    hipMalloc(&cyc, 8 * 65536); hipMalloc(&sink, 4 * 65536);
    for (double gb : gbs) {
        size_t bytes = (size_t)(gb * (1ull << 30));
        uint32_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc %.2f GB failed\n", gb); continue; }
        hipMemset(buf, 0, bytes);
        for (int waves : {1, 1024, 4096, 16384}) {
            const int iters = 2000;
            chase<<<waves, 64>>>(buf, bytes / 4, iters, cyc, sink);
            hipDeviceSynchronize();
            chase<<<waves, 64>>>(buf, bytes / 4, iters, cyc, sink);
            hipDeviceSynchronize();
            static unsigned long long h[16384]; hipMemcpy(h, cyc, 8 * waves, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < waves; i++) s += h[i];
            printf("footprint %7.2f GB  waves %5d  cycles/dependent load %8.1f\n", gb, waves, s / waves / iters);
        }
        hipFree(buf);
    }
    return 0;
}
