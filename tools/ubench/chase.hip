// Dependent-load latency under k_select's concurrency: W waves (one per "tree"), each chasing HOPS dependent 1 KiB row reads
// (64 lanes x 16 B, like a record's hot entries) at pseudo-random rows of its own slab of SLAB bytes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/chase tools/ubench/chase.hip ; ./chase W SLAB_MB HOPS [stride_rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_chase(const uint4* __restrict__ mem, size_t slab_rows, int hops, uint32_t* out, int extra) {
    const int l = threadIdx.x;
    const uint4* base = mem + (size_t)blockIdx.x * slab_rows * 64;
    uint32_t x = blockIdx.x * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int h = 0; h < hops; h++) {
        x = x * 1664525u + 1013904223u;
        const size_t row = (size_t)(((uint64_t)(x >> 4) * slab_rows) >> 28);
        const uint4 v = base[row * 64 + l];
        uint32_t s = v.x + v.y + v.z + v.w;
        // a second, independent access issued in the same trip (header / state in another array): `extra` rows further on
        if (extra) { const uint4 w = base[((row + extra) % slab_rows) * 64 + l]; s += w.x; }
        s = __builtin_amdgcn_readfirstlane(s);
        acc += s;
        x ^= s;                                   // next address depends on the data
    }
    if (l == 0) out[blockIdx.x] = acc;
}
int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 4096;
    const double slab_mb = argc > 2 ? atof(argv[2]) : 40.0;
    const int hops = argc > 3 ? atoi(argv[3]) : 64;
    const int extra = argc > 4 ? atoi(argv[4]) : 0;
    const size_t slab_rows = (size_t)(slab_mb * 1024 * 1024 / 1024);
    const size_t bytes = (size_t)W * slab_rows * 1024;
    uint4* mem; uint32_t* out;
    CK(hipMalloc(&mem, bytes)); CK(hipMalloc(&out, W * 4));
    CK(hipMemset(mem, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        k_chase<<<W, 64>>>(mem, slab_rows, hops, out, extra);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("W=%d slab=%.2f MB total=%.1f GB hops=%d extra=%d: %.1f us/launch  %.0f ns/hop  %.0f GB/s\n", W, slab_mb, bytes / 1e9, hops, extra,
               ms * 1e3, ms * 1e6 / hops, (double)W * hops * 1024 * (extra ? 2 : 1) / (ms * 1e-3) / 1e9);
    }
    return 0;
}
