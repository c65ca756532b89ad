// Calibration of the HBM byte counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) on access patterns with a KNOWN byte count and the
// shape of k_select's traffic: W one-wave workgroups ("trees") on private slabs far larger than the L2 + MALL, so that every access
// is a miss.
//   mode 0: HOPS dependent reads of one 1-KiB row (64 lanes x 16 B: a record's hot run) at random rows       -> W * HOPS * 1024 B fetched
//   mode 1: HOPS reads of 64 B (16 lanes x 4 B: a header / child-slot run) at random 128-B lines              -> W * HOPS * 64 B used
//   mode 2: HOPS writes of one 1-KiB row (an expansion's entry run)                                            -> W * HOPS * 1024 B written
//   mode 3: HOPS writes of 16 B by one lane at random lines (a backup's entry update)                          -> W * HOPS * 16 B written
//   mode 4: HOPS reads of 256 B (64 lanes x 4 B: a child-slot run, a state chunk)                              -> W * HOPS * 256 B used
//   mode 5: HOPS reads of 128 B (64 lanes x 2 B: the action ids)                                               -> W * HOPS * 128 B used
//   mode 6: HOPS reads of 32 B by all lanes at the same address (a record header)                              -> W * HOPS * 32 B used
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/calib tools/ubench/calib.hip ; ./calib MODE [W SLAB_MB HOPS]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_calib(uint4* __restrict__ mem, size_t slab_rows, int hops, int mode, uint32_t* out) {
    const int l = threadIdx.x;
    uint4* base = mem + (size_t)blockIdx.x * slab_rows * 64;
    uint32_t x = blockIdx.x * 2654435761u + 12345u, acc = 0;
    for (int h = 0; h < hops; h++) {
        x = x * 1664525u + 1013904223u;
        const size_t row = (size_t)(((uint64_t)(x >> 4) * slab_rows) >> 28);
        if (mode == 0) { const uint4 v = base[row * 64 + l]; acc += __builtin_amdgcn_readfirstlane(v.x + v.y + v.z + v.w); x ^= acc; }
        else if (mode == 1) { uint32_t v = 0; if (l < 16) v = ((const uint32_t*)(base + row * 64))[l]; acc += __builtin_amdgcn_readfirstlane(v); x ^= acc; }
        else if (mode == 4) { const uint32_t v = ((const uint32_t*)(base + row * 64))[l]; acc += __builtin_amdgcn_readfirstlane(v); x ^= acc; }
        else if (mode == 5) { const uint32_t v = ((const uint16_t*)(base + row * 64))[l]; acc += __builtin_amdgcn_readfirstlane(v); x ^= acc; }
        else if (mode == 6) { const uint2 v = ((const uint2*)(base + row * 64))[l & 3]; acc += __builtin_amdgcn_readfirstlane(v.x + v.y); x ^= acc; }
        else if (mode == 2) base[row * 64 + l] = make_uint4(x, h, l, 7u);
        else if (l == 0) base[row * 64] = make_uint4(x, h, 3u, 7u);
    }
    if (l == 0) out[blockIdx.x] = acc;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int W = argc > 2 ? atoi(argv[2]) : 4096;
    const double slab_mb = argc > 3 ? atof(argv[3]) : 32.0;
    const int hops = argc > 4 ? atoi(argv[4]) : 64;
    const size_t slab_rows = (size_t)(slab_mb * 1024 * 1024 / 1024);
    const size_t bytes = (size_t)W * slab_rows * 1024;
    uint4* mem; uint32_t* out;
    CK(hipMalloc(&mem, bytes)); CK(hipMalloc(&out, W * 4));
    CK(hipMemset(mem, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 4; rep++) k_calib<<<W, 64>>>(mem, slab_rows, hops, mode, out);
    CK(hipDeviceSynchronize());
    const double per = mode == 0 || mode == 2 ? 1024.0 : mode == 1 ? 64.0 : mode == 4 ? 256.0 : mode == 5 ? 128.0 : mode == 6 ? 32.0 : 16.0;
    printf("{\"mode\": %d, \"W\": %d, \"hops\": %d, \"launches\": 4, \"known_bytes_per_launch\": %.0f}\n", mode, W, hops, (double)W * hops * per);
    return 0;
}
