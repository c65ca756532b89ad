// Is the k_select "mode"  a property of the ALLOCATION?  Repeated hipMalloc / hipFree of a forest-sized buffer in one
// process; after each allocation 4096 waves chase HOPS dependent 64-byte reads (4 lanes x 16 B: latency, not bandwidth) at random
// places of their own slab.  Prints ns per hop per incarnation.   ./chase2 [GB=160] [reps=10] [mode: 0 hipMalloc, 1 VMM 1-GiB handles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_chase(const uint4* __restrict__ mem, size_t slab_u4, int hops, uint32_t* out) {
    const int l = threadIdx.x;
    const uint4* base = mem + (size_t)blockIdx.x * slab_u4;
    uint32_t x = blockIdx.x * 2654435761u + 12345u, acc = 0;
    for (int h = 0; h < hops; h++) {
        x = x * 1664525u + 1013904223u;
        const size_t at = (size_t)(((uint64_t)(x >> 4) * (slab_u4 / 4)) >> 28) * 4;      // a 64-byte line of the slab
        uint32_t s = 0;
        if (l < 4) { const uint4 v = base[at + l]; s = v.x + v.y + v.z + v.w; }
        s = __builtin_amdgcn_readfirstlane(s);
        acc += s; x ^= s;
    }
    if (l == 0) out[blockIdx.x] = acc;
}
int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 160.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 10, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int W = 4096, hops = 256;
    const size_t GiB = 1ull << 30;
    const size_t bytes = (size_t)(gb) * GiB;
    const size_t slab_u4 = bytes / W / 16;
    uint32_t* out; CK(hipMalloc(&out, W * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < reps; rep++) {
        uint4* mem = nullptr;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        if (mode == 0) { CK(hipMalloc(&mem, bytes)); }
        else {
            hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            CK(hipMemAddressReserve((void**)&mem, bytes, 1ull << 30, nullptr, 0));
            for (size_t off = 0; off < bytes; off += GiB) {
                hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, GiB, &prop, 0));
                CK(hipMemMap((uint8_t*)mem + off, GiB, 0, h, 0)); hs.push_back(h);
            }
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(mem, bytes, &acc, 1));
        }
        CK(hipMemset(mem, 0, bytes));
        float best = 1e9f, worst = 0.f;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(e0));
            k_chase<<<W, 64>>>(mem, slab_u4, hops, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
        }
        printf("mode %d incarnation %d ptr %p: %.0f .. %.0f ns per hop\n", mode, rep, (void*)mem, best * 1e6 / hops, worst * 1e6 / hops);
        fflush(stdout);
        if (mode == 0) { CK(hipFree(mem)); }
        else {
            CK(hipMemUnmap(mem, bytes));
            for (auto h : hs) CK(hipMemRelease(h));
            CK(hipMemAddressFree(mem, bytes));
        }
    }
    return 0;
}
