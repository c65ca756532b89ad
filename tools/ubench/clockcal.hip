// calibrate clock64() (s_memtime) and wall_clock64() against hipEvent time
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long ticks, long long* out) {
    long long t0 = clock64(), w0 = wall_clock64();
    while (clock64() - t0 < ticks) {}
    out[0] = clock64() - t0; out[1] = wall_clock64() - w0;
}
int main() {
    long long* d; hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (long long ticks : {1000000LL, 10000000LL, 100000000LL}) {
        spin<<<1, 64>>>(ticks, d); hipDeviceSynchronize();
        hipEventRecord(a); spin<<<1, 64>>>(ticks, d); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("clock64 ticks %lld in %.3f ms -> %.1f MHz ; wall_clock64 ticks %lld -> %.1f MHz\n", h[0], ms, h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
    }
}
