// What serialises chunk i+1 behind chunk i's kernel in the host pipeline?  
// stream A: long kernel (512 single-wave blocks, 30 ms).  stream B, issued right after: [H2D copy] -> short kernel.
// Each kernel records wall_clock64 at start/end; variants toggle the H2D copy, timing events and its size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(64) spin(long long cycles, long long* stamp) {
    __shared__ int pad[1024];
    pad[threadIdx.x] = 0;
    long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
    if (pad[threadIdx.x] == 77) stamp[2] = 1;
}
int main() {
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0));   // kHz
    const size_t N = (size_t)256 << 20;
    void *h, *d; long long* stamps; long long hs[8];
    CK(hipHostMalloc(&h, N, hipHostMallocDefault)); memset(h, 1, N);
    CK(hipMalloc(&d, N)); CK(hipMalloc(&stamps, 64)); void *h2, *d2; CK(hipHostMalloc(&h2, 4 << 20, hipHostMallocDefault)); CK(hipMalloc(&d2, 4 << 20));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    spin<<<1, 64, 0, a>>>(100, stamps); CK(hipDeviceSynchronize());
    for (int variant = 0; variant < 9; ++variant) {
        const bool copy = variant == 1 || variant >= 3, events = variant >= 2;
        const size_t sz = variant == 4 ? (size_t)1 << 20 : N;
        CK(hipMemset(stamps, 0, 64));
        if (variant == 5) CK(hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, a));    // a copy ahead of the long kernel too
        if (events) CK(hipEventRecord(e0, a));
        spin<<<512, 64, 0, a>>>((long long)clk * 30, stamps);
        if (events) CK(hipEventRecord(e1, a));
        if (variant == 6) CK(hipMemcpyAsync(h2, d2, 1 << 20, hipMemcpyDeviceToHost, a));    // D2H queued behind the long kernel, same stream
        if (variant == 7) CK(hipMemcpyAsync(d2, h2, 1 << 20, hipMemcpyHostToDevice, a));    // H2D queued behind the long kernel
        if (variant == 8) CK(hipMemcpyAsync((char*)d2 + (1 << 20), d2, 1 << 20, hipMemcpyDeviceToDevice, a));
        if (copy) CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, b));
        if (events) CK(hipEventRecord(e2, b));
        spin<<<512, 64, 0, b>>>((long long)clk * 5, stamps + 4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hs, stamps, 64, hipMemcpyDeviceToHost));
        printf("variant %d (copy %d MB=%zu, events %d): A runs [0, %.1f] ms, B's kernel runs [%.1f, %.1f] ms\n", variant, copy, copy ? sz >> 20 : 0, events,
               (hs[1] - hs[0]) / (double)clk, (hs[4] - hs[0]) / (double)clk, (hs[5] - hs[0]) / (double)clk);
    }
    return 0;
}
