// PCIe and stream-concurrency probe.  hipcc --offload-arch=gfx950 -O3 pcie_streams.hip -o pcie_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(long long cycles, int* out) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
int main() {
    const size_t N = (size_t)2 << 30;
    void *h0, *h1, *d0, *d1;
    double t = now(); CK(hipHostMalloc(&h0, N, hipHostMallocDefault)); printf("hipHostMalloc 2 GiB: %.0f ms\n", (now() - t) * 1e3);
    CK(hipHostMalloc(&h1, N, hipHostMallocDefault)); CK(hipMalloc(&d0, N)); CK(hipMalloc(&d1, N));
    memset(h0, 1, N); memset(h1, 2, N);
    hipStream_t s0, s1, s2; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        t = now(); CK(hipMemcpyAsync(d0, h0, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double a = now() - t;
        t = now(); CK(hipMemcpyAsync(h1, d1, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double b = now() - t;
        t = now(); CK(hipMemcpyAsync(d0, h0, N, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h1, d1, N, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double c = now() - t;
        printf("H2D %.1f GB/s  D2H %.1f GB/s  both at once: %.1f GB/s each direction\n", N / a / 1e9, N / b / 1e9, N / c / 1e9);
    }
    // pageable D2H
    void* p = malloc(N); memset(p, 3, N);
    t = now(); CK(hipMemcpy(p, d1, N, hipMemcpyDeviceToHost)); printf("D2H into pageable memory: %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d1, p, N, hipMemcpyHostToDevice)); printf("H2D from pageable memory: %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); CK(hipHostRegister(p, N, hipHostRegisterDefault)); printf("hipHostRegister 2 GiB: %.0f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipMemcpy(p, d1, N, hipMemcpyDeviceToHost)); printf("D2H into registered memory: %.1f GB/s\n", N / (now() - t) / 1e9);
    CK(hipHostUnregister(p));
    // stream concurrency: two kernels, each 512 single-wave blocks (fits 8x over), 20 ms each
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0)); if (!clk) clk = 100000;
    const long long cyc = (long long)clk * 20;   // kHz * 20 ms
    spin<<<1, 64, 0, s0>>>(1000, nullptr); CK(hipDeviceSynchronize());
    for (int blocks : {512, 4096, 16384}) {
        t = now(); spin<<<blocks, 64, 0, s0>>>(cyc, nullptr); CK(hipStreamSynchronize(s0)); double one = now() - t;
        t = now(); spin<<<blocks, 64, 0, s0>>>(cyc, nullptr); spin<<<blocks, 64, 0, s1>>>(cyc, nullptr);
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double two = now() - t;
        t = now(); spin<<<blocks, 64, 0, s0>>>(cyc, nullptr); spin<<<blocks, 64, 0, s1>>>(cyc, nullptr); spin<<<blocks, 64, 0, s2>>>(cyc, nullptr);
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double three = now() - t;
        printf("spin kernel %5d blocks x 64 thr, 20 ms each: one %.1f ms | two streams %.1f ms | three streams %.1f ms\n", blocks, one * 1e3, two * 1e3, three * 1e3);
    }
    // copy/kernel overlap
    t = now(); spin<<<4096, 64, 0, s0>>>(cyc * 5, nullptr); CK(hipMemcpyAsync(d0, h0, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h1, d1, N, hipMemcpyDeviceToHost, s2));
    CK(hipDeviceSynchronize()); printf("100 ms kernel + 2 GiB H2D + 2 GiB D2H on three streams: %.1f ms\n", (now() - t) * 1e3);
    return 0;
}
