// Replica of the host pipeline's device-side command pattern (abea_host.cpp): chunks rotate through 4 streams, each
// chunk = H2D -> kernel -> results back.  Which way of getting the results back lets chunk i+1 start while chunk i runs?
//   mode 0: hipMemcpyAsync D2H on the chunk's stream (SDMA)        mode 1: a copy kernel storing into pinned host memory
//   mode 2: nothing comes back (upper bound)                        mode 3: D2H on a separate copy stream behind an event
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void __launch_bounds__(64) spin(long long cycles, long long* stamp) {
    long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void __launch_bounds__(256) copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0));
    const int NC = 5; const double dur[NC] = {22, 12, 15, 12, 4}; const size_t up_mb[NC] = {64, 100, 250, 250, 60}, dn_mb[NC] = {6, 8, 16, 16, 8};
    void *h_up, *h_dn, *d_up[4], *d_dn[4]; long long* stamps;
    CK(hipHostMalloc(&h_up, (size_t)1 << 30, hipHostMallocDefault)); CK(hipHostMalloc(&h_dn, (size_t)256 << 20, hipHostMallocDefault));
    memset(h_up, 1, (size_t)1 << 30);
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&d_up[i], (size_t)256 << 20)); CK(hipMalloc(&d_dn[i], (size_t)64 << 20)); }
    CK(hipMalloc(&stamps, 4096));
    hipStream_t st[4], cs; hipEvent_t kd[4];
    for (int i = 0; i < 4; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&kd[i], hipEventDisableTiming)); }
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    spin<<<1, 64, 0, st[0]>>>(100, stamps); CK(hipDeviceSynchronize());
    for (int mode = 0; mode < 4; ++mode) for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(stamps, 0, 4096));
        const double t0 = now();
        for (int c = 0; c < NC; ++c) {
            hipStream_t s = st[c % 4];
            CK(hipMemcpyAsync(d_up[c % 4], (char*)h_up + ((size_t)c << 27), up_mb[c] << 20, hipMemcpyHostToDevice, s));
            spin<<<512 << (c > 1), 64, 0, s>>>((long long)(clk * dur[c]), stamps + 8 * c);
            if (mode == 0) CK(hipMemcpyAsync((char*)h_dn + ((size_t)c << 24), d_dn[c % 4], dn_mb[c] << 20, hipMemcpyDeviceToHost, s));
            if (mode == 1) copy_out<<<64, 256, 0, s>>>((const uint4*)d_dn[c % 4], (uint4*)((char*)h_dn + ((size_t)c << 24)), (dn_mb[c] << 20) / 16);
            if (mode == 3) { CK(hipEventRecord(kd[c % 4], s)); CK(hipStreamWaitEvent(cs, kd[c % 4], 0));
                             CK(hipMemcpyAsync((char*)h_dn + ((size_t)c << 24), d_dn[c % 4], dn_mb[c] << 20, hipMemcpyDeviceToHost, cs)); }
        }
        const double t_issue = now() - t0;
        CK(hipDeviceSynchronize());
        const double t_all = now() - t0;
        long long hs[512]; CK(hipMemcpy(hs, stamps, 4096, hipMemcpyDeviceToHost));
        if (rep == 1) {
            printf("mode %d: issue %.2f ms, all done %.1f ms; kernels:", mode, t_issue, t_all);
            for (int c = 0; c < NC; ++c) printf(" [%.1f, %.1f]", (hs[8 * c] - hs[0]) / (double)clk, (hs[8 * c + 1] - hs[0]) / (double)clk);
            printf("\n");
        }
    }
    return 0;
}
