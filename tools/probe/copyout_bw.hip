// Bandwidth of a GPU kernel storing into pinned host memory (abea_copy_out_kernel) vs hipMemcpyAsync D2H.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT> __global__ void __launch_bounds__(256) copy_out(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(s[i], d + i); else d[i] = s[i];
    }
}
int main() {
    const size_t N = (size_t)1 << 30;
    void *h, *d; CK(hipHostMalloc(&h, N, hipHostMallocDefault)); CK(hipMalloc(&d, N)); CK(hipMemset(d, 1, N));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("hipMemcpyAsync D2H: %.1f GB/s\n", N / ms / 1e6);
    }
    for (int g : {16, 64, 256, 512, 2048, 8192}) {
        copy_out<1><<<g, 256>>>((const u32x4*)d, (u32x4*)h, N / 16);
        CK(hipEventRecord(e0)); copy_out<1><<<g, 256>>>((const u32x4*)d, (u32x4*)h, N / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); float a = N / ms / 1e6;
        CK(hipEventRecord(e0)); copy_out<0><<<g, 256>>>((const u32x4*)d, (u32x4*)h, N / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy-out kernel %5d blocks: nontemporal %.1f GB/s, plain %.1f GB/s\n", g, a, N / ms / 1e6);
    }
    return 0;
}
