// Does "v_mul_f32 d, -a, a div:2" (output modifier, IEEE mode off) equal (-0.5f * a) * a bit for bit on gfx950,
// and what sign does inf - inf produce?  Build: hipcc --offload-arch=gfx950 -O2 -o build/omod_test tools/ubench/omod_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const float* a, uint32_t* out_ref, uint32_t* out_omod, uint32_t* out_omod_ieee, int n, uint32_t* misc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i];
    float r = __fmul_rn(__fmul_rn(-0.5f, x), x);
    float o, o2;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n\t"
                 "v_mul_f32_e64 %0, -%2, %2 div:2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 1\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_e64 %1, -%2, %2 div:2\n\t" : "=&v"(o), "=&v"(o2) : "v"(x));
    out_ref[i] = __float_as_uint(r); out_omod[i] = __float_as_uint(o); out_omod_ieee[i] = __float_as_uint(o2);
    if (i == 0) {
        float ninf = -__builtin_inff(), z;
        asm volatile("v_sub_f32 %0, %1, %1" : "=v"(z) : "v"(ninf));
        misc[0] = __float_as_uint(z);
        float one = 1.0f;
        asm volatile("v_sub_f32 %0, %1, %1" : "=v"(z) : "v"(one));
        misc[1] = __float_as_uint(z);
        uint32_t acc = 0x12345678u, t = 0x80000000u, rr;
        asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(rr) : "v"(acc), "v"(t));
        misc[2] = rr;                    // expect (acc << 1) | 1 = 0x2468acf1
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t u = (uint32_t)(s >> 16);
        float f;
        if (i % 4 == 0) { memcpy(&f, &u, 4); if (f != f) f = 1.0f; }           // any bit pattern
        else f = ((int)(u % 2000001) - 1000000) * 1e-4f * ((i % 4 == 1) ? 1.0f : (i % 4 == 2) ? 1e-3f : 30.f);   // typical z-scores
        h[i] = f;
    }
    h[0] = 0.f; h[1] = -0.f; h[2] = 1e-20f; h[3] = 1e-30f; h[4] = 3e19f; h[5] = __builtin_inff(); h[6] = 1.1754944e-38f;
    float* d; uint32_t *r, *o, *o2, *m;
    hipMalloc(&d, n * 4); hipMalloc(&r, n * 4); hipMalloc(&o, n * 4); hipMalloc(&o2, n * 4); hipMalloc(&m, 64);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, r, o, o2, n, m);
    std::vector<uint32_t> hr(n), ho(n), ho2(n); uint32_t hm[4];
    hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ho.data(), o, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ho2.data(), o2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 16, hipMemcpyDeviceToHost);
    long bad = 0, bad_norm = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) {
        if (hr[i] != ho[i]) {
            ++bad;
            float fr; memcpy(&fr, &hr[i], 4);
            if (fr != 0.f && (fr < -1e-30f)) { if (bad_norm < 5) printf("  a=%g ref=%08x omod=%08x\n", h[i], hr[i], ho[i]); ++bad_norm; }
        }
        if (hr[i] != ho2[i]) ++bad2;
    }
    printf("omod (IEEE off): %ld mismatches of %d, %ld with |ref| > 1e-30;  omod with IEEE on: %ld mismatches\n", bad, n, bad_norm, bad2);
    for (int i = 0; i < 7; ++i) printf("  a=%g ref=%08x omod=%08x ieee=%08x\n", h[i], hr[i], ho[i], ho2[i]);
    printf("(-inf)-(-inf) = %08x   1-1 = %08x   alignbit(acc,t,31) = %08x (expect 2468acf1)\n", hm[0], hm[1], hm[2]);
    return 0;
}
