// Can the band-fill loop hand its trace bits to memory through the SCALAR store path instead of packing them with VALU?
// Per "band": WORK VALU instructions, then either
//   mode 0  nothing (floor)
//   mode 1  4 x v_cmp_ge_f32 -> SGPR pairs, 2 x s_store_dwordx4 (32 B per band), s_waitcnt lgkmcnt(0) before the next band's stores
//   mode 2  today's scheme: 4 x v_sub_f32 + 4 x v_alignbit_b32 per band, one global_store_dwordx4 per 32 bands
// 4 waves per SIMD on every SIMD, like abea_align_kernel.  Prints ns per band per wave and checks the stored masks.
// hipcc --offload-arch=gfx950 -O3 sstore.hip -o sstore && ./sstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP12(x) x x x x x x x x x x x x
template <int MODE> __global__ __launch_bounds__(64) void k(unsigned long long* out, uint4* vout, int bands) {
    const int lane = threadIdx.x;
    float a0 = lane * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.0001f, b1 = 0.9999f;
    double d0 = a0, d1 = a1;
    unsigned long long* dst = out + (size_t)blockIdx.x * bands * 4;       // 32 B per band per wave
    uint4* vdst = vout + (size_t)blockIdx.x * ((bands + 31) / 32) * 64 + lane;
    unsigned acc = 0, q0 = 0, q1 = 0, q2 = 0;
    for (int b = 0; b < bands; ++b) {
        // 48 VALU of mixed classes standing for the cell arithmetic
        REP12(asm volatile("v_add_f64 %0, %0, %4\n v_mul_f32 %2, %2, %6\n v_add_f64 %1, %1, %5\n v_mul_f32 %3, %3, %7"
                           : "+v"(d0), "+v"(d1), "+v"(a0), "+v"(a1) : "v"(d1), "v"(d0), "v"(b0), "v"(b1));)
        const float t = (float)b + 0.5f;
        if (MODE == 1) {
            unsigned long long m0, m1, m2, m3;
            asm volatile("v_cmp_ge_f32 %0, %4, %5\n v_cmp_ge_f32 %1, %6, %5\n v_cmp_ge_f32 %2, %7, %5\n v_cmp_ge_f32 %3, %8, %5\n"
                         "s_nop 1\n"
                         : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a2), "v"(t), "v"(a3), "v"(a2 + 7.f), "v"(a3 + 13.f) : "vcc");
            asm volatile("s_waitcnt lgkmcnt(0)\n"                       // the previous band's stores have read their SGPRs
                         "s_mov_b64 s[40:41], %1\n s_mov_b64 s[42:43], %2\n s_mov_b64 s[44:45], %3\n s_mov_b64 s[46:47], %4\n"
                         "s_store_dwordx4 s[40:43], %0, 0x0\n s_store_dwordx4 s[44:47], %0, 0x10\n"
                         :: "s"(dst), "s"(m0), "s"(m1), "s"(m2), "s"(m3)
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory");
            dst += 4;
        }
        if (MODE == 2) {
            float s0, s1, s2, s3;
            asm volatile("v_sub_f32 %0, %4, %8\n v_sub_f32 %1, %5, %8\n v_sub_f32 %2, %6, %8\n v_sub_f32 %3, %7, %8"
                         : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a2), "v"(a3), "v"(a2 + 7.f), "v"(a3 + 13.f), "v"(t));
            asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %0, %0, %3, 31\n v_alignbit_b32 %0, %0, %4, 31"
                         : "+v"(acc) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
            if ((b & 7) == 7) { q0 = q1; q1 = q2; q2 = acc; }
            if ((b & 31) == 31) { *vdst = make_uint4(q0, q1, q2, acc); vdst += 64; }
        }
    }
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)\n s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0 && (a0 + a1 + (float)(d0 + d1)) == 12345.678f) out[0] = acc;   // keep the work alive
}
template <int MODE> float run(unsigned long long* d, uint4* v, int waves, int bands) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<waves, 64>>>(d, v, 64);
    hipEventRecord(e0); k<MODE><<<waves, 64>>>(d, v, bands); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    const int waves = 256 * 4 * 4, bands = 20000;
    unsigned long long* d; uint4* v;
    hipMalloc(&d, (size_t)waves * bands * 32);
    hipMalloc(&v, (size_t)waves * ((bands + 31) / 32) * 1024);
    hipMemset(d, 0xAB, (size_t)waves * bands * 32);
    const float t0 = run<0>(d, v, waves, bands), t2 = run<2>(d, v, waves, bands), t1 = run<1>(d, v, waves, bands);
    printf("floor (48 VALU per band)          : %8.2f ms  %6.1f ns per band per wave\n", t0, t0 * 1e6 / bands);
    printf("sub+alignbit, vector store / 32 b : %8.2f ms  %6.1f ns per band per wave  (+%.1f)\n", t2, t2 * 1e6 / bands, (t2 - t0) * 1e6 / bands);
    printf("v_cmp -> SGPR, 2 scalar stores     : %8.2f ms  %6.1f ns per band per wave  (+%.1f)\n", t1, t1 * 1e6 / bands, (t1 - t0) * 1e6 / bands);
    // correctness of the scalar-store path: masks of the last run, a few waves
    std::vector<unsigned long long> h((size_t)bands * 4);
    int bad = 0;
    for (int w : {0, 1, 1000, waves - 1}) {
        hipMemcpy(h.data(), d + (size_t)w * bands * 4, (size_t)bands * 32, hipMemcpyDeviceToHost);
        for (int b = 0; b < bands; ++b) {
            const float t = (float)b + 0.5f;
            unsigned long long want[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                const float a2 = l + 2.f, a3 = l + 3.f;
                const float x[4] = {a2, a3, a2 + 7.f, a3 + 13.f};
                for (int q = 0; q < 4; ++q) if (x[q] >= t) want[q] |= 1ull << l;
            }
            for (int q = 0; q < 4; ++q) if (h[(size_t)b * 4 + q] != want[q]) { if (bad < 5) printf("wave %d band %d mask %d: %016llx want %016llx\n", w, b, q, h[(size_t)b * 4 + q], want[q]); ++bad; }
        }
    }
    printf("scalar-store read-back: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad != 0;
}
