// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the fill loop uses.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 2) { REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (OP == 3) { REP16(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if (OP == 4) { REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 5) { REP16(asm volatile("v_mov_b32_dpp %0, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a3), "v"(a2), "v"(a1), "v"(a0));) }
        if (OP == 6) { REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 7) { REP16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 8) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 9) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 10) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0" : "+v"(d0));) }   // dependent chain
        if (OP == 11) { REP16(asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(d0), "+v"(a0));) }  // dependent chain
        if (OP == 13) { REP16(asm volatile("v_cndmask_b32 %0, %1, %2, %4\n v_cndmask_b32 %1, %2, %3, %4\n v_cndmask_b32 %2, %3, %0, %4\n v_cndmask_b32 %3, %0, %1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(0x00ff00ff00ff00ffull));) }
        if (OP == 14) { unsigned long long m0, m1; REP16(asm volatile("v_cmp_ge_f32 %4, %0, %1\n v_cmp_eq_f32 %5, %1, %2\n v_cmp_ge_f32 %4, %2, %3\n v_cmp_eq_f32 %5, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(m0), "=&s"(m1));) }
        if (OP == 15) { int s0; REP16(asm volatile("v_readlane_b32 %4, %0, 0\n v_readlane_b32 %4, %1, 49\n v_readlane_b32 %4, %2, 5\n v_readlane_b32 %4, %3, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(s0));) }
        if (OP == 16) { REP16(asm volatile("v_lshl_or_b32 %0, %1, 2, %0\n v_alignbit_b32 %1, %0, %1, 4\n v_lshl_or_b32 %2, %3, 2, %2\n v_alignbit_b32 %3, %2, %3, 4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 17) { int s0 = i, s1 = i + 1, s2 = i + 2, s3 = i + 3; REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));) a0 += (float)(s0 + s1 + s2 + s3); }
        if (OP == 18) { int s0 = i, s1 = i + 1; REP16(asm volatile("s_add_u32 %2, %2, 1\n v_add_f32 %0, %0, %0\n s_add_u32 %3, %3, 1\n v_add_f32 %1, %1, %1" : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1));) a2 += (float)(s0 + s1); }
        if (OP == 19) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 20) { REP16(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 21) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 12) { REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(a0));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3);
}
template <int OP> void run(const char* name, int waves_per_simd) {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4 * 2);
    int iters = 2000;
    dim3 grid(256 * 4 * waves_per_simd), block(64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, block>>>(d, 10);
    hipEventRecord(e0); k<OP><<<grid, block>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n_inst = (double)iters * 64;          // instructions per wave
    double ns_per_inst_per_simd = ms * 1e6 / (n_inst * waves_per_simd);
    printf("%-34s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.1GHz, %.2f @2.4GHz)\n", name, waves_per_simd,
           ns_per_inst_per_simd, ns_per_inst_per_simd * 2.1, ns_per_inst_per_simd * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_add_f32", w); run<1>("v_add_f64", w); run<4>("v_mul_f64", w); run<2>("v_cvt_f64_f32", w); run<3>("v_cvt_f32_f64", w);
        run<5>("v_mov_b32_dpp wave_shl/shr", w); run<6>("v_max3_f32", w); run<7>("v_mov_b64", w); run<8>("v_cndmask_b32 vcc", w); run<9>("v_pk_add_f32", w);
        run<13>("v_cndmask_b32 e64 sgpr mask", w); run<14>("v_cmp_f32 -> sgpr pair", w); run<15>("v_readlane_b32", w);
        run<16>("v_lshl_or / v_alignbit", w); run<17>("s_add_u32 (SALU only)", w); run<18>("s_add_u32 + v_add_f32 mixed", w);
        run<19>("v_mov_b32", w); run<20>("v_max_f32", w); run<21>("v_pk_mul_f32", w);
        run<10>("v_add_f64 dependent chain", w); run<11>("cvt f64<->f32 dependent chain", w); run<12>("v_add_f32 dependent chain", w);
    }
    return 0;
}
