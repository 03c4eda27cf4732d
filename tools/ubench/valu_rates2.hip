// Issue cost of more candidate instructions for the band-fill loop (round 2), same method as valu_rates.hip:
// 4 independent instructions x16 per loop iteration, 4 waves/SIMD on every SIMD; ns per wave64 instruction per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2 && ./valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define F4(s) asm volatile(s : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
#define D4(s) asm volatile(s : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e0), "v"(e1) : "vcc");
template <int OP> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.5f, b1 = 0.75f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e0 = 1.5, e1 = 0.75;
    int sidx = (iters & 31);
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(F4("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5")) }
        if (OP == 1) { REP16(F4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %5, %4\n v_fma_f32 %3, %3, %5, %4")) }
        if (OP == 2) { REP16(F4("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %5\n v_sub_f32 %3, %3, %5")) }
        if (OP == 3) { REP16(D4("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %5\n v_max_f64 %3, %3, %5")) }
        if (OP == 4) { REP16(D4("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %5, %4\n v_fma_f64 %3, %3, %5, %4")) }
        if (OP == 5) { REP16(F4("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %5\n v_cmp_lt_f32 vcc, %3, %5")) }
        if (OP == 6) { REP16(F4("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %4, 31\n v_alignbit_b32 %2, %2, %5, 31\n v_alignbit_b32 %3, %3, %5, 31")) }
        if (OP == 7) { REP16(F4("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %5\n v_lshl_or_b32 %3, %3, 1, %5")) }
        if (OP == 8) { REP16(F4("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3")) }
        if (OP == 9) { REP16(F4("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %5\n v_or_b32 %3, %3, %5")) }
        if (OP == 10) { REP16(F4("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %5, %4\n v_and_or_b32 %3, %3, %5, %4")) }
        if (OP == 11) { REP16(F4("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %5, %2, %4\n v_bfi_b32 %3, %5, %3, %4")) }
        if (OP == 12) { REP16(F4("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %5, %4\n v_perm_b32 %3, %3, %5, %4")) }
        if (OP == 13) { REP16(F4("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc")) }
        if (OP == 14) { REP16(F4("v_mov_b32_dpp %0, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %5 wave_rol:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 wave_rol:1 row_mask:0xf bank_mask:0xf")) }
        if (OP == 15) { REP16(F4("v_add_f32_dpp %0, %4, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %5, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %5, %3 wave_shr:1 row_mask:0xf bank_mask:0xf")) }
        if (OP == 16) { REP16(F4("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %5 row_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 row_shl:1 row_mask:0xf bank_mask:0xf")) }
        if (OP == 17) { REP16(asm volatile("v_writelane_b32 %0, 7, %4\n v_writelane_b32 %1, 7, %4\n v_writelane_b32 %2, 7, %4\n v_writelane_b32 %3, 7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sidx));) }
        if (OP == 18) { int s0; REP16(asm volatile("v_readlane_b32 %4, %0, %5\n v_readlane_b32 %4, %1, %5\n v_readlane_b32 %4, %2, %5\n v_readlane_b32 %4, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(s0) : "s"(sidx));) }
        if (OP == 19) { REP16(F4("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %5\n v_max_f32 %3, %3, %5")) }
        if (OP == 20) { REP16(F4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc")) }
        if (OP == 21) { REP16(D4("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %5\n v_add_f64 %3, %3, %5")) }
        if (OP == 22) { REP16(F4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5")) }
        if (OP == 23) { REP16(F4("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %5, %4\n v_med3_f32 %3, %3, %5, %4")) }
        if (OP == 24) { REP16(F4("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5")) }
        if (OP == 25) { REP16(D4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %5, %4\n v_pk_fma_f32 %3, %3, %5, %4" ) ) }
        // mixes: 2 f64 adds + 2 f32 fma (does full-rate f32 co-issue / hide behind f64?)
        if (OP == 26) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_fma_f32 %2, %2, %6, %7\n v_add_f64 %1, %1, %5\n v_fma_f32 %3, %3, %7, %6" : "+v"(d0), "+v"(d1), "+v"(a0), "+v"(a1) : "v"(e0), "v"(e1), "v"(b0), "v"(b1));) }
        if (OP == 27) { REP16(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %3, %1\n v_cvt_f64_f32 %1, %3\n v_cvt_f32_f64 %2, %0" : "+v"(d0), "+v"(d1), "+v"(a0), "+v"(a1));) }
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
    double n_inst = (double)iters * 64;
    double ns = ms * 1e6 / (n_inst * waves_per_simd);
    printf("%-40s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ns, ns * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {4}) {
        run<22>("v_add_f32", w); run<0>("v_mul_f32", w); run<1>("v_fma_f32", w); run<2>("v_sub_f32", w); run<19>("v_max_f32", w); run<23>("v_med3_f32", w);
        run<21>("v_add_f64", w); run<3>("v_max_f64", w); run<4>("v_fma_f64", w); run<27>("cvt f64<->f32 mix", w);
        run<5>("v_cmp_lt_f32 -> vcc (e32)", w); run<20>("v_cndmask_b32 vcc", w); run<13>("v_addc_co_u32 (shift in carry)", w);
        run<6>("v_alignbit_b32", w); run<7>("v_lshl_or_b32", w); run<8>("v_lshlrev_b32", w); run<9>("v_or_b32", w); run<24>("v_xor_b32", w);
        run<10>("v_and_or_b32", w); run<11>("v_bfi_b32", w); run<12>("v_perm_b32", w);
        run<14>("v_mov_b32_dpp wave_ror/rol", w); run<16>("v_mov_b32_dpp row_shr/shl", w); run<15>("v_add_f32_dpp wave_shl/shr", w);
        run<17>("v_writelane_b32 (sgpr lane)", w); run<18>("v_readlane_b32 (sgpr lane)", w);
        run<25>("v_pk_fma_f32", w); run<26>("2x v_add_f64 + 2x v_fma_f32 mixed", w);
    }
    return 0;
}
