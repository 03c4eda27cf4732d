// Micro-benchmark behind DESIGN.md §4.5 (round 5; round-4 verdict item 7): would a lone long read — f5c's default 512-read /
// 2-Mbase batches leave most wave slots empty, and a batch lasts as long as its longest read at ~225 ns per band — run faster
// as TWO wavefronts (one cell per lane, lanes 0..49 of wave 0 = offsets 0..49, of wave 1 = offsets 50..99) that exchange the
// boundary score and the two band-end scores through LDS behind one s_barrier per band?
//
// The band step is modelled, not copied: the instruction classes and the dependency structure of one interior band of
// abea_fill.inc (per cell: sub, cvt, mul_f64, cvt, mul, fma, cvt, 5 add_f64, 3 cvt, max3, 2 sub; per band: the DPP shift of the
// f64 copy, trace packing, v_readlane + v_cmp + scalar branch of the move decision), with the loop-carried dependency of the
// real loop (this band's max -> exact f64 copy -> shifted -> next band's sums).
//   one_wave   : 2 cells per lane, the statement as shipped                      -> ns per band, lone wave
//   two_waves  : 1 cell per lane + ds_write_b64 / s_barrier / ds_read_b64 exchange -> ns per band, lone pair
//   barrier    : the exchange alone (write, barrier, read, wait)                 -> its cost per band
// hipcc --offload-arch=gfx950 -O3 two_wave_band.hip -o two_wave_band && ./two_wave_band
#include <hip/hip_runtime.h>
#include <cstdio>

// GENERATED (see the git history of this file for the three-line generator): cell 0 alone, and cells 0 and 1 interleaved as
// tools/gen_fill_asm.py interleaves them (an in-order wave issues the two independent chains alternately)
#define C0 "v_sub_f32 v72, v78, v79\n" \
    "v_cvt_f64_f32 v[70:71], v72\n" \
    "v_mul_f64 v[70:71], v[70:71], v[82:83]\n" \
    "v_cvt_f32_f64 v72, v[70:71]\n" \
    "v_mul_f32 v74, v72, v72\n" \
    "v_fma_f32 v74, -0.5, v74, v80\n" \
    "v_cvt_f64_f32 v[70:71], v74\n" \
    "v_add_f64 v[72:73], v[64:65], %[lp]\n" \
    "v_add_f64 v[74:75], v[66:67], %[lp]\n" \
    "v_add_f64 v[72:73], v[72:73], v[70:71]\n" \
    "v_add_f64 v[74:75], v[74:75], v[70:71]\n" \
    "v_add_f64 v[70:71], v[68:69], %[lp]\n" \
    "v_cvt_f32_f64 v72, v[72:73]\n" \
    "v_cvt_f32_f64 v74, v[74:75]\n" \
    "v_cvt_f32_f64 v70, v[70:71]\n" \
    "v_max3_f32 v76, v72, v74, v70\n" \
    "v_sub_f32 v77, v74, v72\n" \
    "v_sub_f32 v72, v70, v76\n"
#define C01 "v_sub_f32 v72, v78, v79\n" \
    "v_sub_f32 v96, v102, v103\n" \
    "v_cvt_f64_f32 v[70:71], v72\n" \
    "v_cvt_f64_f32 v[94:95], v96\n" \
    "v_mul_f64 v[70:71], v[70:71], v[82:83]\n" \
    "v_mul_f64 v[94:95], v[94:95], v[106:107]\n" \
    "v_cvt_f32_f64 v72, v[70:71]\n" \
    "v_cvt_f32_f64 v96, v[94:95]\n" \
    "v_mul_f32 v74, v72, v72\n" \
    "v_mul_f32 v98, v96, v96\n" \
    "v_fma_f32 v74, -0.5, v74, v80\n" \
    "v_fma_f32 v98, -0.5, v98, v104\n" \
    "v_cvt_f64_f32 v[70:71], v74\n" \
    "v_cvt_f64_f32 v[94:95], v98\n" \
    "v_add_f64 v[72:73], v[64:65], %[lp]\n" \
    "v_add_f64 v[96:97], v[88:89], %[lp]\n" \
    "v_add_f64 v[74:75], v[66:67], %[lp]\n" \
    "v_add_f64 v[98:99], v[90:91], %[lp]\n" \
    "v_add_f64 v[72:73], v[72:73], v[70:71]\n" \
    "v_add_f64 v[96:97], v[96:97], v[94:95]\n" \
    "v_add_f64 v[74:75], v[74:75], v[70:71]\n" \
    "v_add_f64 v[98:99], v[98:99], v[94:95]\n" \
    "v_add_f64 v[70:71], v[68:69], %[lp]\n" \
    "v_add_f64 v[94:95], v[92:93], %[lp]\n" \
    "v_cvt_f32_f64 v72, v[72:73]\n" \
    "v_cvt_f32_f64 v96, v[96:97]\n" \
    "v_cvt_f32_f64 v74, v[74:75]\n" \
    "v_cvt_f32_f64 v98, v[98:99]\n" \
    "v_cvt_f32_f64 v70, v[70:71]\n" \
    "v_cvt_f32_f64 v94, v[94:95]\n" \
    "v_max3_f32 v76, v72, v74, v70\n" \
    "v_max3_f32 v100, v96, v98, v94\n" \
    "v_sub_f32 v77, v74, v72\n" \
    "v_sub_f32 v101, v98, v96\n" \
    "v_sub_f32 v72, v70, v76\n" \
    "v_sub_f32 v96, v94, v100\n"
// fixed registers: v[64:65] D0, v[66:67] U0, v[68:69] L0, v[70:71] lpd0, v[72:73] td0, v[74:75] tu0, v76 mf0, v77 fr0, v78 x0,
//                  v79 g0, v80 ck0, v[82:83] istd0; cell 1: +24
// this band's maxima -> exact f64 copies -> shifted by one lane: the next band's U / L (D = the previous U): the loop-carried chain
#define CARRY0                                                                                                   \
    "v_cvt_f64_f32 v[64:65], v76\n"                                                                             \
    "s_nop 1\n"                                                                                                 \
    "v_mov_b32_dpp v66, v64 wave_shr:1 row_mask:0xf bank_mask:0xf\n"                                            \
    "v_mov_b32_dpp v67, v65 wave_shr:1 row_mask:0xf bank_mask:0xf\n"                                            \
    "v_mov_b64 v[68:69], v[64:65]\n"
#define CARRY1                                                                                                   \
    "v_cvt_f64_f32 v[88:89], v100\n"                                                                            \
    "s_nop 1\n"                                                                                                 \
    "v_mov_b32_dpp v90, v88 wave_shr:1 row_mask:0xf bank_mask:0xf\n"                                            \
    "v_mov_b32_dpp v91, v89 wave_shr:1 row_mask:0xf bank_mask:0xf\n"                                            \
    "v_mov_b64 v[92:93], v[88:89]\n"
// trace packing + the move decision of the NEXT band (readlane of cell 0, compare with the cell-99 register, scalar test)
#define BOOK                                                                                                    \
    "v_alignbit_b32 v110, v110, v77, 31\n"                                                                      \
    "v_alignbit_b32 v110, v110, v72, 31\n"                                                                      \
    "v_readlane_b32 s20, v76, 0\n"                                                                              \
    "v_cmp_lt_f32 vcc, s20, v100\n"                                                                             \
    "v_mov_b32_dpp v78, v78 wave_ror:1 row_mask:0xf bank_mask:0xf\n"                                            \
    "s_nop 0\n"                                                                                                 \
    "s_bitcmp1_b64 vcc, 49\n"                                                                                   \
    "s_lshl1_add_u32 s21, s21, 1\n"
#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v82","v83", \
             "v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v106","v107","v110","s20","s21","vcc","scc","memory"
#define INIT "v_mov_b32 v78, 1.0\n v_mov_b32 v79, 0.5\n v_mov_b32 v80, 0.25\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0x3ff00000\n"         \
             "v_mov_b32 v102, 2.0\n v_mov_b32 v103, 0.5\n v_mov_b32 v104, 0.25\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0x3ff00000\n"    \
             "v_mov_b64 v[64:65], 0\n v_mov_b64 v[66:67], 0\n v_mov_b64 v[68:69], 0\n v_mov_b64 v[88:89], 0\n v_mov_b64 v[90:91], 0\n" \
             "v_mov_b64 v[92:93], 0\n v_mov_b32 v110, 0\n s_mov_b32 s21, 1\n"

__global__ void __launch_bounds__(64) one_wave(float* out, int bands) {
    const double lp = -0.7;
    asm volatile(INIT ::: CLOB);
    for (int b = 0; b < bands; ++b) asm volatile(C01 CARRY0 CARRY1 BOOK :: [lp] "v"(lp) : CLOB);
    float r; asm volatile("v_mov_b32 %0, v76" : "=v"(r)); out[blockIdx.x * 64 + threadIdx.x] = r;
}
// two wavefronts, one cell per lane: the boundary score (offset 49 <-> 50) and the band-end scores cross through LDS
__global__ void __launch_bounds__(128) two_waves(float* out, int bands) {
    __shared__ double xch[4];
    const double lp = -0.7;
    const int wv = threadIdx.x >> 6;
    double* mine = xch + wv; double* theirs = xch + (wv ^ 1);
    asm volatile(INIT ::: CLOB);
    for (int b = 0; b < bands; ++b) {
        asm volatile(C0 CARRY0
                     "v_alignbit_b32 v110, v110, v77, 31\n"
                     "ds_write_b64 %[mine], v[64:65]\n"                    /* lane 49's / lane 0's copy in the real thing: one lane writes */
                     "s_waitcnt lgkmcnt(0)\n"
                     "s_barrier\n"
                     "ds_read_b64 v[88:89], %[theirs]\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     "v_readlane_b32 s20, v76, 0\n"
                     "v_cmp_lt_f64 vcc, v[88:89], v[64:65]\n"
                     "v_mov_b32_dpp v78, v78 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                     "s_nop 0\n"
                     "s_bitcmp1_b64 vcc, 49\n"
                     "s_lshl1_add_u32 s21, s21, 1\n"
                     :: [lp] "v"(lp), [mine] "v"((unsigned)(size_t)mine), [theirs] "v"((unsigned)(size_t)theirs) : CLOB);
    }
    float r; asm volatile("v_mov_b32 %0, v76" : "=v"(r)); out[blockIdx.x * 128 + threadIdx.x] = r;
}
__global__ void __launch_bounds__(128) barrier_only(float* out, int bands) {
    __shared__ double xch[4];
    const int wv = threadIdx.x >> 6;
    double* mine = xch + wv; double* theirs = xch + (wv ^ 1);
    asm volatile(INIT ::: CLOB);
    for (int b = 0; b < bands; ++b)
        asm volatile("ds_write_b64 %[mine], v[64:65]\n s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b64 v[88:89], %[theirs]\n s_waitcnt lgkmcnt(0)\n"
                     "v_add_f64 v[64:65], v[88:89], v[64:65]\n"
                     :: [mine] "v"((unsigned)(size_t)mine), [theirs] "v"((unsigned)(size_t)theirs) : CLOB);
    float r; asm volatile("v_mov_b32 %0, v64" : "=v"(r)); out[blockIdx.x * 128 + threadIdx.x] = r;
}
template <class K> static double run(K kern, int threads, int blocks, int bands, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads>>>(d, 1000);
    hipEventRecord(e0); kern<<<blocks, threads>>>(d, bands); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / bands;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    const int bands = 400000;
    for (int blocks : {1, 64, 256, 1024, 4096}) {          // 1 = a lone read; 256 = one per CU; 1024 = one per SIMD; 4096 = the full GPU (4 per SIMD)
        const double a = run(one_wave, 64, blocks, bands, d), b = run(two_waves, 128, blocks, bands, d), c = run(barrier_only, 128, blocks, bands, d);
        printf("%5d reads in flight: one wave per read %.1f ns per band | two waves per read %.1f ns per band (x%.2f) | exchange alone %.1f ns\n",
               blocks, a, b, a / b, c);
    }
    return 0;
}
