/* abea_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for adaptive banded event
 * alignment.  Not a port of the reference's align.cu/align.hip: the band lives in VGPRs of ONE
 * wavefront (2 cells per lane, lanes 0..49), neighbour exchange is a single DPP wave shift per
 * band, the diagonal comes for free from the previous band's shifted copy, the band-move
 * decision is two v_readlane + a scalar branch, trace is packed to 2 bit/cell and leaves the CU
 * as one coalesced 1 KiB store per 32 bands.  No block barriers in the fill loop.
 *
 * Arithmetic follows the reference CPU path (src/align.c), not its GPU kernels: score sums are
 * evaluated in fp64 and rounded once to fp32 (align.c:382-384), float expressions are kept
 * un-fused (this file MUST be compiled with -ffp-contract=off), per-read log constants come from
 * the host (glibc).  See DESIGN.md for the derivations cited in comments below.
 *
 * Kernels:  abea_pre_kernel   (align-pre: k-mer ranks -> read-scaled emission params, event means SoA)
 *           abea_align_kernel (band fill + adaptive band movement + online end-point scan, then the
 *                              align-post traceback walk, pair expansion and ordered QC sums and — when
 *                              asked for — scaling_single of the read: postalign + recalibrate_model; fused)
 *           abea_copy_out_kernel, and the event-detection kernels abea_ev_* (row N2)
 */
#include <hip/hip_runtime.h>
#include <type_traits>
#include "abea_device.h"
#include "abea_fill.inc"
#include "abea_walk.inc"

#define NINF (-__builtin_inff())

/* ---------------------------------------------------------------- cross-lane primitives */
/* lane i <- lane i-1 ; lane 0 keeps `oldv`  (DPP wave_shr:1) */
static __device__ __forceinline__ int dpp_from_lower_i(int oldv, int v) {
    return __builtin_amdgcn_update_dpp(oldv, v, 0x138, 0xf, 0xf, false);
}
/* lane i <- lane i+1 ; lane 63 keeps `oldv` (DPP wave_shl:1) */
static __device__ __forceinline__ int dpp_from_upper_i(int oldv, int v) {
    return __builtin_amdgcn_update_dpp(oldv, v, 0x130, 0xf, 0xf, false);
}
static __device__ __forceinline__ float dpp_from_lower_f(float oldv, float v) {
    return __int_as_float(dpp_from_lower_i(__float_as_int(oldv), __float_as_int(v)));
}
static __device__ __forceinline__ float dpp_from_upper_f(float oldv, float v) {
    return __int_as_float(dpp_from_upper_i(__float_as_int(oldv), __float_as_int(v)));
}
static __device__ __forceinline__ double dpp_from_lower_d(double oldv, double v) {
    int lo = dpp_from_lower_i(__double2loint(oldv), __double2loint(v));
    int hi = dpp_from_lower_i(__double2hiint(oldv), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double dpp_from_upper_d(double oldv, double v) {
    int lo = dpp_from_upper_i(__double2loint(oldv), __double2loint(v));
    int hi = dpp_from_upper_i(__double2hiint(oldv), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
static __device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
/* v_writelane_b32 with an immediate lane (gfx9 allows one SGPR source; this clang has no writelane builtin) */
template <int LANE> static __device__ __forceinline__ int writelane_i(int oldv, int val) {
    int r = oldv;
    const int s = __builtin_amdgcn_readfirstlane(val);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(r) : "s"(s), "n"(LANE));
    return r;
}
template <int LANE> static __device__ __forceinline__ float writelane_f(float oldv, float val) {
    return __int_as_float(writelane_i<LANE>(__float_as_int(oldv), __float_as_int(val)));
}
static __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
static __device__ __forceinline__ double uni_d(double v) {
    return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v)));
}
static __device__ __forceinline__ const void* uni_p(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    return (const void*)(((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)a));
}

/* ---------------------------------------------------------------- selftest */
extern "C" __global__ void abea_selftest_kernel(int* out) {
    int l = threadIdx.x;
    out[l]       = dpp_from_lower_i(-1, l);        /* expect l-1, lane 0 -> -1 */
    out[64 + l]  = dpp_from_upper_i(-1, l);        /* expect l+1, lane 63 -> -1 */
    out[128 + l] = readlane_i(l * 3, 49);          /* expect 147 */
    out[192 + l] = writelane_i<49>(l, 777);        /* expect l, lane 49 -> 777 */
    /* correctly rounded float quotient via one fp64 multiply (DESIGN.md "division") */
    float a = 3.0f + l * 0.37f, b = 1.5f + l * 0.013f;
    double ib = 1.0 / (double)b;
    float q1 = (float)((double)a * ib);
    float q2 = a / b;
    out[256 + l] = (__float_as_int(q1) == __float_as_int(q2)) ? 1 : 0;
}

/* ---------------------------------------------------------------- align-pre */
static __device__ __forceinline__ uint32_t base_code(uint32_t c) {
    /* align.c:19-32: A0 C1 G2 T3, anything else ranks as 0.  Arithmetic on the three comparisons, not a chain of ?: — the
     * compiler turned that chain into a switch and the switch into four exec-masked branches per base (round 6: 25 instructions
     * and a full s_waitcnt per base in every rank loop of the library). */
    return (uint32_t)(c == (uint32_t)'C') + 2u * (uint32_t)(c == (uint32_t)'G') + 3u * (uint32_t)(c == (uint32_t)'T');
}

/* Rank of the k-mer that starts at seq[i] (align.c:36-47, first base most significant); the read has L bases and a NUL behind
 * them.  One 12-byte load (any alignment; k <= 9) and the codes picked out of registers where the window stays inside the read's
 * L + 1 bytes, base by base for the last few k-mers of a read: never a byte beyond the read's own terminator is touched (the
 * device entry reads sequences from caller-owned memory). */
static __device__ __forceinline__ uint32_t kmer_rank_at(const char* __restrict__ seq, int i, int L, int kmer_size) {
    uint32_t rank = 0;
    if (i + 12 <= L + 1) {
        uint32_t w[3];
        __builtin_memcpy(w, seq + i, 12);
        #pragma unroll
        for (int j = 0; j < ABEA_MAX_KMER_SIZE; ++j)
            if (j < kmer_size) rank = (rank << 2) | base_code((w[j >> 2] >> (8 * (j & 3))) & 0xFFu);
    } else {
        for (int j = 0; j < kmer_size; ++j) rank = (rank << 2) | base_code((uint32_t)(unsigned char)seq[i + j]);
    }
    return rank;
}

extern "C" __global__ __launch_bounds__(256)
void abea_pre_kernel(const abea_read_desc* __restrict__ descs,
                     const char* __restrict__ reads, const abea_event_t* __restrict__ events,
                     const abea_model_t* __restrict__ model, int kmer_size,
                     abea_kpar_t* __restrict__ kpar_all, float* __restrict__ evm_all, uint32_t* __restrict__ krank_all) {
    const abea_read_desc* d = descs + blockIdx.x;
    if (d->n_groups == 0) return;                      /* skipped read */
    const int K = d->n_kmers, E = d->n_events;
    const float scale = d->scale, shift = d->shift;
    const char* seq = reads + d->read_off;
    abea_kpar_t* kp = kpar_all + d->kpar_off;
    uint32_t* const kr = krank_all ? krank_all + d->kpar_off : nullptr;     /* the ranks, for phase 4 of a fused launch */
    const int L = K + kmer_size - 1;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const uint32_t rank = kmer_rank_at(seq, i, L, kmer_size);
        const abea_model_t m = model[rank];
        abea_kpar_t p;
        p.gpm  = __fadd_rn(__fmul_rn(scale, m.level_mean), shift);   /* align.c:137-138, mul then add, no FMA */
        p.ck   = __fsub_rn(-0.918938f, m.level_log_stdv);            /* align.c:111-113 */
        p.istd = 1.0 / (double)m.level_stdv;
        kp[i] = p;
        if (kr) kr[i] = rank;
    }
    if (!events) return;                               /* host path: the means were uploaded straight into evm */
    const abea_event_t* ev = events + d->event_off;
    float* evm = evm_all + d->evm_off;
    /* align.c:131 reads .mean only.  Four loads in flight per thread before the first store: the loop is pure streaming */
    int i = threadIdx.x;
    for (; i + 3 * (int)blockDim.x < E; i += 4 * blockDim.x) {
        const float m0 = ev[i].mean, m1 = ev[i + blockDim.x].mean, m2 = ev[i + 2 * blockDim.x].mean, m3 = ev[i + 3 * blockDim.x].mean;
        evm[i] = m0; evm[i + blockDim.x] = m1; evm[i + 2 * blockDim.x] = m2; evm[i + 3 * blockDim.x] = m3;
    }
    for (; i < E; i += blockDim.x) evm[i] = ev[i].mean;
}

/* ---------------------------------------------------------------- band fill */
/* One DP cell (align.c:378-392).  D,U,L are the neighbour scores as exact doubles. */
static __device__ __forceinline__ void abea_cell(float x, float gpm, float ck, double istd,
                                                 double D, double U, double L,
                                                 double lp_step, double lp_stay, double lp_skip,
                                                 float& m, uint32_t& from) {
    float dx = __fsub_rn(x, gpm);
    float a  = (float)((double)dx * istd);                     /* == dx / stdv, correctly rounded */
    /* align.c:113: ck + (-0.5f*a)*a.  Halving is exact, so RN((-0.5a)*a) = -0.5*RN(a*a) and one fma adds that product to ck
     * with the single rounding of the reference's add (identical unless a*a underflows and ck == 0: one subnormal ulp) */
    float lp = __fmaf_rn(-0.5f, __fmul_rn(a, a), ck);
    double lpd = (double)lp;
    float sd = (float)((D + lp_step) + lpd);                   /* align.c:382 */
    float su = (float)((U + lp_stay) + lpd);                   /* align.c:383 */
    float sl = (float)(L + lp_skip);                           /* align.c:384 */
    /* align.c:386-392: ties prefer L over U over D; all -inf -> FROM_L */
    float m1 = fmaxf(sd, su);
    uint32_t f = (su >= sd) ? 1u : 0u;
    m = fmaxf(m1, sl);
    from = (sl >= m1) ? 2u : f;
}

/* ================================================================ scaling_single, per read, by the wavefront that aligned it
 * postalign (align.c:561-661) + recalibrate_model (align.c:666-773) + the flags of scaling_single (f5c.c:736-807).
 * Rounds 1-3 ran this as one / two kernels behind the alignment kernel of a chunk; on a GPU whose 4096 wave slots are held
 * by the alignment kernels of the other chunks those small kernels waited milliseconds for slots, and the lane-per-read
 * chains of the recalibration took 4-15 ms per chunk (rocprofv3 timeline, profiles/r04/e_fused_20k_kernel_trace.csv).
 * Now it is the tail of abea_align_kernel: no launch, no queueing, no 'M'-state records through HBM.
 *   base_to_event_map: every k-mer owns one contiguous run of pairs; its first pair repeats the previous event iff it was
 *     reached by a skip (FROM_L), all later pairs of the run are new events.  Written by phase 3 of the kernel while it expands
 *     the walk (two codes per pair decide), so the pair lists themselves need not exist in HBM.
 *   'M' states = first event of each k-mer that has events and whose rank differs from the previous such k-mer
 *     (hmm_state, align.c:637; counted at align.c:677-686), found 64 k-mers at a time with ballots.
 *   recalibrate_model's five normal-equation sums and its variance sum are sequential fp64 chains in k order whose terms are
 *     full-mantissa doubles: no re-association is exact, the chains stay sequential.  But the TERMS are independent: the 64
 *     lanes compute them in parallel (the fp64 division included), park them in LDS in 'M'-state order, and then lanes 0..4
 *     (one per sum) add their column in order — a step of the chain is one LDS read (pipelined eight deep) and one v_add_f64,
 *     not a trip to HBM.  Same operations, same order, same bits as align.c:688-753. */
/* written by this wavefront a moment ago: read past the CU's L1, which may hold a neighbour's stale copy of a shared line */
static __device__ __forceinline__ abea_index_pair_t load_map_l2(const abea_index_pair_t* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    abea_index_pair_t r; r.start = (int32_t)(uint32_t)v; r.stop = (int32_t)(uint32_t)(v >> 32);
    return r;
}

static __device__ void abea_fused_not_aligned(const abea_fused_scaling& fs, int out_idx, int lane) {
    if (lane == 0) {                                     /* f5c.c:786-794: could not align */
        fs.flag_io[out_idx] |= ABEA_FAILED_ALIGNMENT; fs.epb[out_idx] = 0.0; fs.nalign[out_idx] = 0;
        if (fs.var_f64) fs.var_f64[out_idx] = -1.0;
    }
}

/* 16 bytes this wavefront stored a moment ago, read past the CU's L1 (the trace lines were loaded through it by the walk) */
static __device__ __forceinline__ uint4 load_u4_l2(const uint4* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

/* 64 terms of one chain, in order, 16 bytes per LDS read; lanes outside the chain's set idle through the same instructions */
static __device__ __forceinline__ double chain64(const double* col, int cnt, double acc) {
    const double2* c2 = reinterpret_cast<const double2*>(col);
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {
        const double2 a = c2[i / 2], b = c2[i / 2 + 1], c = c2[i / 2 + 2], d = c2[i / 2 + 3];
        acc += a.x; acc += a.y; acc += b.x; acc += b.y; acc += c.x; acc += c.y; acc += d.x; acc += d.y;
    }
    for (; i < cnt; ++i) acc += col[i];
    return acc;
}

/* The map has been written by phase 3 of the alignment kernel; event_span = read_pos of the list's last pair minus that of its
 * first (align.c:602); lds = 5 columns of ABEA_P4_COL doubles of wave-private LDS; recs = the read's own trace scratch, dead since
 * the walk, at least n_kmers uint4 long.
 * Round 6 (the "diet"): rounds 4-5 swept the k-mers twice, and each sweep recomputed every k-mer's rank base by base (the compiler
 * had made four branches and a full memory wait out of every base), reloaded the map and gathered the model again.  Now the first
 * sweep leaves one 16-byte record {level_mean, level_stdv, event mean} per 'M' state, in 'M'-state order, in the trace scratch,
 * and the variance pass reads those records DENSELY, 64 'M' states per step — no ranks, no map, no ballots —; the ranks come from
 * the array align-pre leaves behind the k-mer parameters (it has every rank in a register anyway), the three normal-equation
 * terms that depend on the model alone from a table built at abea_init with the reference's expressions, and every chain takes
 * its terms from LDS 16 bytes at a time. */
#define ABEA_P4_COL 66            /* doubles per column: 5 columns start 4 banks apart */
static __device__ void abea_scaling_single_wave(const abea_read_desc* d, const abea_fused_scaling& fs, int lane, int event_span,
                                                const float* __restrict__ evm, double* lds, uint4* __restrict__ recs) {
    const int out_idx = d->out_idx;
    const int K = d->n_kmers;
    const abea_index_pair_t* map = fs.b2e + d->kmer_off;
    const abea_model_t* __restrict__ model = fs.model;
    const double events_per_base = (double)event_span / K;   /* align.c:602 */

    /* ---- sweep over the k-mers in k order, 64 at a time: the 'M' states, their records, and the terms of the five
     *      normal-equation sums (align.c:697-706), added by lanes 0..4 in 'M'-state order.
     *      Software-pipelined (round 6): a block's wave time was 2.8 us, two thirds of it memory latency in a row — map entry and
     *      sequence window, then the gathers that depend on them (model entry, event mean), then the chain.  Now the loads of
     *      block b + 2 are in flight while block b is worked on, the gathers of block b + 1 while the chain of block b runs, and
     *      the wavefront orders its own LDS traffic without workgroup barriers (one wavefront = the workgroup; its LDS operations
     *      execute in issue order), which also keeps the prefetched loads in flight: __syncthreads() waits for every outstanding
     *      global load. ---- */
    struct blk_in { abea_index_pair_t m; uint32_t rank; };                /* stage A: map entry + the k-mer's rank (left by align-pre) */
    struct blk_st { abea_index_pair_t m; int rank; bool valid, isM; int pos, cnt; abea_model_t mo; double t0, t1, t2; float raw; };
    /* unconditional loads from clamped addresses: nothing for the compiler to wait for at the point of issue; lanes past K - 1 are
     * sorted out in stage().  Relaxed atomic loads: a plain load is sunk by the optimizer down to its first use, a block and a half
     * later — the opposite of a prefetch —, atomic loads stay where they are written. */
    const uint32_t* __restrict__ krank = fs.krank + d->kpar_off;
    const double* __restrict__ mterms = fs.mterms;
    auto load_in = [&](int k0) {
        blk_in a;
        const int k = min(k0 + lane, K - 1);
        a.m = load_map_l2(map + k);
        a.rank = __hip_atomic_load(krank + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return a;
    };
    int n_M = 0, n_align = 0;
    int carry_rank = -1;
    auto stage = [&](const blk_in& a, int k0) {                           /* stage B: 'M' states, gathers issued (not waited for) */
        blk_st b; b.m = a.m; b.rank = (int)a.rank; b.mo.level_mean = b.mo.level_stdv = b.mo.level_log_stdv = 0.f; b.raw = 0.f;
        b.t0 = b.t1 = b.t2 = 0.0;
        const int k = k0 + lane;
        if (k >= K) { b.m.start = -1; b.m.stop = -1; }                    /* the clamped load fetched entry K - 1 again */
        b.valid = b.m.start != -1;
        /* the map's entries tile the events of the path in k order (every event is new for exactly one k-mer), so the number
         * of events per k-mer IS the map: one byte per k-mer for the host entry instead of eight (255 = "255 or more": the
         * host then rebuilds that read's map from the walk) */
        if (fs.kcnt && k < K)
            fs.kcnt[d->kmer_off + k] = b.valid ? (uint8_t)min(b.m.stop - b.m.start + 1, 255) : (uint8_t)0;
        const unsigned long long vm = __ballot(b.valid);
        const unsigned long long lower = vm & ((1ull << lane) - 1ull);
        const int src = lower ? 63 - __clzll(lower) : 0;
        const int below = __shfl(b.rank, src, 64);    /* every lane takes part: the source lane may have lower == 0 */
        const int prev_rank = lower ? below : carry_rank;
        b.isM = b.valid && (b.rank != prev_rank);         /* hmm_state 'M', align.c:637; counted at align.c:677-686 */
        const unsigned long long mm = __ballot(b.isM);
        b.cnt = __popcll(mm);
        b.pos = __popcll(mm & ((1ull << lane) - 1ull));   /* this 'M' state's place among the block's */
        if (b.isM) {
            b.mo = model[b.rank]; b.raw = evm[b.m.start];
            b.t0 = mterms[3 * b.rank]; b.t1 = mterms[3 * b.rank + 1]; b.t2 = mterms[3 * b.rank + 2];
        }
        n_align += b.valid ? (b.m.stop - b.m.start + 1) : 0;
        if (vm) carry_rank = __shfl(b.rank, 63 - __clzll(vm), 64);
        return b;
    };
    double acc = 0.0;                                    /* lanes 0..4: A00, A01, A11, b0, b1 */
    {
        blk_in a1 = load_in(0), a2 = load_in(64);
        blk_st cur = stage(a1, 0);
        a1 = a2;
        for (int k0 = 0; k0 < K; k0 += 64) {
            a2 = load_in(k0 + 128);
            if (cur.isM) {
                recs[n_M + cur.pos] = make_uint4(__float_as_uint(cur.mo.level_mean), __float_as_uint(cur.mo.level_stdv), __float_as_uint(cur.raw), 0u);
                const double mu = cur.mo.level_mean, e = cur.raw, inv_var = cur.t0;     /* inv_var = 1. / (level_stdv * level_stdv), from the table */
                lds[0 * ABEA_P4_COL + cur.pos] = inv_var;
                lds[1 * ABEA_P4_COL + cur.pos] = cur.t1;               /* mu * inv_var */
                lds[2 * ABEA_P4_COL + cur.pos] = cur.t2;               /* mu * mu * inv_var */
                lds[3 * ABEA_P4_COL + cur.pos] = e * inv_var;
                lds[4 * ABEA_P4_COL + cur.pos] = mu * e * inv_var;
            }
            const int cnt = cur.cnt;
            n_M += cnt;
            const blk_st nxt = stage(a1, k0 + 64);        /* its gathers fly while the chain below runs */
            __builtin_amdgcn_wave_barrier();
            if (lane < 5) acc = chain64(lds + lane * ABEA_P4_COL, cnt, acc);
            __builtin_amdgcn_wave_barrier();
            cur = nxt; a1 = a2;
        }
    }
    __syncthreads();                                      /* the record stores are complete (and in L2) before they are read back */
    const bool calibrated = n_M >= fs.min_rescale;        /* align.c:688: not enough 'M' states, no recalibration */
    double shift = 0, scale = 0;
    if (calibrated) {
        const double A00 = __shfl(acc, 0, 64), A01 = __shfl(acc, 1, 64), A11 = __shfl(acc, 2, 64);
        const double b0 = __shfl(acc, 3, 64), b1 = __shfl(acc, 4, 64);
        const double A10 = A01;
        const double div = A00 * A11 - A01 * A10;         /* align.c:721-723 */
        shift = -(A01 * b1 - A11 * b0) / div;
        scale = (A00 * b1 - A10 * b0) / div;
        /* ---- the variance sum (align.c:738-751) over the records, 64 'M' states per step; lane 0 owns the chain; the records
         *      of the next step are loaded while it runs ---- */
        acc = 0.0;
        auto load_rec = [&](int i) { return i < n_M ? load_u4_l2(recs + i) : make_uint4(0u, 0u, 0u, 0u); };
        uint4 rc = load_rec(lane);
        for (int i0 = 0; i0 < n_M; i0 += 64) {
            const uint4 rn = load_rec(i0 + 64 + lane);
            if (i0 + lane < n_M) {
                const double level_mean = __uint_as_float(rc.x), level_stdv = __uint_as_float(rc.y), raw_event = __uint_as_float(rc.z);
                const double yi = (raw_event - shift - scale * level_mean);
                lds[lane] = yi * yi / (level_stdv * level_stdv);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) acc = chain64(lds, min(64, n_M - i0), acc);
            __builtin_amdgcn_wave_barrier();
            rc = rn;
        }
    }
    for (int off = 32; off > 0; off >>= 1) n_align += __shfl_xor(n_align, off, 64);
    if (lane == 0) {
        int flag = 0;
        float fvar = fs.sc_io[out_idx].var;
        double var = -1.0;
        if (calibrated) {
            var = acc / n_M;                              /* align.c:752-753 */
            var = sqrt(var);
            abea_scalings_t o = fs.sc_io[out_idx];
            o.shift = (float)shift; o.scale = (float)scale; o.var = (float)var;
            fs.sc_io[out_idx] = o;
            fvar = o.var;
        }
        if (!calibrated || fvar > 2.5) flag |= ABEA_FAILED_CALIBRATION;       /* f5c.c:776-782 */
        else if (events_per_base > 5.0) flag |= ABEA_FAILED_QUALITY_CHK;      /* f5c.c:799-805 */
        fs.flag_io[out_idx] |= flag;
        fs.epb[out_idx] = events_per_base;
        fs.nalign[out_idx] = n_align;
        if (fs.var_f64) fs.var_f64[out_idx] = var;      /* align.c:760: log_var = log(var) in double, glibc's: done by the host */
    }
}


/* ================================================================ the fused alignment kernel
 * One wavefront = one read, from band 2 to the finished pair list:
 *   phase 1  band fill + adaptive band movement + online end-point scan      (VALU/DPP bound)
 *   phase 2  traceback walk over the packed trace, on the scalar unit        (SALU bound)
 *   phase 3  expansion of the walk's 2-bit codes into (k-mer, event) pairs,
 *            ordered fp64 emission sum, QC; base_to_event_map when phase 4 is on (small, VALU)
 *   phase 4  (optional) scaling_single: 'M' states, recalibrate_model's chains fed from LDS, flags
 * Fusing the phases lets the scalar-unit-bound walks of finished reads overlap the VALU-bound fills
 * of the other waves resident on the same CU, and removes two kernel boundaries per batch.
 *
 * Phase-1 state layout:
 *   lane l owns band offsets o0 = 2l and o1 = 2l+1.  Offsets 0..99 (lanes 0..49) are the band;
 *   offsets 100..127 (lanes 50..63) never hold scores (offset 100 is pinned to -inf, the rest is never read) but DO hold
 *   what enters the band next (round 4: no LDS rings): their k-mer quads are offsets 100..127 as ever — a "right" move is
 *   one DPP wave shift per register, so upcoming k-mers slide towards offset 99 — and lanes 52..63 of the EVENT
 *   registers hold the next 24 events, lane 63's cell 1 first: a "down" move shifts the event registers with wave_ror, so
 *   the next event arrives in lane 0 from lane 63.  Every 24th move of a kind the twelve lanes are overwritten under an
 *   EXEC mask from "pending" registers that two global loads filled 24 moves earlier (tools/gen_fill_asm.py).
 *   Neighbours (DESIGN.md "frames"): right move: left = P[o], up = P[o+1], diag = previous band's up;
 *   down move: left = P[o-1], up = P[o], diag = previous band's left.
 * Trace layout (per read): per 32 bands one uint4 per lane = 128 bits, 8 bands per dword with the OLDEST band in the
 *   top nibble: band (b & 31) at bits [4*((b&31)^7), +4) = {f(o0) | f(o1) << 2}, f = 2*[left == max] + [up >= diag]
 *   (0 FROM_D, 1 FROM_U, 2 and 3 FROM_L; the border variant only writes 0..2).  Lane 50 instead carries the band moves:
 *   .x = move bits of this group (bit 31-(b&31), 1 = right), .y = move bits of the group below. */

static __device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total) {
    int s = v;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(s, off, 64);
        if (lane >= off) s += t;
    }
    total = __shfl(s, 63, 64);
    return s - v;
}

/* four waves per SIMD: 128 VGPRs, of which the fill statement's fixed window takes v64..v127 */
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void abea_align_kernel(const abea_read_desc* __restrict__ descs,
                       const float* __restrict__ evm_all, const abea_kpar_t* __restrict__ kpar_all,
                       uint4* __restrict__ trace_all, uint32_t* __restrict__ codes_all,
                       abea_pair_t* __restrict__ pairs_all, int32_t* __restrict__ n_pairs,
                       abea_read_diag* __restrict__ diag,
                       unsigned long long* __restrict__ pair_cursor, int64_t* __restrict__ pair_off_out,
                       const abea_fused_scaling fs) {
    /* pairs_all == nullptr: the pair lists are not materialised on the device (the host entry expands them from the
     * walk codes).  pair_cursor != nullptr: pair lists are packed back to back in completion order (atomic bump
     * allocation of n entries per read, offset reported in pair_off_out[]) instead of at desc.pair_off. */
    /* 4 KiB of LDS per wavefront: phase 3's emission buffer (1024 floats in walk order).  The fill loop does not touch LDS:
     * upcoming events and k-mers wait in the idle lanes 52..63 of the band registers (below). */
    __shared__ __attribute__((aligned(4096))) uint4 smem[256];
    float* const lp_s = reinterpret_cast<float*>(smem);                      /* 1024 x 4 B */

    const abea_read_desc* d = descs + blockIdx.x;
    const int lane = threadIdx.x;
    const int out_idx = d->out_idx;
    const int n_groups = d->n_groups;
    abea_read_diag dg;
    dg.sum_emission = 0.0; dg.n_aligned = 0; dg.best_event = 0; dg.max_score = NINF;
    dg.max_gap = 0; dg.spanned = 0; dg.flags = 0; dg.pad = 0;
    if (n_groups == 0) {                               /* align_single guards (f5c.c:813-814) */
        if (lane == 0) { n_pairs[out_idx] = 0; dg.flags = ABEA_RF_SKIPPED; if (diag) diag[out_idx] = dg; }
        if (fs.b2e) abea_fused_not_aligned(fs, out_idx, lane);
        return;
    }
    const int E = d->n_events, K = d->n_kmers;
    const float* __restrict__ evm = evm_all + d->evm_off;
    const abea_kpar_t* __restrict__ kpar = kpar_all + d->kpar_off;
    uint4* __restrict__ trace = trace_all + d->trace_off;

    float best = NINF; int best_e = 0, best_llk = 0;
#ifdef ABEA_PROFILE_PHASES
    const unsigned long long t_start = wall_clock64();
#endif

    /* ============================================================ phase 1: band fill */
    {
    const int nb_pad = n_groups * ABEA_GROUP;
    const double lp_skip = d->lp_skip, lp_stay = d->lp_stay, lp_step = d->lp_step, lp_trim = d->lp_trim;
    const int o0 = 2 * lane, o1 = o0 + 1;              /* offsets owned by this lane */
    [[maybe_unused]] const bool hi = lane >= 50;        /* FIFO lanes: scores pinned to -inf */

    /* ---- state after bands 0 and 1 (align.c:277-291) ---- */
    int ll_e = 50, ll_k = -51;                          /* lower-left of band 1 */
    float Pf0 = NINF, Pf1 = NINF;                       /* band 1 scores (offset 50 = trim of event 0) */
    if (lane == 25) Pf0 = (float)lp_trim;
    [[maybe_unused]] double P0 = (double)Pf0, P1 = (double)Pf1;
    /* band 1 was a "down" move from band 0: in band-1 frame U[o] = band0[o], L[o] = band0[o-1];
     * band 0 is -inf except offset 50 (start cell, 0.0f) */
    double U0 = (lane == 25) ? 0.0 : (double)NINF, U1 = (double)NINF;
    double L0 = (double)NINF, L1 = (lane == 25) ? 0.0 : (double)NINF;

    /* per-offset inputs in band-1 frame: event = ll_e - o (offsets 0..127), kmer = ll_k + o */
    float x0, x1, g0, g1, c0, c1; double i0, i1;
    {
        const int e0 = ll_e - o0, e1 = ll_e - o1;
        x0 = (e0 >= 0 && e0 < E) ? evm[e0] : 0.f;
        x1 = (e1 >= 0 && e1 < E) ? evm[e1] : 0.f;
        const int k0 = ll_k + o0, k1 = ll_k + o1;
        abea_kpar_t z; z.gpm = 0.f; z.ck = 0.f; z.istd = 0.0;
        const abea_kpar_t p0 = (k0 >= 0 && k0 < K) ? kpar[k0] : z;
        const abea_kpar_t p1 = (k1 >= 0 && k1 < K) ? kpar[k1] : z;
        g0 = p0.gpm; c0 = p0.ck; i0 = p0.istd;
        g1 = p1.gpm; c1 = p1.ck; i1 = p1.istd;
    }
    /* Lanes 52..63 of the event registers hold the next 24 events (lane 63's cell 1 first; a down move rotates the wave);
     * their k-mer quads hold offsets 104..127 as they always did.  The pending registers are what the next refill, 24 moves
     * of a kind later, puts into those lanes. */
    float px0, px1, kag, kac, kbg, kbc; double kai, kbi;
    int e_cnt = 24, k_cnt = 24;
    {
        const int e_in = ll_e + 1, q = 2 * (63 - lane);
        if (lane >= 52) { x1 = evm[min(e_in + q, E - 1)]; x0 = evm[min(e_in + q + 1, E - 1)]; }
        px1 = evm[min(e_in + 24 + q, E - 1)]; px0 = evm[min(e_in + 24 + q + 1, E - 1)];
        const abea_kpar_t ta = kpar[min(max(ll_k + 24 + 2 * lane, 0), K - 1)];
        const abea_kpar_t tb = kpar[min(max(ll_k + 24 + 2 * lane + 1, 0), K - 1)];
        kag = ta.gpm; kac = ta.ck; kai = ta.istd; kbg = tb.gpm; kbc = tb.ck; kbi = tb.istd;
    }

    /* trace accumulator: 4 bits per band shifted in from the right, COMPLEMENTED (see abea_fill.inc); bands 0,1:
     * only band 1 offset 50 = FROM_U */
    uint32_t acc = (lane == 25) ? 0xFEu : 0xFFu;
    [[maybe_unused]] uint32_t a0 = 0; uint32_t a1 = 0, a2 = 0, a3 = 0;
    uint32_t mvacc = 0, mvprev = 0;                     /* band-move bits of this group / the group below */
    int b = 2;


    while (b < nb_pad) {
        /* bands for which every one of the 100 cells is inside the matrix and neither the trim column
         * nor the last k-mer column can be in band, whatever the moves: one of ll_e / ll_k grows per band */
        int run = min(min(E - 2 - ll_e, K - 102 - ll_k), nb_pad - b);
        {
            /* hand-scheduled loops (tools/gen_fill_asm.py; executed instruction by instruction against the oracle on the CPU by
             * tools/gfx950_emu.py, tests/test_asm_emulated.py).  Interior variant: `run` bands of align.c:300-410 while every
             * cell is provably in range.  Border variant (validity masks align.c:337-346, trim column align.c:324-333, online
             * end-point scan align.c:424-445): the first ~100 bands until ll_k >= 0 and ll_e >= 99, and everything after the
             * band has touched the bottom/right edge of the matrix (never interior again: ll_e, ll_k only grow) */
            uint32_t toff = (uint32_t)lane * 16u + (uint32_t)(b >> 5) * 1024u;
            uint32_t t0, t1, t2, t3, t4, cnt, per; uint64_t cm0a, cm0b, cm1a, cm1b, cv0, cv1;
            /* "s" operands must be provably wave-uniform */
            int s_ll_e = uni(ll_e), s_ll_k = uni(ll_k);
            int s_b = uni(b);
            const bool interior = (ll_k >= 0) && (ll_e >= 99) && (run > 0);
            const bool past_edge = (E - 2 - ll_e <= 0) || (K - 102 - ll_k <= 0);
            const int s_b_end = interior ? uni(b + run)
                              : past_edge ? uni(nb_pad)
                              : uni(min(nb_pad, b + max(max(-ll_k, 99 - ll_e), 256)));   /* chunk: re-check for the interior variant later */
            const int Km1 = K - 1, Em1 = E - 1;
            const uint64_t m50 = 1ull << ABEA_MOVE_LANE;
            uint32_t s_mvacc = (uint32_t)uni((int)mvacc), s_mvprev = (uint32_t)uni((int)mvprev);
            const double u_step = uni_d(lp_step), u_stay = uni_d(lp_stay), u_skip = uni_d(lp_skip), u_trim = uni_d(lp_trim);
            const float* u_evm = (const float*)uni_p(evm);
            const abea_kpar_t* u_kpar = (const abea_kpar_t*)uni_p(kpar);
            uint4* u_trace = (uint4*)uni_p(trace);
            uint32_t s_best = (uint32_t)uni((int)__float_as_uint(best));
            int s_best_e = uni(best_e), s_best_llk = uni(best_llk);
#define ABEA_FILL_OUTS \
                  [Pf0] "+v"(Pf0), [Pf1] "+v"(Pf1), [x0] "+v"(x0), [x1] "+v"(x1), \
                  [g0] "+v"(g0), [c0] "+v"(c0), [g1] "+v"(g1), [c1] "+v"(c1), \
                  [px0] "+v"(px0), [px1] "+v"(px1), [kag] "+v"(kag), [kac] "+v"(kac), [kbg] "+v"(kbg), [kbc] "+v"(kbc), \
                  [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [acc] "+v"(acc), [toff] "+v"(toff), \
                  [i0] "+v"(i0), [i1] "+v"(i1), [kai] "+v"(kai), [kbi] "+v"(kbi), \
                  [L0] "+v"(L0), [L1] "+v"(L1), [U0] "+v"(U0), [U1] "+v"(U1), [e_cnt] "+s"(e_cnt), [k_cnt] "+s"(k_cnt), \
                  [ll_e] "+s"(s_ll_e), [ll_k] "+s"(s_ll_k), \
                  [mvacc] "+s"(s_mvacc), [mvprev] "+s"(s_mvprev), [b] "+s"(s_b), \
                  [t0] "=&s"(t0), [t1] "=&s"(t1), [cnt] "=&s"(cnt), [per] "=&s"(per), [cm0a] "=&s"(cm0a), [cm0b] "=&s"(cm0b), \
                  [cm1a] "=&s"(cm1a), [cm1b] "=&s"(cm1b)
#define ABEA_FILL_INS \
                  [lane] "v"(lane), [lp_step] "s"(u_step), [lp_stay] "s"(u_stay), [lp_skip] "s"(u_skip), \
                  [Km1] "s"(uni(Km1)), [Em1] "s"(uni(Em1)), \
                  [b_end] "s"(s_b_end), [m50] "s"(m50), [evm] "s"(u_evm), [kpar] "s"(u_kpar), [trace] "s"(u_trace), \
                  [ninf] "s"(0xff800000u)
            asm volatile(ABEA_FILL_ASM
                : ABEA_FILL_OUTS, [t2] "=&s"(t2), [t3] "=&s"(t3), [t4] "=&s"(t4), [cv0] "=&s"(cv0), [cv1] "=&s"(cv1),
                  [best] "+s"(s_best), [best_e] "+s"(s_best_e), [best_llk] "+s"(s_best_llk)
                : ABEA_FILL_INS, [lp_trim] "s"(u_trim), [mode] "s"(uni(interior ? 0 : 1))
                : ABEA_FILL_CLOBBERS);
            best = __uint_as_float(s_best); best_e = s_best_e; best_llk = s_best_llk;
            ll_e = s_ll_e; ll_k = s_ll_k; b = s_b; run = 0;
            mvacc = s_mvacc; mvprev = s_mvprev;
            P0 = (double)Pf0; P1 = (double)Pf1;
        }
    }
    }
    __syncthreads();            /* this wave's trace stores are complete before it reads them back */
#ifdef ABEA_PROFILE_PHASES
    const unsigned long long t_fill = wall_clock64();
#endif

    /* ============================================================ phase 2: traceback walk */
    if (best == NINF) {                                  /* no in-band end cell, SURVEY §9-I */
        if (lane == 0) { n_pairs[out_idx] = 0; dg.flags = ABEA_RF_NO_END; if (diag) diag[out_idx] = dg; }
        if (fs.b2e) abea_fused_not_aligned(fs, out_idx, lane);
        return;
    }
    uint32_t* codes = codes_all + d->code_off;

    /* align.c:452-499, wave-uniform, on the scalar unit.  The trace of the current 32-band group sits
     * in 4 VGPRs (one uint4 per lane); the 128 bits of the lane pair the path is in are held in SGPRs,
     * so a step is pure SALU bit picking; v_readlane only when the path changes lane pair or group.
     * Each step emits a 2-bit code; 16 codes -> one dword, 64 dwords -> one coalesced store. */
    /* hand-written scalar walk (tools/gen_fill_asm.py: gen_walk) */
    int n, max_gap, last_k;
    uint32_t cwd, cv, walk_reloads;
    {
        uint32_t o_sh2, o_nfl;
        const uint4* u_trace = (const uint4*)uni_p(trace);
        uint32_t* u_codes = (uint32_t*)uni_p(codes);
        asm volatile(ABEA_WALK_ASM
            : [last_k] "=&s"(last_k), [o_cwd] "=&s"(cwd), [o_sh2] "=&s"(o_sh2), [o_nfl] "=&s"(o_nfl),
              [o_maxgap] "=&s"(max_gap), [o_reloads] "=&s"(walk_reloads), [o_cv] "=&v"(cv)
            : [k0] "s"(uni(K - 1)), [e0] "s"(uni(best_e)), [llk0] "s"(uni(best_llk)),
              [trace] "s"(u_trace), [codes] "s"(u_codes), [lane] "v"(lane)
            : ABEA_WALK_CLOBBERS);
        n = (int)(o_nfl * 16u + o_sh2 / 2u);
    }
    if ((n & 15) != 0 && lane == ((n >> 4) & 63)) cv = cwd;
    if ((n & 1023) != 0 && lane <= (((n - 1) >> 4) & 63)) codes[(size_t)(n >> 10) * 64 + lane] = cv;
    __syncthreads();                                     /* this wave's code words -> all its lanes */
#ifdef ABEA_PROFILE_PHASES
    const unsigned long long t_walk = wall_clock64();
#endif

    /* ============================================================ phase 3: expansion + QC
     * prefix sums turn codes into (k,e) pairs written in forward order; log-emissions are summed in
     * walk order, in double (align.c:473-476) */
    abea_pair_t* pairs = nullptr;
    if (pairs_all) {
        long long off = d->pair_off;
        if (pair_cursor) {
            unsigned long long o = 0;
            if (lane == 0) o = atomicAdd(pair_cursor, (unsigned long long)n);
            off = (long long)(((unsigned long long)(unsigned)uni((int)(o >> 32)) << 32) | (unsigned)uni((int)o));
            if (lane == 0) pair_off_out[out_idx] = off;
        }
        pairs = pairs_all + off;
    }
    double sum = 0.0;
    int base_k = K - 1, base_e = best_e;
    /* fused scaling_single: base_to_event_map (postalign, align.c:571-596) is written from here, straight from the walk — the pair
     * lists need not exist in HBM for it.  In walk order (step t = 0 at the end cell) code[t] is the move from pair t to pair t + 1,
     * i.e. to its PREDECESSOR in the list: 1 = same k-mer ("up"), 2 = same event ("left").  A pair opens its k-mer's run of the list
     * iff code[t] != 1 (or it is the last step), closes it iff code[t-1] != 1 (or t = 0), and is a new event iff code[t] != 2:
     * the rules of the pair-list formulation with prev / next replaced by the two codes (checked on the CPU against postalign). */
    abea_index_pair_t* const map = fs.b2e ? fs.b2e + d->kmer_off : nullptr;
    uint32_t carry_code = 0;                             /* code 15 of lane 63 of the previous 1024-step stretch */
    int e_first = -1;                                    /* read_pos of the list's first pair (walk step n - 1) */
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i0 = c0 + 16 * lane;
        const int cnt = max(0, min(16, n - i0));
        const uint32_t w = (cnt > 0) ? codes[(c0 >> 4) + lane] : 0u;
        uint32_t prev_top = (uint32_t)__shfl_up((int)(w >> 30), 1, 64);   /* code[t-1] of this lane's first step */
        if (lane == 0) prev_top = carry_code;
        int dk = 0, de = 0;
        #pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t cd = (w >> (2 * j)) & 3u;
            if (j < cnt) { dk += (cd != 1u); de += (cd != 2u); }
        }
        int tk, te;
        const int pk = wave_excl_scan(dk, lane, tk), pe = wave_excl_scan(de, lane, te);
        int kk = base_k - pk, ee = base_e - pe;
        #pragma unroll
        for (int j = 0; j < 16; ++j) {
            float lp = 0.f;
            if (j < cnt) {
                abea_pair_t p; p.ref_pos = kk; p.read_pos = ee;
                if (pairs) pairs[n - 1 - (i0 + j)] = p;
                const abea_kpar_t kp = kpar[kk];
                const float dx = __fsub_rn(evm[ee], kp.gpm);
                const float a = (float)((double)dx * kp.istd);
                lp = __fadd_rn(kp.ck, __fmul_rn(__fmul_rn(-0.5f, a), a));
                const uint32_t cd = (w >> (2 * j)) & 3u;
                if (map) {
                    const int t = i0 + j;
                    const uint32_t cprev = j ? (w >> (2 * (j - 1))) & 3u : prev_top;
                    const bool run_start = (t == n - 1) || cd != 1u;
                    const bool run_end = (t == 0) || cprev != 1u;
                    const bool is_new = (t == n - 1) || cd != 2u;
                    const int next_e = ee + ((t > 0 && cprev != 2u) ? 1 : 0);        /* read_pos of the successor in the list */
                    if (run_start) map[kk].start = is_new ? ee : (!run_end ? next_e : -1);
                    if (run_end) map[kk].stop = (!run_start || is_new) ? ee : -1;
                    if (t == n - 1) e_first = ee;
                }
                kk -= (cd != 1u); ee -= (cd != 2u);
            }
            lp_s[j * 64 + lane] = lp;                 /* [j][lane]: conflict-free stores */
        }
        __syncthreads();
        const int m = min(1024, n - c0);
        #pragma unroll 8
        for (int i = 0; i < m; ++i) sum += (double)lp_s[(i & 15) * 64 + (i >> 4)];   /* uniform, strictly in walk order */
        __syncthreads();
        base_k -= tk; base_e -= te;
        carry_code = (uint32_t)__shfl((int)(w >> 30), 63, 64);
    }

    /* ---- QC (align.c:526-543) ---- */
    const double avg = sum / (double)n;
    const int spanned = (last_k == 0);                   /* first emitted pair is always k = K-1 */
    const bool fail = (avg < -5.0) || !spanned || (max_gap > 50);
    if (lane == 0) {
        n_pairs[out_idx] = fail ? 0 : n;
        if (diag) {
            dg.sum_emission = sum; dg.n_aligned = n; dg.best_event = best_e;
            dg.max_score = best; dg.max_gap = max_gap; dg.spanned = spanned;
            dg.flags = fail ? ABEA_RF_QC_FAIL : 0;
            dg.pad = (int32_t)walk_reloads;              /* diagnostic: trace groups the walk had to load whole */
#ifdef ABEA_PROFILE_PHASES   /* experiment build only: overwrite the diagnostics with a phase timeline (100 MHz ticks) */
            const unsigned long long t_end = wall_clock64();
            dg.sum_emission = (double)t_start; dg.best_event = (int)(t_fill - t_start);
            dg.max_gap = (int)(t_walk - t_fill); dg.spanned = (int)(t_end - t_walk);
#endif
            diag[out_idx] = dg;
        }
    }

    /* ============================================================ phase 4 (optional): scaling_single for this read */
    if (fs.b2e) {
        if (fail) {                                      /* f5c.c:786-794; the map of a read that failed QC stays {-1, -1} */
            for (int kq = lane; kq < K; kq += 64) { abea_index_pair_t z; z.start = -1; z.stop = -1; map[kq] = z; }
            abea_fused_not_aligned(fs, out_idx, lane);
        } else {
            for (int off = 32; off > 0; off >>= 1) e_first = max(e_first, __shfl_xor(e_first, off, 64));   /* one lane had it */
            __syncthreads();                             /* phase 3's map stores are complete (and in L2) before the sweeps */
            abea_scaling_single_wave(d, fs, lane, best_e - e_first, evm, reinterpret_cast<double*>(smem), trace);
#ifdef ABEA_PROFILE_PHASES   /* experiment build: ticks from the end of the walk to the end of phase 4 into diag.pad (minus .spanned = phase 4) */
            if (lane == 0 && diag) diag[out_idx].pad = (int32_t)(wall_clock64() - t_walk);
#endif
        }
    }
}


/* ================================================================ results -> pinned host memory
 * The host pipeline's chunks return their (small) result block with this kernel instead of a hipMemcpyAsync: an SDMA
 * copy queued behind a 20 ms alignment kernel blocks its SDMA ring for that long, and the H2D copies of the following
 * chunks that land on the same ring start only when that kernel has finished (measured: profiles/r02_*).
 * A kernel on the chunk's own stream has no such side effect.  16 bytes per lane, coalesced, straight over PCIe. */
extern "C" __global__ __launch_bounds__(256)
void abea_copy_out_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* s = reinterpret_cast<const u32x4*>(src);
    u32x4* d = reinterpret_cast<u32x4*>(dst);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(s[i], d + i);
}


/* ================================================================ event detection on the device (row N2)
 * event_single's front half (f5c.c:682-712): ADC -> pA, getevents() (events.c:562-582 = detect_events on the
 * whole signal; the trim result there is discarded) and estimate_scalings_using_mom (align.c:58-106).
 * Everything in it is order-dependent floating point, so nothing is re-associated where that could change a bit.
 * Two forms share this section.  The COMMON PATH (round 6; "the common path without the arrays" below: abea_ev_spec2 / fix2 /
 * scan2 / create3 kernels) works straight from the samples for every read whose prefix sums are provably exact; the ARRAY
 * FORM listed here (rounds 3-5) keeps the prefix sums and t-statistics in HBM arrays: it runs behind the common path on the
 * reads that path flagged, and for every read under ABEA_EV_PATH=arrays.  The array form splits the work so that only
 * what is inherently sequential runs sequentially:
 *   pass 1  abea_ev_sums_kernel    lane-per-read: x = (adc+offset)*raw_unit, S[i+1] = S[i] + x, Q[i+1] = Q[i] + x*x
 *                                  (fp64, sample order; events.c:303-313)                      ~12 instr / sample
 *   pass 2  abea_ev_tstat_kernel   fully parallel over samples: the two windowed t-statistics (events.c:324-369)
 *   pass 3  the two-detector peak-picking automaton (events.c:380-452); its only output is the list of peak
 *           positions.  A fired peak resets its detector to a state that depends on the current sample alone, so a
 *           run started from the reset state anywhere in the read falls into step with the true trajectory within a
 *           few samples (measured: 4 on average, tools/proto/spec_detect.c).  That makes the automaton parallel over
 *           segments of ABEA_EV_SEG samples, exactly:
 *             3a abea_ev_spec_kernel   every (read, segment) from the reset state -> its peaks + end state
 *             3b abea_ev_fix_kernel    every segment again from its predecessor's end state, in lockstep with a replay
 *                                      of 3a, until the two states are equal: the true peaks of the prefix, how many of
 *                                      3a's peaks that prefix replaces, and a per-read flag if they never met
 *             3c abea_ev_scan_kernel   lane-per-read running sum of the per-segment peak counts
 *             3d abea_ev_gather_kernel peaks into the per-read list
 *             3e abea_ev_detect_kernel the sequential automaton, lane-per-read, for flagged reads only (by induction
 *                                      over segments an unflagged read's list is the sequential one)
 *   pass 4  abea_ev_create_kernel  fully parallel over events: event_t from the prefix sums at consecutive peaks
 *                                  (events.c:466-513)
 *   pass 5  abea_ev_scalings_kernel lane-per-read: method-of-moments scalings (align.c:58-106)
 * Scratch is interleaved per wavefront: sample i of the read in lane l of wave w lives at wave_base[w] + i*64 + l,
 * so the lane-per-read passes load/store 64 consecutive elements per instruction and pass 2 is a flat stream. */
struct abea_evdet {
    float peak_value; int peak_pos; long long masked_to; bool valid;
};

/* events.c:343-366 on the four window sums: sum1 / sumsq1 = left window (kept in double), sum2d / sumsq2d = right window
 * (rounded to float first, as the reference's float locals do) */
static __device__ __forceinline__ float abea_tstat_w(double sum1, double sum2d, double sumsq1, double sumsq2d, float wf) {
    const float sum2 = (float)sum2d;
    const float sumsq2 = (float)sumsq2d;
    const float mean1 = (float)(sum1 / (double)wf);
    const float mean2 = sum2 / wf;
    float combined_var = (float)(((sumsq1 / (double)wf - (double)(mean1 * mean1)) + (double)(sumsq2 / wf)) -
                                 (double)(mean2 * mean2));
    combined_var = fmaxf(combined_var, 1.17549435e-38f);            /* FLT_MIN */
    const float delta_mean = mean2 - mean1;
    return fabsf(delta_mean) / sqrtf(combined_var / wf);
}
static __device__ __forceinline__ float abea_tstat(double s_lo, double s_mid, double s_hi, double q_lo, double q_mid,
                                                   double q_hi, float wf) {
    return abea_tstat_w(s_mid - s_lo, s_hi - s_mid, q_mid - q_lo, q_hi - q_mid, wf);
}

/* ---- pass 1, parallel form.  S and Q are sequential fp64 sums (events.c:303-313), but when every sample of a read is a
 * multiple of one quantum 2^q and n * max|x| < 2^(q+53), every partial sum in ANY order is exactly representable, no
 * addition rounds, and a segmented scan gives the sequential result bit for bit.  1a sums segments of ABEA_EV_SEG
 * samples and records the exponent range of x and of the float squares; 1b scans the segment totals per read and
 * decides; 1c rewrites the prefix sums from the segment offsets; reads that fail the test (a sample near 0 pA among
 * ~100 pA ones, or a very long read) take the sequential kernel (1d). ---- */
#define ABEA_EV_SEG_SUM 512
static __device__ __forceinline__ float abea_pa(int raw, float offset, float raw_unit) {
    return __fmul_rn(__fadd_rn((float)raw, offset), raw_unit);       /* f5c.c:694-696 */
}

/* lo / hi = smallest nonzero / largest |value| of n floats as bit patterns (biased exponents; the quantum of a float with
 * exponent field e is 2^(max(e,1) - 150)); the sum of n values below 2^(emax + 1 - 127) stays below 2^(emax - 126 + nbits):
 * true = every partial sum of any subset, in any order, is exactly representable in a double */
static __device__ __forceinline__ bool abea_ev_sum_exact(uint32_t lo, uint32_t hi, int n) {
    if (hi == 0u) return true;                                       /* all zeros */
    if (hi >= 0x7f800000u) return false;                             /* inf / NaN: leave it to the sequential form */
    const int nbits = 32 - __clz(n);
    const int emin = max((int)(lo >> 23), 1), emax = max((int)(hi >> 23), 1);
    return (emax - emin) + nbits + 24 <= 53;
}

extern "C" __global__ __launch_bounds__(256)
void abea_ev_psum_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                         const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                         const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                         const int32_t* __restrict__ wave_nseg, double* __restrict__ segsum_all,
                         uint32_t* __restrict__ segexp_all) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    const int lo = j * ABEA_EV_SEG_SUM, hi = min(lo + ABEA_EV_SEG_SUM, n);
    if (lo >= hi) return;
    const int16_t* __restrict__ sig = signal + sig_ptr[r];
    const float offset = scaling[3 * r], raw_unit = scaling[3 * r + 1] / scaling[3 * r + 2];   /* f5c.c:693 */
    double S = 0.0, Q = 0.0;
    uint32_t xmin = 0x7f800000u, xmax = 0u, ymin = 0x7f800000u, ymax = 0u;
    auto take = [&](int raw) {
        const float x = abea_pa(raw, offset, raw_unit);
        const float y = __fmul_rn(x, x);
        S += (double)x; Q += (double)y;
        const uint32_t ax = __float_as_uint(x) & 0x7fffffffu, ay = __float_as_uint(y);
        xmax = max(xmax, ax); ymax = max(ymax, ay);
        xmin = min(xmin, ax ? ax : 0x7f800000u); ymin = min(ymin, ay ? ay : 0x7f800000u);
    };
    int p = lo;
    if ((((uintptr_t)(sig + lo)) & 15u) == 0u)                       /* sig_ptr a multiple of 8 samples: 16-byte loads */
        for (; p + 8 <= hi; p += 8) {
            const uint4 q = *reinterpret_cast<const uint4*>(sig + p);
            take((int)(short)(q.x & 0xffffu)); take((int)(short)(q.x >> 16)); take((int)(short)(q.y & 0xffffu)); take((int)(short)(q.y >> 16));
            take((int)(short)(q.z & 0xffffu)); take((int)(short)(q.z >> 16)); take((int)(short)(q.w & 0xffffu)); take((int)(short)(q.w >> 16));
        }
    for (; p < hi; ++p) take(sig[p]);
    const int64_t seg = seg_base[w] + j;
    double* __restrict__ ss = segsum_all + seg * 2 * 64 + lane;
    uint32_t* __restrict__ se = segexp_all + seg * 4 * 64 + lane;
    ss[0] = S; ss[64] = Q;
    se[0] = xmin; se[64] = xmax; se[128] = ymin; se[192] = ymax;
}

extern "C" __global__ __launch_bounds__(64)
void abea_ev_pscan_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                          const int64_t* __restrict__ seg_base, double* __restrict__ segsum_all,
                          const uint32_t* __restrict__ segexp_all, int32_t* __restrict__ need_seq) {
    const int lane = threadIdx.x;
    const int slot = blockIdx.x * 64 + lane;
    if (slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    if (n <= 0) return;
    const int nseg = (n + ABEA_EV_SEG_SUM - 1) / ABEA_EV_SEG_SUM;
    double* __restrict__ ss = segsum_all + seg_base[blockIdx.x] * 2 * 64 + lane;
    const uint32_t* __restrict__ se = segexp_all + seg_base[blockIdx.x] * 4 * 64 + lane;
    double S = 0.0, Q = 0.0;
    uint32_t xmin = 0x7f800000u, xmax = 0u, ymin = 0x7f800000u, ymax = 0u;
    for (int j = 0; j < nseg; ++j) {
        const double s = ss[0], q = ss[64];
        ss[0] = S; ss[64] = Q;                                       /* exclusive offsets for 1c */
        S += s; Q += q;
        xmin = min(xmin, se[0]); xmax = max(xmax, se[64]); ymin = min(ymin, se[128]); ymax = max(ymax, se[192]);
        ss += 2 * 64; se += 4 * 64;
    }
    need_seq[r] = (abea_ev_sum_exact(xmin, xmax, n) && abea_ev_sum_exact(ymin, ymax, n)) ? 0 : 1;
}

extern "C" __global__ __launch_bounds__(256)
void abea_ev_pwrite_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                           const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                           const float* __restrict__ scaling, const int64_t* __restrict__ wave_base,
                           const int64_t* __restrict__ seg_base, const int32_t* __restrict__ wave_nseg,
                           const double* __restrict__ segsum_all, const int32_t* __restrict__ need_seq,
                           double* __restrict__ S_all, double* __restrict__ Q_all) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    if (need_seq[r]) return;
    const int n = n_samples[r];
    const int lo = j * ABEA_EV_SEG_SUM, hi = min(lo + ABEA_EV_SEG_SUM, n);
    /* round 4: S and Q of a sample sit side by side (one 16-byte element at [row][lane]; the host allocates the two arrays
     * back to back, Q_all = S_all + entries, and the pair array overlays them): the event pass looks both up per boundary,
     * and a look-up costs a 64-byte sector whatever it reads */
    (void)Q_all;
    double2* __restrict__ SQw = reinterpret_cast<double2*>(S_all) + wave_base[w] + lane;
    if (j == 0) SQw[0] = make_double2(0.0, 0.0);
    if (lo >= hi) return;
    const int16_t* __restrict__ sig = signal + sig_ptr[r];
    const float offset = scaling[3 * r], raw_unit = scaling[3 * r + 1] / scaling[3 * r + 2];
    const double* __restrict__ ss = segsum_all + (seg_base[w] + j) * 2 * 64 + lane;
    double S = ss[0], Q = ss[64];
    auto take = [&](int p, int raw) {
        const float x = abea_pa(raw, offset, raw_unit);
        S += (double)x; Q += (double)__fmul_rn(x, x);
        SQw[(size_t)(p + 1) * 64] = make_double2(S, Q);
    };
    int p = lo;
    if ((((uintptr_t)(sig + lo)) & 15u) == 0u)
        for (; p + 8 <= hi; p += 8) {
            const uint4 q = *reinterpret_cast<const uint4*>(sig + p);
            take(p + 0, (int)(short)(q.x & 0xffffu)); take(p + 1, (int)(short)(q.x >> 16));
            take(p + 2, (int)(short)(q.y & 0xffffu)); take(p + 3, (int)(short)(q.y >> 16));
            take(p + 4, (int)(short)(q.z & 0xffffu)); take(p + 5, (int)(short)(q.z >> 16));
            take(p + 6, (int)(short)(q.w & 0xffffu)); take(p + 7, (int)(short)(q.w >> 16));
        }
    for (; p < hi; ++p) take(p, sig[p]);
}

extern "C" __global__ __launch_bounds__(64)
void abea_ev_sums_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                         const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                         const float* __restrict__ scaling, const int64_t* __restrict__ wave_base,
                         double* __restrict__ S_all, double* __restrict__ Q_all, const int32_t* __restrict__ need_seq) {
    const int lane = threadIdx.x;
    const int slot = blockIdx.x * 64 + lane;
    if (slot >= n_reads) return;
    const int r = order[slot];
    if (need_seq && !need_seq[r]) return;                           /* pass 1d: only reads whose sums may round */
    const int n = n_samples[r];
    const int16_t* __restrict__ sig = signal + sig_ptr[r];
    const float offset = scaling[3 * r], range = scaling[3 * r + 1], digitisation = scaling[3 * r + 2];
    const float raw_unit = range / digitisation;                    /* f5c.c:693 */
    (void)Q_all;
    double2* __restrict__ SQw = reinterpret_cast<double2*>(S_all) + wave_base[blockIdx.x] + lane;   /* {S, Q} pairs, see abea_ev_pwrite_kernel */
    double S = 0.0, Q = 0.0;
    SQw[0] = make_double2(0.0, 0.0);                                /* S[0] = Q[0] = 0 */
    auto sample = [&](int i, int raw) {
        const float x = ((float)raw + offset) * raw_unit;            /* f5c.c:694-696 */
        S = S + (double)x;                                           /* events.c:309-312; the square is a float product */
        Q = Q + (double)(x * x);
        SQw[(size_t)(i + 1) * 64] = make_double2(S, Q);
    };
    int i = 0;
    const int head = min(n, (int)(((16u - ((uintptr_t)sig & 15u)) & 15u) >> 1));
    for (; i < head; ++i) sample(i, sig[i]);
    /* every lane streams its own read, so a 16-byte load is 64 different cache lines and takes ~1-2 us: keep 32
     * samples (four loads) in flight ahead of the 32 being summed */
    auto load4 = [&](int at, uint4* q) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const uint4*>(sig + at + 8 * u);   /* callers keep at + 32 <= n */
    };
    if (i + 32 <= n) {
        uint4 nxt[4];
        load4(i, nxt);
        while (i + 32 <= n) {
            uint4 cur[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            if (i + 64 <= n) load4(i + 32, nxt);
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = i + 8 * u;
                sample(b + 0, (int)(short)(cur[u].x & 0xffffu)); sample(b + 1, (int)(short)(cur[u].x >> 16));
                sample(b + 2, (int)(short)(cur[u].y & 0xffffu)); sample(b + 3, (int)(short)(cur[u].y >> 16));
                sample(b + 4, (int)(short)(cur[u].z & 0xffffu)); sample(b + 5, (int)(short)(cur[u].z >> 16));
                sample(b + 6, (int)(short)(cur[u].w & 0xffffu)); sample(b + 7, (int)(short)(cur[u].w >> 16));
            }
            i += 32;
        }
    }
    for (; i < n; ++i) sample(i, sig[i]);
}

/* detector parameters (events.c:52-65): DNA event_detection_defaults, RNA event_detection_rna */
struct abea_ev_par { int w1, w2, half1, half2; float thr1, thr2, h; };
static __device__ __forceinline__ abea_ev_par ev_par(int rna) {
    abea_ev_par P;
    P.w1 = rna ? 7 : 3; P.w2 = rna ? 14 : 6;
    P.half1 = P.w1 / 2; P.half2 = P.w2 / 2;                          /* window_length / 2, integer (events.c:441) */
    P.thr1 = rna ? 2.5f : 1.4f; P.thr2 = 9.0f;
    P.h = rna ? 1.0f : 0.2f;
    return P;
}

/* windows W1 < W2; a thread makes POS consecutive positions of its read from one window of POS + 2*W2 prefix-sum rows
 * (p0-W2 .. p0+POS-1+W2) */
template <int W1, int W2, int POS>
static __device__ __forceinline__ void tstat_body(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                          const int64_t* __restrict__ wave_base, const int32_t* __restrict__ wave_len,
                          const double* __restrict__ S_all, const double* __restrict__ Q_all,
                          float* __restrict__ t1_all, float* __restrict__ t2_all, const int32_t* __restrict__ need) {
    /* grid.y = wave, grid.x tiles the positions of that wave; a 256-thread block covers 4 positions x 64 lanes */
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int slot = w * 64 + lane;
    /* need != NULL: this is the array form behind the fused common path — only flagged reads have prefix sums at all */
    if (slot >= n_reads || (need && !need[order[slot]])) return;
    const int n = n_samples[order[slot]];
    const int64_t base = wave_base[w];
    const int len = wave_len[w];
    (void)Q_all;
    const double2* SQw = reinterpret_cast<const double2*>(S_all) + base + lane;
    constexpr int ROWS = POS + 2 * W2;
    for (int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * POS; p0 < len; p0 += gridDim.x * 4 * POS) {
        double sr[ROWS], qr[ROWS];
        #pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = min(max(p0 - W2 + i, 0), len - 1);
            const double2 v = SQw[(int64_t)row * 64];
            sr[i] = v.x; qr[i] = v.y;
        }
        #pragma unroll
        for (int i = 0; i < POS; ++i) {
            const int p = p0 + i;
            if (p >= len) break;
            float a = 0.f, b = 0.f;
            if (p < n) {                                             /* events.c:336-347: zero at the boundaries */
                if (n >= 2 * W1 && p >= W1 && p <= n - W1)
                    a = abea_tstat(sr[i + W2 - W1], sr[i + W2], sr[i + W2 + W1], qr[i + W2 - W1], qr[i + W2], qr[i + W2 + W1], (float)W1);
                if (n >= 2 * W2 && p >= W2 && p <= n - W2)
                    b = abea_tstat(sr[i], sr[i + W2], sr[i + 2 * W2], qr[i], qr[i + W2], qr[i + 2 * W2], (float)W2);
            }
            t1_all[base + (int64_t)p * 64 + lane] = a;
            t2_all[base + (int64_t)p * 64 + lane] = b;
        }
    }
}

extern "C" __global__ __launch_bounds__(256)
void abea_ev_tstat_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                          const int64_t* __restrict__ wave_base, const int32_t* __restrict__ wave_len,
                          const double* __restrict__ S_all, const double* __restrict__ Q_all,
                          float* __restrict__ t1_all, float* __restrict__ t2_all, int rna,
                          const int32_t* __restrict__ need) {
    if (rna) tstat_body<7, 14, 4>(n_reads, order, n_samples, wave_base, wave_len, S_all, Q_all, t1_all, t2_all, need);
    else tstat_body<3, 6, 8>(n_reads, order, n_samples, wave_base, wave_len, S_all, Q_all, t1_all, t2_all, need);
}

/* ---- pass 3: segment-parallel automaton ---- */
#define ABEA_EV_SEG   512     /* samples per segment */
#define ABEA_EV_FIX   64      /* DNA: a segment must meet its speculative replay within this many samples */
#define ABEA_EV_FIX_RNA 128   /* RNA: longer dwells, rarer peaks, so the replay gets a longer window */
#define ABEA_EV_FIXCAP 48     /* peaks the true prefix may emit inside the window.  DNA: the short detector fires at most
                               * every 3rd sample, the long one every 5th: 64/3 + 64/5 + 2 = 36; RNA (half windows 3 / 7):
                               * every 5th / 9th: 128/5 + 128/9 + 2 = 41.  More than the cap flags the read for the
                               * sequential kernel, so the bound is a performance matter, not a correctness one. */

struct abea_det2 { float pv0, pv1; int pp0, pp1; int v0, v1; int masked; };

static __device__ __forceinline__ void det2_reset(abea_det2& s) {
    s.pv0 = s.pv1 = 3.402823466e+38f; s.pp0 = s.pp1 = -1; s.v0 = s.v1 = 0; s.masked = -1;
}
/* events.c:380-452 at position p for both detectors, with selects.  Returns bit 0 / bit 1 = short / long detector
 * fired; f0 / f1 = the peak positions they emit (short first, as in the reference's k loop). */
static __device__ __forceinline__ int det2_step(abea_det2& s, int p, float c0, float c1, int& f0, int& f1, const abea_ev_par& P) {
    const float h = P.h;                                             /* peak_height, events.c:52-65 */
    int fired = 0;
    {   /* short window (DNA: threshold 1.4, window 3); nothing ever masks it after position 0 */
        const bool srch = s.pp0 == -1;
        const bool lower = c0 < s.pv0;
        const bool rise = !lower && (c0 - s.pv0 > h);
        const bool higher = c0 > s.pv0;
        const float pv_i = higher ? c0 : s.pv0;
        const int pp_i = higher ? p : s.pp0;
        const bool dom = !srch && (pv_i > P.thr1);                   /* masks the long detector (events.c:418-424) */
        const bool valid_i = s.v0 || ((pv_i - c0 > h) && (pv_i > P.thr1));
        const bool fire = !srch && valid_i && ((p - pp_i) > P.half1);
        f0 = pp_i;
        fired |= fire ? 1 : 0;
        s.pv0 = srch ? ((lower || rise) ? c0 : s.pv0) : (fire ? c0 : pv_i);
        s.pp0 = srch ? (rise ? p : -1) : (fire ? -1 : pp_i);
        s.v0 = srch ? s.v0 : (fire ? 0 : (valid_i ? 1 : 0));
        s.masked = dom ? pp_i + P.w1 : s.masked;
        s.pp1 = dom ? -1 : s.pp1;
        s.pv1 = dom ? 3.402823466e+38f : s.pv1;
        s.v1 = dom ? 0 : s.v1;
    }
    {   /* long window (DNA: threshold 9.0, window 6) */
        const bool active = s.masked < p;
        const bool srch = s.pp1 == -1;
        const bool lower = c1 < s.pv1;
        const bool rise = !lower && (c1 - s.pv1 > h);
        const bool higher = c1 > s.pv1;
        const float pv_i = higher ? c1 : s.pv1;
        const int pp_i = higher ? p : s.pp1;
        const bool inpeak = active && !srch;
        const bool valid_i = s.v1 || ((pv_i - c1 > h) && (pv_i > P.thr2));
        const bool fire = inpeak && valid_i && ((p - pp_i) > P.half2);
        f1 = pp_i;
        fired |= fire ? 2 : 0;
        const bool sr = active && srch;
        s.pv1 = sr ? ((lower || rise) ? c1 : s.pv1) : (inpeak ? (fire ? c1 : pv_i) : s.pv1);
        s.pp1 = sr ? (rise ? p : -1) : (inpeak ? (fire ? -1 : pp_i) : s.pp1);
        s.v1 = inpeak ? (fire ? 0 : (valid_i ? 1 : 0)) : s.v1;
    }
    return fired;
}
/* equal as far as any position > p can tell (a mask that has expired is no mask) */
static __device__ __forceinline__ bool det2_equal(const abea_det2& a, const abea_det2& b, int p) {
    return __float_as_int(a.pv0) == __float_as_int(b.pv0) && __float_as_int(a.pv1) == __float_as_int(b.pv1) &&
           a.pp0 == b.pp0 && a.pp1 == b.pp1 && a.v0 == b.v0 && a.v1 == b.v1 && max(a.masked, p) == max(b.masked, p);
}

/* Segment records are interleaved like everything else: field f of segment j of the read in lane l of read-wave w is
 * at ((seg_base[w] + j) * stride + f) * 64 + l. */
extern "C" __global__ __launch_bounds__(256)
void abea_ev_spec_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                         const int64_t* __restrict__ wave_base, const float* __restrict__ t1_all,
                         const float* __restrict__ t2_all, const int64_t* __restrict__ seg_base,
                         const int32_t* __restrict__ wave_nseg, uint16_t* __restrict__ spec_all,
                         int32_t* __restrict__ segrec_all, int rna) {
    const abea_ev_par P = ev_par(rna);
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    const int lo = max(j * ABEA_EV_SEG, 1);                          /* masked_to starts at 0: position 0 is skipped */
    const int hi = min((j + 1) * ABEA_EV_SEG, n);
    if (lo >= hi && j > 0) return;
    const int64_t base = wave_base[w] + lane;
    const float* __restrict__ t1 = t1_all + base;
    const float* __restrict__ t2 = t2_all + base;
    const int64_t seg = seg_base[w] + j;
    uint16_t* __restrict__ out = spec_all + seg * ABEA_EV_SEG * 64 + lane;
    int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + lane;
    abea_det2 s; det2_reset(s);
    int cnt = 0;
    const int last = max(n - 1, 0);
    float a1[8], a2[8];
    #pragma unroll
    for (int q = 0; q < 8; ++q) { a1[q] = t1[(size_t)min(lo + q, last) * 64]; a2[q] = t2[(size_t)min(lo + q, last) * 64]; }
    for (int p0 = lo; p0 < hi; p0 += 8) {
        float b1[8], b2[8];
        #pragma unroll
        for (int q = 0; q < 8; ++q) { b1[q] = t1[(size_t)min(p0 + 8 + q, last) * 64]; b2[q] = t2[(size_t)min(p0 + 8 + q, last) * 64]; }
        #pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (p0 + q < hi) {
                int f0, f1;
                const int fired = det2_step(s, p0 + q, a1[q], a2[q], f0, f1, P);
                if (fired & 1) { out[(size_t)cnt * 64] = (uint16_t)(f0 - j * ABEA_EV_SEG); ++cnt; }
                if (fired & 2) { out[(size_t)cnt * 64] = (uint16_t)(f1 - j * ABEA_EV_SEG); ++cnt; }
            }
        }
        #pragma unroll
        for (int q = 0; q < 8; ++q) { a1[q] = b1[q]; a2[q] = b2[q]; }
    }
    rec[0 * 64] = cnt;                                               /* peaks of the speculative run */
    rec[1 * 64] = 0;                                                 /* peaks of the true prefix (3b) */
    rec[2 * 64] = 0;                                                 /* speculative peaks the prefix replaces (3b) */
    rec[4 * 64] = __float_as_int(s.pv0); rec[5 * 64] = __float_as_int(s.pv1);
    rec[6 * 64] = s.pp0; rec[7 * 64] = s.pp1; rec[8 * 64] = s.v0; rec[9 * 64] = s.v1; rec[10 * 64] = s.masked;
}

extern "C" __global__ __launch_bounds__(256)
void abea_ev_fix_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                        const int64_t* __restrict__ wave_base, const float* __restrict__ t1_all,
                        const float* __restrict__ t2_all, const int64_t* __restrict__ seg_base,
                        const int32_t* __restrict__ wave_nseg, int32_t* __restrict__ fix_all,
                        int32_t* __restrict__ segrec_all, int32_t* __restrict__ need_seq, int rna) {
    const abea_ev_par P = ev_par(rna);
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;           /* segment 0 starts from the true state already */
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    const int lo = j * ABEA_EV_SEG;
    const int hi = min(lo + ABEA_EV_SEG, n);
    if (lo >= hi) return;
    const int64_t base = wave_base[w] + lane;
    const float* __restrict__ t1 = t1_all + base;
    const float* __restrict__ t2 = t2_all + base;
    const int64_t seg = seg_base[w] + j;
    int32_t* __restrict__ out = fix_all + seg * ABEA_EV_FIXCAP * 64 + lane;
    int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + lane;
    const int32_t* __restrict__ prev = rec - 12 * 64;
    abea_det2 tr, sp;
    tr.pv0 = __int_as_float(prev[4 * 64]); tr.pv1 = __int_as_float(prev[5 * 64]);
    tr.pp0 = prev[6 * 64]; tr.pp1 = prev[7 * 64]; tr.v0 = prev[8 * 64]; tr.v1 = prev[9 * 64]; tr.masked = prev[10 * 64];
    det2_reset(sp);
    int nfix = 0, skip = 0, p = lo;
    bool met = det2_equal(tr, sp, lo - 1);
    const int stop = min(hi, lo + (rna ? ABEA_EV_FIX_RNA : ABEA_EV_FIX));
    while (!met && p < stop) {
        const float c0 = t1[(size_t)p * 64], c1 = t2[(size_t)p * 64];
        int f0, f1, g0, g1;
        const int ft = det2_step(tr, p, c0, c1, f0, f1, P);
        const int fs = det2_step(sp, p, c0, c1, g0, g1, P);
        if (ft & 1) { if (nfix < ABEA_EV_FIXCAP) out[(size_t)nfix * 64] = f0; ++nfix; }
        if (ft & 2) { if (nfix < ABEA_EV_FIXCAP) out[(size_t)nfix * 64] = f1; ++nfix; }
        skip += (fs & 1) + ((fs >> 1) & 1);
        met = det2_equal(tr, sp, p);
        ++p;
    }
    /* a segment shorter than the window that never met still ends the read: nothing follows it, but its own peaks
     * after the window are unknown, so it is flagged too unless the window covered it */
    if ((!met && p < hi) || nfix > ABEA_EV_FIXCAP) need_seq[r] = 1;
    if (!met && p >= hi) skip = rec[0 * 64];                         /* the true run covered the whole segment */
    rec[1 * 64] = nfix;
    rec[2 * 64] = skip;
}

extern "C" __global__ __launch_bounds__(64)
void abea_ev_scan_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                         const int64_t* __restrict__ seg_base, int32_t* __restrict__ segrec_all,
                         int32_t* __restrict__ n_events) {
    const int lane = threadIdx.x;
    const int slot = blockIdx.x * 64 + lane;
    if (slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    if (n <= 0) { n_events[r] = 0; return; }
    const int nseg = (n + ABEA_EV_SEG - 1) / ABEA_EV_SEG;
    int32_t* __restrict__ rec = segrec_all + seg_base[blockIdx.x] * 12 * 64 + lane;
    int run = 0;
    for (int j = 0; j < nseg; ++j) {
        const int cnt = rec[0] + rec[64] - rec[128];
        rec[3 * 64] = run;
        run += cnt;
        rec += 12 * 64;
    }
    n_events[r] = run + 1;                                           /* events.c:491-497: one more event than peaks */
}

extern "C" __global__ __launch_bounds__(256)
void abea_ev_gather_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                           const int64_t* __restrict__ seg_base, const int32_t* __restrict__ wave_nseg,
                           const uint16_t* __restrict__ spec_all, const int32_t* __restrict__ fix_all,
                           const int32_t* __restrict__ segrec_all, const int64_t* __restrict__ peak_base,
                           const int32_t* __restrict__ event_cap, int32_t* __restrict__ peaks_all) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    if (j * ABEA_EV_SEG >= n) return;
    const int cap = event_cap[r];
    const int64_t seg = seg_base[w] + j;
    const int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + lane;
    const uint16_t* __restrict__ sp = spec_all + seg * ABEA_EV_SEG * 64 + lane;
    const int32_t* __restrict__ fx = fix_all + seg * ABEA_EV_FIXCAP * 64 + lane;
    int32_t* __restrict__ pk = peaks_all + peak_base[w] + lane;
    const int nspec = rec[0], nfix = min(rec[64], ABEA_EV_FIXCAP), skip = rec[128];
    int at = rec[3 * 64];
    for (int e = 0; e < nfix; ++e, ++at) if (at < cap) pk[(size_t)at * 64] = fx[(size_t)e * 64];
    for (int e = skip; e < nspec; ++e, ++at) if (at < cap) pk[(size_t)at * 64] = j * ABEA_EV_SEG + (int)sp[(size_t)e * 64];
}

extern "C" __global__ __launch_bounds__(64)
void abea_ev_detect_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                           const int64_t* __restrict__ wave_base, const float* __restrict__ t1_all,
                           const float* __restrict__ t2_all, const int64_t* __restrict__ peak_base,
                           const int32_t* __restrict__ event_cap, int32_t* __restrict__ peaks_all,
                           int32_t* __restrict__ n_events, const int32_t* __restrict__ need_seq, int rna) {
    const abea_ev_par P = ev_par(rna);
    const int lane = threadIdx.x;
    const int slot = blockIdx.x * 64 + lane;
    if (slot >= n_reads) return;
    const int r = order[slot];
    if (need_seq && !need_seq[r]) return;                           /* pass 3e: only reads whose segments never met */
    const int n = n_samples[r];
    const int cap = event_cap[r];
    if (n <= 0) { n_events[r] = 0; return; }
    const int64_t base = wave_base[blockIdx.x] + lane;
    const float* __restrict__ t1 = t1_all + base;
    const float* __restrict__ t2 = t2_all + base;
    int32_t* __restrict__ pk = peaks_all + peak_base[blockIdx.x] + lane;

    abea_evdet det[2];
    for (int k = 0; k < 2; ++k) {
        det[k].peak_value = 3.402823466e+38f; det[k].peak_pos = -1; det[k].masked_to = 0; det[k].valid = false;
    }
    const float thr[2] = {P.thr1, P.thr2};                           /* events.c:52-65 */
    const int win[2] = {P.w1, P.w2};
    const float peak_height = P.h;
    int n_pk = 0;

    /* events.c:380-452 at position p, written with selects instead of branches: the 64 reads of a wavefront are in
     * unrelated automaton states, so branchy code would execute every path anyway */
    auto detect_at = [&](int p, float ts0, float ts1) {
        const float ts[2] = {ts0, ts1};
        #pragma unroll
        for (int k = 0; k < 2; ++k) {
            abea_evdet& d = det[k];
            const float cur = ts[k];
            const bool active = d.masked_to < (long long)p;
            const bool searching = d.peak_pos == -1;
            /* CASE 1: no maximum recorded yet */
            const bool lower = cur < d.peak_value;
            const bool rise = !lower && (cur - d.peak_value > peak_height);
            /* CASE 2: in a peak */
            const bool higher = cur > d.peak_value;
            const float pv_i = higher ? cur : d.peak_value;
            const int pp_i = higher ? p : d.peak_pos;
            const bool inpeak = active && !searching;
            if (k == 0) {                                            /* a short-window peak that will fire masks the long detector */
                const bool dom = inpeak && (pv_i > thr[0]);
                det[1].masked_to = dom ? (long long)pp_i + win[0] : det[1].masked_to;
                det[1].peak_pos = dom ? -1 : det[1].peak_pos;
                det[1].peak_value = dom ? 3.402823466e+38f : det[1].peak_value;
                det[1].valid = dom ? false : det[1].valid;
            }
            const bool valid_i = d.valid || ((pv_i - cur > peak_height) && (pv_i > thr[k]));
            const bool fire = inpeak && valid_i && ((unsigned)(p - pp_i) > (unsigned)(win[k] / 2));
            if (fire) {                                              /* peaks[peak_count++] (events.c:443) */
                if (n_pk < cap) pk[(size_t)n_pk * 64] = pp_i;
                ++n_pk;
            }
            const bool srch = active && searching;
            d.peak_value = srch ? ((lower || rise) ? cur : d.peak_value) : (inpeak ? (fire ? cur : pv_i) : d.peak_value);
            d.peak_pos = srch ? (rise ? p : -1) : (inpeak ? (fire ? -1 : pp_i) : d.peak_pos);
            d.valid = inpeak ? (fire ? false : valid_i) : d.valid;
        }
    };
    /* the automaton is a serial chain with no loads of its own: its inputs are fetched 8 positions at a time, two
     * blocks ahead, and a fired peak is a 4-byte store */
    const int last = max(n - 1, 0);
    float a1[8], a2[8], b1[8], b2[8];
    #pragma unroll
    for (int j = 0; j < 8; ++j) {
        a1[j] = t1[(size_t)min(j, last) * 64]; a2[j] = t2[(size_t)min(j, last) * 64];
        b1[j] = t1[(size_t)min(8 + j, last) * 64]; b2[j] = t2[(size_t)min(8 + j, last) * 64];
    }
    for (int p0 = 0; p0 < n; p0 += 8) {
        float c1[8], c2[8];
        #pragma unroll
        for (int j = 0; j < 8; ++j) {
            c1[j] = t1[(size_t)min(p0 + 16 + j, last) * 64]; c2[j] = t2[(size_t)min(p0 + 16 + j, last) * 64];
        }
        #pragma unroll
        for (int j = 0; j < 8; ++j) if (p0 + j < n) detect_at(p0 + j, a1[j], a2[j]);
        #pragma unroll
        for (int j = 0; j < 8; ++j) { a1[j] = b1[j]; a2[j] = b2[j]; b1[j] = c1[j]; b2[j] = c2[j]; }
    }
    n_events[r] = n_pk + 1;                                          /* events.c:491-497: one more event than peaks */
}

/* ================================================================ the common path without the arrays (round 6)
 * Rounds 3-5 moved 128 bytes of HBM traffic per sample for 6.6 algorithmic (profiles/pmc_traffic.json -> detector): the fp64 prefix
 * sums {S, Q} written once (16 B per sample) and fetched back by the t-statistics and by the event creation, the two float
 * t-statistics (8 B) written and fetched by the automaton.  None of that is needed where the prefix sums are EXACT (every real
 * read; the test of abea_ev_pscan_kernel): a difference S[b] - S[a] of two exact prefix sums is the exact sum of the samples in
 * [a, b), which any order of fp64 additions reproduces bit for bit as long as no partial sum can round — so
 *   abea_ev_spec2_kernel   a lane walks its (read, segment) once, straight from the 2-byte samples: the four window sums slide
 *                          (+ entering sample, - leaving sample, all exact), the two t-statistics come from them with the very
 *                          expressions of events.c:343-366 (abea_tstat_w) and go into the automaton in the same step — no
 *                          prefix-sum array, no t-statistic array;
 *   abea_ev_fix2_kernel    the replay window of abea_ev_fix_kernel with the t-statistics recomputed the same way;
 *   abea_ev_scan2_kernel   a wavefront per read: where each segment's events start, and the exponent ranges spec2 recorded ->
 *                          a read whose sums may round is flagged;
 *   abea_ev_create3_kernel a lane per event, straight from the segments' peak lists: the event's sums are added up from its samples.
 * A flagged read (sums may round, or a segment that never met its replay) goes through the array kernels above, which run
 * behind the common path on flagged reads only: sequential prefix sums, t-statistic arrays, the sequential automaton, events
 * from the prefix sums.  ABEA_EV_PATH=arrays selects the array form for every read (the A/B of profiles/r06).
 * Samples reach a lane through a private LDS row: a lane is in the middle of ITS read, 64 lanes of a wavefront in 64 different
 * reads, so a load touches 64 cache lines; fetching a whole 128-byte line per lane at a time (eight 16-byte loads back to back)
 * uses every byte of a line while it is hot — the lane-per-read kernels that take 16 bytes per line and come back for the next
 * 16 a microsecond later re-fetch lines (abea_ev_psum_kernel: 10 B per sample counted for 2 read). */
/* The t-statistic with its five divisions by the window length W in 3 instructions each instead of the 10 (float) / 12 (double) of
 * an IEEE division:    q0 = x * rc;  r = fma(-q0, W, x);  q = fma(r, rc, q0),  rc = RN(1 / W)    — the correctly rounded x / W for
 * W in {3, 6, 7, 14} (events.c:52-65) and every x the guard below lets through:
 *   - r is x - q0 W exactly (q0 is within 2^-52 of x / W relative, so the difference fits the format) and the value the last fma
 *     rounds, v = q0 + r rc, is within |r| |rc - 1/W| <= 2^-51 ulp of x / W = q0 + r / W;
 *   - x / W, in units of its own ulp, is an integer multiple of 1/3 or 1/7 (W = 2^s 3 or 2^s 7): it is representable or at least 1/14
 *     ulp away from every representable number and from every midpoint between two — so v lies on the same side of every rounding
 *     boundary as x / W and RN(v) = RN(x / W).
 * That needs q0, r and q normal.  Doubles here are exact sums of <= 14 floats: finite is enough.  Floats: all 2^32 inputs were
 * checked against x / W per divisor (tools/proto/div_const_check.c): the only mismatches are x = +-inf (the sequence makes NaN) and
 * |x / W| < 4 FLT_MIN (double rounding among subnormals) — those, NaN and the FLT_MIN clamp of a zero variance take the IEEE
 * divisions of abea_tstat_w, a branch that real signals do not enter. */
template <int W>
static __device__ __forceinline__ float abea_tstat_fast(double sum1, double sum2d, double sumsq1, double sumsq2d) {
    constexpr float wf = (float)W, rcf = 1.0f / (float)W;
    constexpr double wd = (double)W, rcd = 1.0 / (double)W;
    auto divf = [&](float x) { const float q0 = __fmul_rn(x, rcf); return __fmaf_rn(__fmaf_rn(-q0, wf, x), rcf, q0); };
    auto divd = [&](double x) { const double q0 = __dmul_rn(x, rcd); return __fma_rn(__fma_rn(-q0, wd, x), rcd, q0); };
    const float sum2 = (float)sum2d;
    const float sumsq2 = (float)sumsq2d;
    const float mean1 = (float)divd(sum1);
    const float mean2 = divf(sum2);
    float combined_var = (float)(((divd(sumsq1) - (double)(mean1 * mean1)) + (double)divf(sumsq2)) - (double)(mean2 * mean2));
    combined_var = fmaxf(combined_var, 1.17549435e-38f);            /* FLT_MIN */
    const float delta_mean = mean2 - mean1;
    float t = fabsf(delta_mean) / sqrtf(divf(combined_var));
    constexpr float tiny = 0x1p-118f, inf = __builtin_inff();
    const float a2 = fabsf(sum2), aq = fabsf(sumsq2);
    const bool ok = a2 >= tiny && a2 < inf && aq >= tiny && aq < inf && combined_var >= tiny && combined_var < inf &&
                    fabs(sum1) < (double)inf && fabs(sumsq1) < (double)inf;     /* zeros too go the IEEE way (-0 / W = -0, the sequence +0) */
    if (!ok) t = abea_tstat_w(sum1, sum2d, sumsq1, sumsq2d, wf);
    return t;
}

template <int W1, int W2>
struct abea_ev_win {
    float x[2 * W2];                      /* samples p - W2 .. p + W2 - 1 in pA (0 outside the read) when position p is evaluated */
    double sl1, sr1, sl2, sr2, ql1, qr1, ql2, qr2;     /* sums over [p-W1,p), [p,p+W1), [p-W2,p), [p,p+W2): samples / float squares */
    __device__ __forceinline__ void init() {
        sl1 = sr1 = sl2 = sr2 = ql1 = qr1 = ql2 = qr2 = 0.0;
        #pragma unroll
        for (int i = 0; i < 2 * W2; ++i) {
            const double xd = (double)x[i], yd = (double)__fmul_rn(x[i], x[i]);
            if (i < W2) { sl2 += xd; ql2 += yd; } else { sr2 += xd; qr2 += yd; }
            if (i >= W2 - W1 && i < W2) { sl1 += xd; ql1 += yd; }
            if (i >= W2 && i < W2 + W1) { sr1 += xd; qr1 += yd; }
        }
    }
    /* p -> p + 1; xin = sample p + W2 */
    __device__ __forceinline__ void advance(float xin) {
        const float xo2 = x[0], xo1 = x[W2 - W1], xm = x[W2], xi1 = x[W2 + W1];
        const double dm = (double)xm, ym = (double)__fmul_rn(xm, xm);
        sl2 = (sl2 + dm) - (double)xo2;          ql2 = (ql2 + ym) - (double)__fmul_rn(xo2, xo2);
        sr2 = (sr2 + (double)xin) - dm;          qr2 = (qr2 + (double)__fmul_rn(xin, xin)) - ym;
        sl1 = (sl1 + dm) - (double)xo1;          ql1 = (ql1 + ym) - (double)__fmul_rn(xo1, xo1);
        sr1 = (sr1 + (double)xi1) - dm;          qr1 = (qr1 + (double)__fmul_rn(xi1, xi1)) - ym;
        #pragma unroll
        for (int i = 0; i + 1 < 2 * W2; ++i) x[i] = x[i + 1];
        x[2 * W2 - 1] = xin;
    }
    /* events.c:336-347: zero where a window does not fit the read */
    __device__ __forceinline__ void tstats(int p, int n, float& a, float& b) const {
        const float ta = abea_tstat_fast<W1>(sl1, sr1, ql1, qr1);
        const float tb = abea_tstat_fast<W2>(sl2, sr2, ql2, qr2);
        a = (n >= 2 * W1 && p >= W1 && p <= n - W1) ? ta : 0.f;
        b = (n >= 2 * W2 && p >= W2 && p <= n - W2) ? tb : 0.f;
    }
};

/* BLK = samples per staged block (a lane fetches BLK / 8 16-byte chunks back to back); a lane's LDS row is a ring of two blocks + 8
 * entries of pending peaks.  The newest sample a step needs is at most 2 W2 + 7 + (BLK - 1) samples past the start of the block the step
 * began in, which must stay below 2 BLK: RNA (W2 = 14) takes 96 bytes at a time (BLK = 48, 208-byte rows, 12 wavefronts per CU =
 * what its 142 VGPRs allow), DNA (W2 = 6) half lines (BLK = 32, 144-byte rows, 17 per CU) — the loop is a long dependent chain
 * per lane (two t-statistics, an automaton step) and lives on occupancy. */
template <int W1, int W2, int BLK>
static __device__ __forceinline__ void spec2_body(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                          const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                          const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                          const int32_t* __restrict__ wave_nseg, uint16_t* __restrict__ spec_all,
                          int32_t* __restrict__ segrec_all, uint32_t* __restrict__ segexp_all, const abea_ev_par P,
                          uint16_t* __restrict__ rows) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x;
    const int j = blockIdx.x;
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    const int seg_lo = j * ABEA_EV_SEG;
    const int lo = max(seg_lo, 1);                                   /* masked_to starts at 0: position 0 is skipped */
    const int hi = min(seg_lo + ABEA_EV_SEG, n);
    if (lo >= hi && j > 0) return;
    const int16_t* __restrict__ sig = signal + sig_ptr[r];
    const float offset = scaling[3 * r], raw_unit = scaling[3 * r + 1] / scaling[3 * r + 2];   /* f5c.c:693 */
    const int64_t seg = seg_base[w] + j;
    /* the speculative run's peaks: ONE list per (read, segment), contiguous (the array form interleaves the 64 reads of a wave: fine
     * for a lane-per-read reader, a 128-byte stride for abea_ev_create3_kernel's lane-per-event one).  A lane collects 8 entries in
     * its LDS row and stores them as one 16-byte chunk.  Measured, 2048 reads, spec2 + create3 ms and FETCH x 2 + WRITE of the two
     * per sample: interleaved 2.35 + 0.93, 18.4 B; contiguous with 2-byte stores 2.44 + 0.78, 20.0 B (each store dirties a line
     * that is evicted long before it fills); contiguous in 16-byte chunks 2.52 + 0.73, 15.0 B — the same time, the least traffic */
    uint16_t* __restrict__ out = spec_all + (seg * 64 + lane) * ABEA_EV_SEG;
    int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + lane;
    uint32_t* __restrict__ se = segexp_all + seg * 4 * 64 + lane;
    static_assert(2 * W2 + 7 + BLK - 1 < 2 * BLK, "the ring must hold the newest sample of every step");
    constexpr int ROW = 2 * BLK + 8, RINGN = 2 * BLK;                /* ring positions: index modulo RINGN (a mask where BLK is a power of two) */
    uint16_t* __restrict__ my = rows + lane * ROW;

    /* the row is a ring of two BLK-sample blocks; block b holds the samples i0 + BLK b .. i0 + BLK b + BLK - 1, i0 = the sample at the
     * 16-byte boundary at or below the first one needed */
    const int q0 = lo - W2;
    const int sh = (int)(((uintptr_t)sig + (uintptr_t)(2 * (int64_t)q0)) & 15u) >> 1;
    const int i0 = q0 - sh;
    auto load_block = [&](int b) {
        const int b0 = i0 + BLK * b;
        if (b0 >= n || b0 + BLK <= 0) return;                        /* nothing of the read in it */
        uint4 c[BLK / 8];
        if (b0 >= 0 && b0 + BLK <= n) {
            #pragma unroll
            for (int u = 0; u < BLK / 8; ++u) c[u] = *reinterpret_cast<const uint4*>(sig + b0 + 8 * u);
        } else {                                                     /* a block across an end of the read: never a byte outside it */
            #pragma unroll
            for (int u = 0; u < BLK / 8; ++u) {
                uint32_t h[8];
                #pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const int i = b0 + 8 * u + v;
                    h[v] = (i >= 0 && i < n) ? (uint32_t)(uint16_t)sig[i] : 0u;
                }
                c[u] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            }
        }
        uint2* __restrict__ d = reinterpret_cast<uint2*>(my + (b & 1) * BLK);
        #pragma unroll
        for (int u = 0; u < BLK / 8; ++u) { d[2 * u] = make_uint2(c[u].x, c[u].y); d[2 * u + 1] = make_uint2(c[u].z, c[u].w); }
    };
    uint32_t xmin = 0x7f800000u, xmax = 0u, ymin = 0x7f800000u, ymax = 0u;
    auto fetch = [&](int i) -> float {
        const int raw = (int)(short)my[(unsigned)(i - i0) % (unsigned)RINGN];
        const float x = (i >= 0 && i < n) ? abea_pa(raw, offset, raw_unit) : 0.f;
        if (i >= seg_lo && i < hi) {                                 /* this segment's own samples: the exactness test's input */
            const float y = __fmul_rn(x, x);
            const uint32_t ax = __float_as_uint(x) & 0x7fffffffu, ay = __float_as_uint(y);
            xmax = max(xmax, ax); ymax = max(ymax, ay);
            xmin = min(xmin, ax ? ax : 0x7f800000u); ymin = min(ymin, ay ? ay : 0x7f800000u);
        }
        return x;
    };
    load_block(0); load_block(1);
    abea_ev_win<W1, W2> wn;
    #pragma unroll
    for (int i = 0; i < 2 * W2; ++i) wn.x[i] = fetch(q0 + i);
    wn.init();
    abea_det2 s; det2_reset(s);
    int cnt = 0;
    uint16_t* __restrict__ pend = my + 2 * BLK;                      /* 8 pending entries behind the sample ring */
    auto flush = [&](int at) {                                       /* entries at .. at + 7 of the list (16-byte aligned) */
        const uint2 lo2 = reinterpret_cast<const uint2*>(pend)[0], hi2 = reinterpret_cast<const uint2*>(pend)[1];
        *reinterpret_cast<uint4*>(out + at) = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
    };
    auto emit = [&](int pos) {
        pend[cnt & 7] = (uint16_t)(pos - seg_lo);
        ++cnt;
        if ((cnt & 7) == 0) flush(cnt - 8);
    };
    const int steps = hi - lo;
    for (int t = 0; t < steps; ++t) {
        if (t != 0 && t % BLK == 0) load_block(t / BLK + 1);        /* samples up to i0 + t + 2 W2 + 7 < BLK (t / BLK + 2) */
        const int p = lo + t;
        float a, b;
        wn.tstats(p, n, a, b);
        int f0, f1;
        const int fired = det2_step(s, p, a, b, f0, f1, P);
        if (fired & 1) emit(f0);
        if (fired & 2) emit(f1);
        wn.advance(fetch(p + W2));
    }
    if (cnt & 7) flush(cnt & ~7);                                    /* the last, partial chunk (its tail is never read) */
    rec[0 * 64] = cnt;                                               /* as abea_ev_spec_kernel */
    rec[1 * 64] = 0;
    rec[2 * 64] = 0;
    rec[4 * 64] = __float_as_int(s.pv0); rec[5 * 64] = __float_as_int(s.pv1);
    rec[6 * 64] = s.pp0; rec[7 * 64] = s.pp1; rec[8 * 64] = s.v0; rec[9 * 64] = s.v1; rec[10 * 64] = s.masked;
    se[0] = xmin; se[64] = xmax; se[128] = ymin; se[192] = ymax;
}

/* one kernel per parameter set: a shared one is allocated the registers and the LDS of the larger (RNA) body */
extern "C" __global__ __launch_bounds__(64)
void abea_ev_spec2_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                          const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                          const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                          const int32_t* __restrict__ wave_nseg, uint16_t* __restrict__ spec_all,
                          int32_t* __restrict__ segrec_all, uint32_t* __restrict__ segexp_all) {
    __shared__ __attribute__((aligned(16))) uint16_t rows[64 * (2 * 32 + 8)];
    spec2_body<3, 6, 32>(n_reads, order, signal, sig_ptr, n_samples, scaling, seg_base, wave_nseg, spec_all, segrec_all, segexp_all, ev_par(0), rows);
}
extern "C" __global__ __launch_bounds__(64)
void abea_ev_spec2_rna_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                              const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                              const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                              const int32_t* __restrict__ wave_nseg, uint16_t* __restrict__ spec_all,
                              int32_t* __restrict__ segrec_all, uint32_t* __restrict__ segexp_all) {
    __shared__ __attribute__((aligned(16))) uint16_t rows[64 * (2 * 48 + 8)];
    spec2_body<7, 14, 48>(n_reads, order, signal, sig_ptr, n_samples, scaling, seg_base, wave_nseg, spec_all, segrec_all, segexp_all, ev_par(1), rows);
}

template <int W1, int W2>
static __device__ __forceinline__ void fix2_body(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                         const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                         const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                         const int32_t* __restrict__ wave_nseg, int32_t* __restrict__ fix_all,
                         int32_t* __restrict__ segrec_all, int32_t* __restrict__ need_seq, const abea_ev_par P, int window) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;           /* segment 0 starts from the true state already */
    const int slot = w * 64 + lane;
    if (j >= wave_nseg[w] || slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    const int lo = j * ABEA_EV_SEG;
    const int hi = min(lo + ABEA_EV_SEG, n);
    if (lo >= hi) return;
    const int16_t* __restrict__ sig = signal + sig_ptr[r];
    const float offset = scaling[3 * r], raw_unit = scaling[3 * r + 1] / scaling[3 * r + 2];
    const int64_t seg = seg_base[w] + j;
    int32_t* __restrict__ out = fix_all + seg * ABEA_EV_FIXCAP * 64 + lane;
    int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + lane;
    const int32_t* __restrict__ prev = rec - 12 * 64;
    abea_det2 tr, sp;
    tr.pv0 = __int_as_float(prev[4 * 64]); tr.pv1 = __int_as_float(prev[5 * 64]);
    tr.pp0 = prev[6 * 64]; tr.pp1 = prev[7 * 64]; tr.v0 = prev[8 * 64]; tr.v1 = prev[9 * 64]; tr.masked = prev[10 * 64];
    det2_reset(sp);
    int nfix = 0, skip = 0, p = lo;
    bool met = det2_equal(tr, sp, lo - 1);
    const int stop = min(hi, lo + window);
    if (!met) {
        auto fetch = [&](int i) -> float { return (i >= 0 && i < n) ? abea_pa((int)sig[i], offset, raw_unit) : 0.f; };
        abea_ev_win<W1, W2> wn;
        #pragma unroll
        for (int i = 0; i < 2 * W2; ++i) wn.x[i] = fetch(lo - W2 + i);
        wn.init();
        while (!met && p < stop) {
            float c0, c1;
            wn.tstats(p, n, c0, c1);
            int f0, f1, g0, g1;
            const int ft = det2_step(tr, p, c0, c1, f0, f1, P);
            const int fs = det2_step(sp, p, c0, c1, g0, g1, P);
            if (ft & 1) { if (nfix < ABEA_EV_FIXCAP) out[(size_t)nfix * 64] = f0; ++nfix; }
            if (ft & 2) { if (nfix < ABEA_EV_FIXCAP) out[(size_t)nfix * 64] = f1; ++nfix; }
            skip += (fs & 1) + ((fs >> 1) & 1);
            met = det2_equal(tr, sp, p);
            wn.advance(fetch(p + W2));
            ++p;
        }
    }
    /* as abea_ev_fix_kernel */
    if ((!met && p < hi) || nfix > ABEA_EV_FIXCAP) need_seq[r] = 1;
    if (!met && p >= hi) skip = rec[0 * 64];
    rec[1 * 64] = nfix;
    rec[2 * 64] = skip;
}

extern "C" __global__ __launch_bounds__(256)
void abea_ev_fix2_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                         const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                         const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                         const int32_t* __restrict__ wave_nseg, int32_t* __restrict__ fix_all,
                         int32_t* __restrict__ segrec_all, int32_t* __restrict__ need_seq, int rna) {
    const abea_ev_par P = ev_par(rna);
    if (rna) fix2_body<7, 14>(n_reads, order, signal, sig_ptr, n_samples, scaling, seg_base, wave_nseg, fix_all, segrec_all, need_seq, P, ABEA_EV_FIX_RNA);
    else fix2_body<3, 6>(n_reads, order, signal, sig_ptr, n_samples, scaling, seg_base, wave_nseg, fix_all, segrec_all, need_seq, P, ABEA_EV_FIX);
}

/* The per-read part of the common path, one wavefront per READ (the lane-per-read abea_ev_scan_kernel walks a read's ~440 segment
 * records one dependent load after the other: 0.57 ms per 2048 reads).  64 segments at a time: peak counts -> exclusive running sum
 * (rec[3], where the segment's events start), the last peak of the nearest non-empty segment before it (rec[11]: where the segment's
 * first event starts; the lists are walked in list order, never sorted), the exponent ranges abea_ev_spec2_kernel recorded -> the
 * exactness test of abea_ev_pscan_kernel (a read that fails it is flagged for the array kernels), and n_events. */
extern "C" __global__ __launch_bounds__(64)
void abea_ev_scan2_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                          const int64_t* __restrict__ seg_base, int32_t* __restrict__ segrec_all,
                          const uint16_t* __restrict__ spec_all, const int32_t* __restrict__ fix_all,
                          const uint32_t* __restrict__ segexp_all, int32_t* __restrict__ n_events, int32_t* __restrict__ need) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_reads) return;
    const int r = order[slot];
    const int n = n_samples[r];
    if (n <= 0) { if (lane == 0) n_events[r] = 0; return; }
    const int nseg = (n + ABEA_EV_SEG - 1) / ABEA_EV_SEG;
    const int w = slot >> 6, l = slot & 63;
    int run = 0, prev = 0;                                           /* carried over the chunks of 64 segments (wave-uniform) */
    uint32_t xmin = 0x7f800000u, xmax = 0u, ymin = 0x7f800000u, ymax = 0u;
    for (int c0 = 0; c0 < nseg; c0 += 64) {
        const int j = c0 + lane;
        const bool live = j < nseg;
        const int64_t seg = seg_base[w] + (live ? j : c0);
        int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + l;
        const int nspec = live ? rec[0] : 0, nfix_raw = live ? rec[64] : 0, skip = live ? rec[128] : 0;
        const int cnt = nspec + nfix_raw - skip;                     /* as abea_ev_scan_kernel */
        const int nfix = min(nfix_raw, ABEA_EV_FIXCAP);              /* more than the cap: the read is flagged, only the addressing must hold */
        int last = 0;
        if (live && nspec - skip > 0) last = j * ABEA_EV_SEG + (int)spec_all[(seg * 64 + l) * ABEA_EV_SEG + (nspec - 1)];
        else if (live && nfix > 0) last = fix_all[seg * ABEA_EV_FIXCAP * 64 + (int64_t)(nfix - 1) * 64 + l];
        if (live) {
            const uint32_t* __restrict__ se = segexp_all + seg * 4 * 64 + l;
            xmin = min(xmin, se[0]); xmax = max(xmax, se[64]); ymin = min(ymin, se[128]); ymax = max(ymax, se[192]);
        }
        /* inclusive scans over the 64 lanes: the running count, and the nearest lane at or below that has a peak */
        int inc = cnt, src = cnt > 0 ? lane : -1;
        #pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int a = __shfl_up(inc, off, 64), b = __shfl_up(src, off, 64);
            if (lane >= off) { inc += a; src = max(src, b); }
        }
        const int src_ex = __shfl_up(src, 1, 64);                    /* exclusive: nearest lane BELOW with a peak */
        const int from = lane > 0 ? src_ex : -1;
        const int got = __shfl(last, max(from, 0), 64);
        if (live) {
            rec[3 * 64] = run + inc - cnt;
            rec[11 * 64] = from >= 0 ? got : prev;
        }
        const int src_all = __shfl(src, 63, 64);
        if (src_all >= 0) prev = __shfl(last, src_all, 64);
        run += __shfl(inc, 63, 64);
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        xmin = min(xmin, (uint32_t)__shfl_xor((int)xmin, off, 64)); xmax = max(xmax, (uint32_t)__shfl_xor((int)xmax, off, 64));
        ymin = min(ymin, (uint32_t)__shfl_xor((int)ymin, off, 64)); ymax = max(ymax, (uint32_t)__shfl_xor((int)ymax, off, 64));
    }
    if (lane != 0) return;
    n_events[r] = run + 1;                                           /* events.c:491-497: one more event than peaks */
    if (!(abea_ev_sum_exact(xmin, xmax, n) && abea_ev_sum_exact(ymin, ymax, n))) need[r] = 1;
}

/* Events of the unflagged reads (events.c:466-513) straight from the segments' peak lists and the samples: a wavefront per (read,
 * group of ABEA_EV_CGRP segments), a lane per event.  An event ends at one peak of the segment (replay peaks first, then the
 * speculative run's from `skip` on — what abea_ev_gather_kernel copies into a per-read list in the array form; that list no longer
 * exists here) and starts at the peak before it, the neighbouring lane's or rec[11] for the segment's first; the read's last event
 * ends at n.  The two sums are added up from the event's own samples (exact, see above: the reference's sums[end] - sums[start]).
 * grid.x = reads: the 64 reads whose lists share cache lines run side by side. */
#define ABEA_EV_CGRP 8
extern "C" __global__ __launch_bounds__(64)
void abea_ev_create3_kernel(int n_reads, const int32_t* __restrict__ order, const int16_t* __restrict__ signal,
                            const int64_t* __restrict__ sig_ptr, const int32_t* __restrict__ n_samples,
                            const float* __restrict__ scaling, const int64_t* __restrict__ seg_base,
                            const int32_t* __restrict__ segrec_all, const uint16_t* __restrict__ spec_all,
                            const int32_t* __restrict__ fix_all, const int64_t* __restrict__ peak_base,
                            const int32_t* __restrict__ wave_cap, const int32_t* __restrict__ n_events,
                            const int32_t* __restrict__ event_cap, abea_event_t* __restrict__ events,
                            const int64_t* __restrict__ event_ptr, float* __restrict__ mean_all,
                            const int32_t* __restrict__ need, int rna) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_reads) return;
    const int r = order[slot];
    if (need[r]) return;
    const int n = n_samples[r];
    const int n_ev = n_events[r], cap = event_cap[r], ne = min(n_ev, cap);
    if (n <= 0 || ne <= 0) return;
    const int nseg = (n + ABEA_EV_SEG - 1) / ABEA_EV_SEG;
    const int w = slot >> 6, l = slot & 63;
    float* __restrict__ mean = mean_all + peak_base[w] + (int64_t)l * wave_cap[w];   /* detection order: the scalings kernel's input */
    const int16_t* __restrict__ sig = signal + sig_ptr[r];
    const float offset = scaling[3 * r], raw_unit = scaling[3 * r + 1] / scaling[3 * r + 2];
    abea_event_t* __restrict__ ev = events + event_ptr[r];
    for (int j = blockIdx.y * ABEA_EV_CGRP; j < min(nseg, (int)(blockIdx.y + 1) * ABEA_EV_CGRP); ++j) {
        const int64_t seg = seg_base[w] + j;
        const int32_t* __restrict__ rec = segrec_all + seg * 12 * 64 + l;
        const uint16_t* __restrict__ sp = spec_all + (seg * 64 + l) * ABEA_EV_SEG;        /* abea_ev_spec2_kernel's list of this (read, segment) */
        const int32_t* __restrict__ fx = fix_all + seg * ABEA_EV_FIXCAP * 64 + l;
        const int nspec = rec[0], nfix = min(rec[64], ABEA_EV_FIXCAP), skip = rec[128], run = rec[3 * 64];
        int before = rec[11 * 64];                                   /* the last peak before this segment (0: none) */
        const int c = nfix + nspec - skip;
        const int todo = c + (j == nseg - 1 ? 1 : 0);                /* the read's last event ends at n */
        for (int i0 = 0; i0 < todo; i0 += 64) {
            const int i = i0 + lane;
            int end = n;
            if (i < nfix) end = fx[(size_t)i * 64];
            else if (i < c) end = j * ABEA_EV_SEG + (int)sp[skip + i - nfix];
            const int up = __shfl_up(end, 1, 64);
            const int start = lane ? up : before;
            before = __shfl(end, 63, 64);
            const int e = run + i;
            if (i >= todo || e >= ne) continue;
            /* sums[end] - sums[start]: the samples of [start, end); a list that ran backwards would give the negated sum of [end, start) */
            const int a = min(start, end), b = max(start, end);
            double S = 0.0, Q = 0.0;
            for (int q = a; q < b; q += 8) {                          /* 8 samples per load (any alignment): one memory latency per 8, not per sample */
                uint32_t v[4];
                if (q + 8 <= n) {
                    __builtin_memcpy(v, sig + q, 16);
                } else {
                    v[0] = v[1] = v[2] = v[3] = 0u;
                    #pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (q + u < n) v[u >> 1] |= (uint32_t)(uint16_t)sig[q + u] << (16 * (u & 1));
                }
                #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (q + u < b) {
                        const float x = abea_pa((int)(short)(v[u >> 1] >> (16 * (u & 1))), offset, raw_unit);
                        S += (double)x; Q += (double)__fmul_rn(x, x);
                    }
                }
            }
            if (end < start) { S = 0.0 - S; Q = 0.0 - Q; }
            abea_event_t o;                                          /* events.c:497-513 */
            o.start = (unsigned long long)start;
            o.length = (float)((unsigned long long)end - (unsigned long long)start);
            o.mean = (float)S / o.length;
            const float deltasqr = (float)Q;
            const float var = deltasqr / o.length - o.mean * o.mean;
            o.stdv = sqrtf(fmaxf(var, 0.0f));
            const int at = rna ? n_ev - 1 - e : e;                   /* RNA tables go out 3'->5' (f5c.c:711-719), the means stay in order */
            if (at < cap) ev[at] = o;
            mean[e] = o.mean;
        }
    }
}

/* pass 4: events from consecutive peaks (events.c:466-513), fully parallel: one thread per (read, event) */
extern "C" __global__ __launch_bounds__(256)
void abea_ev_create_kernel(int n_reads, const int32_t* __restrict__ order, const int32_t* __restrict__ n_samples,
                           const int64_t* __restrict__ wave_base, const double* __restrict__ S_all,
                           const double* __restrict__ Q_all, const int64_t* __restrict__ peak_base,
                           const int32_t* __restrict__ wave_cap, const int32_t* __restrict__ peaks_all,
                           const int32_t* __restrict__ n_events, const int32_t* __restrict__ event_cap,
                           abea_event_t* __restrict__ events, const int64_t* __restrict__ event_ptr,
                           float* __restrict__ mean_all, int rna, const int32_t* __restrict__ need) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int slot = w * 64 + lane;
    if (slot >= n_reads) return;
    const int r = order[slot];
    if (need && !need[r]) return;                                    /* behind the fused common path: flagged reads only */
    const int n = n_samples[r];
    const int ne = min(n_events[r], event_cap[r]);
    const int64_t base = wave_base[w] + lane;
    const int32_t* __restrict__ pk = peaks_all + peak_base[w] + lane;
    abea_event_t* __restrict__ ev = events + event_ptr[r];
    const int wcap = wave_cap[w];
    /* a thread makes a run of 8 consecutive events of its read: 9 boundary look-ups instead of 16, and 192 contiguous
     * bytes of event_t */
    const int n_ev = n_events[r];
    for (int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; j0 < wcap; j0 += gridDim.x * 32) {
        if (j0 >= ne || n <= 0) continue;
        unsigned long long b[9];
        double sb[9], qb[9];
        #pragma unroll
        for (int i = 0; i < 9; ++i) {                                /* b[i] = start of event j0+i = end of event j0+i-1 */
            const int j = j0 + i - 1;
            /* boundaries past the last event this thread may write (j >= ne: a table cut off at its capacity) are never used — and
             * their peak slots were never written: reading them would index the prefix sums with stale memory (round 5: a table
             * overflowing 9x faulted here) */
            b[i] = (j < 0) ? 0ull : (j >= n_ev - 1 || j >= ne) ? (unsigned long long)n : (unsigned long long)pk[(size_t)min(j, wcap - 1) * 64];
        }
        #pragma unroll
        for (int i = 0; i < 9; ++i) {                                /* one 16-byte look-up per boundary: {S, Q} side by side */
            const double2 v = reinterpret_cast<const double2*>(S_all)[base + (int64_t)b[i] * 64];
            sb[i] = v.x; qb[i] = v.y;
        }
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = j0 + i;
            if (j >= ne) break;
            const unsigned long long start = b[i], end = b[i + 1];
            abea_event_t e;                                          /* events.c:497-513 */
            e.start = start;
            e.length = (float)(end - start);
            e.mean = (float)(sb[i + 1] - sb[i]) / e.length;
            const float deltasqr = (float)(qb[i + 1] - qb[i]);
            const float var = deltasqr / e.length - e.mean * e.mean;
            e.stdv = sqrtf(fmaxf(var, 0.0f));
            /* RNA: event_single() reverses the table to 3'->5' AFTER the scalings are estimated (f5c.c:711-719): the
             * table goes out reversed, mean_all (what the scalings kernel sums, in order) stays in detection order.  A
             * truncated RNA table (n_ev > cap, reported to the caller through n_events) is only partly written and not
             * usable: the caller redoes the read with a larger table (include/abea.h; abea_events_batch_host does). */
            const int at = rna ? n_ev - 1 - j : j;
            if (at < event_cap[r]) ev[at] = e;
            mean_all[peak_base[w] + (int64_t)lane * wcap + j] = e.mean;   /* linear per read, detection order: the scalings kernel's input */
        }
    }
}

/* event tables of a chunk, each in its own slot range of `src`, copied back to back into `dst` (block = read): what
 * abea_events_batch_host sends down over PCIe (event slots are sized from the sample count, ~2x the events found) */
extern "C" __global__ __launch_bounds__(256)
void abea_ev_compact_kernel(int n_reads, const abea_event_t* __restrict__ src, const int64_t* __restrict__ src_ptr,
                            const int64_t* __restrict__ dst_ptr, const int32_t* __restrict__ n_events,
                            abea_event_t* __restrict__ dst) {
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const unsigned long long* __restrict__ s = reinterpret_cast<const unsigned long long*>(src + src_ptr[r]);
    unsigned long long* __restrict__ d = reinterpret_cast<unsigned long long*>(dst + dst_ptr[r]);
    const int64_t words = (int64_t)max(n_events[r], 0) * 3;          /* event_t = 24 bytes */
    for (int64_t i = threadIdx.x; i < words; i += blockDim.x) d[i] = s[i];
}

/* The same compaction into the 12-byte form the tables cross PCIe in (round 6): {uint32 start, float mean, float stdv}.  The events
 * of a read tile its samples (events.c:466-513: start(j+1) = start(j) + the integer length(j), start(0) = 0, the last one ends at
 * n_samples), so `start` of the neighbour and n_samples give every (uint64 start, float length) back on the host — half the bytes
 * of event_t over the link.  Block = read, a thread packs 4 consecutive events into three 16-byte stores; dst_ptr counts records and
 * is a multiple of 4 per read. */
extern "C" __global__ __launch_bounds__(256)
void abea_ev_pack_kernel(int n_reads, const abea_event_t* __restrict__ src, const int64_t* __restrict__ src_ptr,
                         const int64_t* __restrict__ dst_ptr, const int32_t* __restrict__ n_events, uint4* __restrict__ dst) {
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const abea_event_t* __restrict__ s = src + src_ptr[r];
    uint4* __restrict__ d = dst + dst_ptr[r] / 4 * 3;
    const int ne = max(n_events[r], 0);
    for (int g = threadIdx.x; g * 4 < ne; g += blockDim.x) {
        uint32_t w[12];
        #pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = g * 4 + i;
            abea_event_t e;
            if (j < ne) e = s[j]; else { e.start = 0; e.length = 0.f; e.mean = 0.f; e.stdv = 0.f; }
            w[3 * i] = (uint32_t)e.start; w[3 * i + 1] = __float_as_uint(e.mean); w[3 * i + 2] = __float_as_uint(e.stdv);
        }
        d[(size_t)g * 3] = make_uint4(w[0], w[1], w[2], w[3]);
        d[(size_t)g * 3 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        d[(size_t)g * 3 + 2] = make_uint4(w[8], w[9], w[10], w[11]);
    }
}

/* kmer_rank_at in two halves, for callers that want the load in flight before they need the rank: the 12 bytes at seq[i] (assembled
 * base by base, zero-filled, where the window would pass the read's terminator), and the rank from those words */
static __device__ __forceinline__ void kmer_window_at(const char* __restrict__ seq, int i, int L, int kmer_size, uint32_t w[3]) {
    if (i + 12 <= L + 1) {
        __builtin_memcpy(w, seq + i, 12);
    } else {
        w[0] = w[1] = w[2] = 0u;
        #pragma unroll                                               /* static indices: a dynamically indexed w[] is moved to LDS by the compiler */
        for (int j = 0; j < ABEA_MAX_KMER_SIZE; ++j)
            if (j < kmer_size) w[j >> 2] |= (uint32_t)(unsigned char)seq[i + j] << (8 * (j & 3));
    }
}
static __device__ __forceinline__ uint32_t kmer_rank_of_window(const uint32_t w[3], int kmer_size) {
    uint32_t rank = 0;
    #pragma unroll
    for (int j = 0; j < ABEA_MAX_KMER_SIZE; ++j)
        if (j < kmer_size) rank = (rank << 2) | base_code((w[j >> 2] >> (8 * (j & 3))) & 0xFFu);
    return rank;
}

/* pass 5: estimate_scalings_using_mom (align.c:58-106), one wavefront per READ (round 6).
 * Four sums per read: Σ event mean, Σ level, Σ level², then Σ (mean − shift)², all accumulated in double in index order.
 * Rounds 1-5 gave a read one LANE (the detector's layout) and 64 reads a wavefront: the kernel was as long as the longest read's
 * ~10^5 dependent additions times the LDS feed of 64 chains — 6.7 ms per 2048 reads alone, 12.6 ms per chunk inside the pipeline,
 * more than the rest of the detector together (profiles/r06/chain_kernel_stats_before.csv), on 32 busy wavefronts of a 1024-SIMD
 * chip.  Now:
 *   - Σ mean and Σ level are sums of FLOATS in double.  All terms are multiples of the smallest term's ulp 2^(emin-150) and every
 *     partial sum of any subset is below n * 2^(emax-126): when ceil(log2 n) + emax - emin <= 29 no addition of the sum can round,
 *     in ANY order, so 64 lanes add strided subsets and a butterfly finishes — the sequential sum bit for bit (the test the
 *     prefix sums of pass 1 use).  Checked per read and per sum; a sum that fails (a mean of 1e-9 next to means of 100) takes the
 *     sequential form below.
 *   - Σ level² and Σ (mean − shift)² have full-mantissa terms: no re-association is exact, the chains stay sequential — but their
 *     TERMS are computed 64 wide (k-mer rank, model gather, the fp64 product) and parked in LDS, and two lanes add one column each
 *     in order, both chains at once, 64 additions per tile fed by 16-byte LDS reads.  A 100 k-event read is ~0.5 ms of dependent
 *     v_add_f64 on its own wavefront, and thousands of reads run side by side.
 * Event means arrive in detection order from the create kernel's linear per-read array (RNA tables are reversed only afterwards,
 * f5c.c:707-719); the k-mer levels are derived here from the sequence (no intermediate array). */
extern "C" __global__ __launch_bounds__(64)
void abea_ev_scalings_kernel(int n_reads, const int32_t* __restrict__ order, const int64_t* __restrict__ peak_base,
                             const int32_t* __restrict__ wave_cap, const float* __restrict__ mean_lin,
                             const int32_t* __restrict__ n_events, const int32_t* __restrict__ event_cap,
                             const char* __restrict__ reads, const int64_t* __restrict__ read_ptr,
                             const int32_t* __restrict__ read_len, const abea_model_t* __restrict__ model, int kmer_size,
                             abea_scalings_t* __restrict__ scalings, float* __restrict__ level_all,
                             const int64_t* __restrict__ kmer_base, const int32_t* __restrict__ wave_k) {
    __shared__ __attribute__((aligned(16))) double lds[2][2][64];       /* [buffer][chain][term] */
    const int lane = threadIdx.x, slot = blockIdx.x;
    if (slot >= n_reads) return;
    const int r = order[slot];
    const int n_ev = n_events[r], ne = min(n_ev, event_cap[r]);
    const int L = read_len[r], K = L - kmer_size + 1;
    const float* __restrict__ mean = mean_lin + peak_base[slot >> 6] + (int64_t)(slot & 63) * wave_cap[slot >> 6];
    const char* __restrict__ seq = reads + read_ptr[r];
    auto level = [&](int i) { return model[kmer_rank_at(seq, i, L, kmer_size)].level_mean; };
    /* the k-mer levels are derived ONCE, by the strided pass below, and parked in the read's own scratch row: the chains' tiles
     * then cost a float load per term instead of a rank (50 instructions) and a dependent gather — a lone wavefront issues one
     * instruction every ~5 cycles, and what the 62 idle lanes execute between two tiles is time the two chain lanes wait */
    float* __restrict__ lvl = level_all + kmer_base[slot >> 6] + (int64_t)(slot & 63) * wave_k[slot >> 6];
    /* ---- the two float sums: strided partial sums + the exponent range of the terms ---- */
    double ps_e = 0.0, ps_k = 0.0;
    int lo_e = 255, hi_e = 0, lo_k = 255, hi_k = 0;
    auto range = [](float x, int& lo, int& hi) {
        const int e = (int)((__float_as_uint(x) >> 23) & 0xFFu);
        if (x != 0.0f) { lo = min(lo, max(e, 1)); hi = max(hi, e); }      /* NaN / Inf: e = 255 -> never "exact" */
    };
    {   /* a lone wavefront per read is latency-bound, not bandwidth-bound: 32 means per lane and trip in flight (two 16-byte loads of
         * any alignment), four k-mer windows and then their four model gathers */
        int i = 0;
        for (; i + 512 <= ne; i += 512) {
            float m[8];
            __builtin_memcpy(m, mean + i + 4 * lane, 16);
            __builtin_memcpy(m + 4, mean + i + 256 + 4 * lane, 16);
            #pragma unroll
            for (int u = 0; u < 8; ++u) { ps_e += (double)m[u]; range(m[u], lo_e, hi_e); }
        }
        for (i += lane; i < ne; i += 64) { const float m = mean[i]; ps_e += (double)m; range(m, lo_e, hi_e); }
        i = lane;
        for (; i + 192 < K; i += 256) {
            uint32_t w0[3], w1[3], w2[3], w3[3];
            kmer_window_at(seq, i, L, kmer_size, w0); kmer_window_at(seq, i + 64, L, kmer_size, w1);
            kmer_window_at(seq, i + 128, L, kmer_size, w2); kmer_window_at(seq, i + 192, L, kmer_size, w3);
            const float l0 = model[kmer_rank_of_window(w0, kmer_size)].level_mean, l1 = model[kmer_rank_of_window(w1, kmer_size)].level_mean;
            const float l2 = model[kmer_rank_of_window(w2, kmer_size)].level_mean, l3 = model[kmer_rank_of_window(w3, kmer_size)].level_mean;
            ps_k += (double)l0; ps_k += (double)l1; ps_k += (double)l2; ps_k += (double)l3;
            range(l0, lo_k, hi_k); range(l1, lo_k, hi_k); range(l2, lo_k, hi_k); range(l3, lo_k, hi_k);
            lvl[i] = l0; lvl[i + 64] = l1; lvl[i + 128] = l2; lvl[i + 192] = l3;
        }
        for (; i < K; i += 64) { const float l = level(i); ps_k += (double)l; range(l, lo_k, hi_k); lvl[i] = l; }
    }
    __syncthreads();                                                    /* the level stores are complete before the tiles read them back */
    for (int off = 32; off > 0; off >>= 1) {
        ps_e += __shfl_xor(ps_e, off, 64); ps_k += __shfl_xor(ps_k, off, 64);
        lo_e = min(lo_e, __shfl_xor(lo_e, off, 64)); hi_e = max(hi_e, __shfl_xor(hi_e, off, 64));
        lo_k = min(lo_k, __shfl_xor(lo_k, off, 64)); hi_k = max(hi_k, __shfl_xor(hi_k, off, 64));
    }
    auto clog2 = [](int n) { return n > 1 ? 32 - __clz(n - 1) : 0; };
    const bool exact_e = hi_e < 255 && clog2(ne) + hi_e - lo_e <= 29;        /* no term at all: lo = 255, hi = 0: exact, sum = +0.0 */
    const bool exact_k = hi_k < 255 && clog2(K) + hi_k - lo_k <= 29;
    /* ---- sequential chains, two at a time: lane 0 adds column 0, lane 1 column 1, 64 terms per tile.  Past a chain's own end the
     *      term is +0.0 (x + (+0.0) == x bit for bit; the sums start at +0.0 and cannot become -0.0). ---- */
    enum { NONE, KM_SUM, KM_SQ, EV_SUM, EV_SQ };
    double shift = 0.0;
    auto run = [&](int kind0, int kind1, double& out0, double& out1) {
        auto count = [&](int kind) { return kind == NONE ? 0 : (kind == KM_SUM || kind == KM_SQ) ? K : ne; };
        /* a term in stages, its load two tiles ahead of its addition, so that no memory latency sits between two tiles of the chains:
         *   A  the raw input: the k-mer's level (parked above) or the event mean              (a load)
         *   B  (was: rank -> model gather, before the levels were parked)
         *   C  the double that is added                                                        (arithmetic) */
        struct raw { float m; };
        auto stage_a = [&](int kind, int i, raw& R) {
            R.m = 0.f;
            if (i >= count(kind)) return;
            R.m = (kind == KM_SUM || kind == KM_SQ) ? lvl[i] : mean[i];
        };
        auto stage_b = [&](int kind, int i, const raw& R) -> float { return i < count(kind) ? R.m : 0.f; };
        auto stage_c = [&](int kind, int i, float v) -> double {
            if (i >= count(kind)) return 0.0;
            if (kind == KM_SUM || kind == EV_SUM) return (double)v;                                       /* align.c:70 */
            if (kind == KM_SQ) { const double lv = (double)v; return lv * lv; }                           /* align.c:80-81 */
            const double d = (double)v - shift; return d * d;                                             /* align.c:91-92 */
        };
        const int n_it = max(count(kind0), count(kind1));
        double acc = 0.0;
        /* tile t is added while B of tile t + 1 and A of tile t + 2 are in flight.  The block is ONE wavefront: its LDS instructions execute in
         * order, so the stores of a tile are visible to the chain lanes without a workgroup barrier — and __syncthreads() would also wait
         * for every outstanding global load, i.e. for the very stages that are meant to stay in flight across tiles */
        raw a0, a1;
        stage_a(kind0, lane, a0); stage_a(kind1, lane, a1);
        float b0 = stage_b(kind0, lane, a0), b1 = stage_b(kind1, lane, a1);
        stage_a(kind0, 64 + lane, a0); stage_a(kind1, 64 + lane, a1);
        double t0 = stage_c(kind0, lane, b0), t1 = stage_c(kind1, lane, b1);
        b0 = stage_b(kind0, 64 + lane, a0); b1 = stage_b(kind1, 64 + lane, a1);
        stage_a(kind0, 128 + lane, a0); stage_a(kind1, 128 + lane, a1);
        int buf = 0;
        for (int i0 = 0; i0 < n_it; i0 += 64, buf ^= 1) {
            lds[buf][0][lane] = t0;
            lds[buf][1][lane] = t1;
            __builtin_amdgcn_wave_barrier();
            const float c0 = b0, c1 = b1;                                                                 /* B of tile i0 + 64: issued one tile ago */
            if (i0 + 128 < n_it) { b0 = stage_b(kind0, i0 + 128 + lane, a0); b1 = stage_b(kind1, i0 + 128 + lane, a1); }
            if (i0 + 192 < n_it) { stage_a(kind0, i0 + 192 + lane, a0); stage_a(kind1, i0 + 192 + lane, a1); }
            if (lane < 2) {
                const double2* col = reinterpret_cast<const double2*>(lds[buf][lane]);
                #pragma unroll
                for (int q = 0; q < 32; q += 4) {
                    const double2 a = col[q], b = col[q + 1], c2 = col[q + 2], d2 = col[q + 3];
                    acc += a.x; acc += a.y; acc += b.x; acc += b.y; acc += c2.x; acc += c2.y; acc += d2.x; acc += d2.y;
                }
            }
            if (i0 + 64 < n_it) { t0 = stage_c(kind0, i0 + 64 + lane, c0); t1 = stage_c(kind1, i0 + 64 + lane, c1); }
        }
        __builtin_amdgcn_wave_barrier();
        out0 = __shfl(acc, 0, 64); out1 = __shfl(acc, 1, 64);
    };
    double ev_sum = ps_e, km_sum = ps_k, unused = 0.0;
    if (!exact_k && !exact_e) run(KM_SUM, EV_SUM, km_sum, ev_sum);
    else if (!exact_k) run(KM_SUM, NONE, km_sum, unused);
    else if (!exact_e) run(EV_SUM, NONE, ev_sum, unused);
    shift = ev_sum / n_ev - km_sum / K;                                 /* align.c:86 */
    double km_sq = 0.0, ev_sq = 0.0;
    run(KM_SQ, EV_SQ, km_sq, ev_sq);
    if (lane != 0) return;
    const double scale = (ev_sq / n_ev) / (km_sq / K);                  /* align.c:95 */
    abea_scalings_t o; o.shift = (float)shift; o.scale = (float)scale; o.var = 1.0f; o.log_var = 0.0f;
    scalings[r] = o;
}
