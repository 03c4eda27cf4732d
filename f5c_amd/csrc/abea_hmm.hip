/* abea_hmm.hip — row N4 of SURVEY §8f: the profile-HMM forward score of call-methylation on gfx950.
 *
 * What it computes: profile_hmm_score -> profile_hmm_score_r9 -> profile_hmm_fill_generic_r9<ProfileHMMForwardOutputR9>
 * (reference src/hmm.c:314-735, called from meth.c:473 twice per CpG group: unmethylated and methylated sequence), as
 * the reference is compiled: ESL_LOG_SUM (table-driven float log-sum, logsum.h:61-71), CACHED_LOG, no HMM_REVERSE_FIX.
 *
 * Mapping.  The matrix has one row per event and three states (K skip, B bad event, M match) per k-mer block:
 *     M[r][b] <- row r-1 : M,B of block b and M,B,K of block b-1 (+ the soft start in block 0)
 *     B[r][b] <- row r-1 : M,B of block b
 *     K[r][b] <- row r   : M,B,K of block b-1                       (serial along the row)
 * so every cell of an anti-diagonal d = r + b is independent.  A job is scored by a SEGMENT of SEG lanes of a
 * wavefront (SEG = 16: four jobs per wave, DPP row_shr; SEG = 64: one job per wave, DPP wave_shr), lane = k-mer block,
 * one step per diagonal; a lane keeps its last two (M,B,K) triples in registers and reads its left neighbour's with one
 * DPP shift (the start block, all -inf, is the DPP `old` operand at the segment's first lane).  The table-driven
 * log-sum is NOT associative, so nothing is re-ordered: each state adds its terms in the reference's order and the end
 * state accumulates row by row in the last block's lane.  The 16000-entry table (64 000 B, built on the host with glibc
 * so that it is the reference's table bit for bit) sits in LDS, shared by the four wavefronts of a workgroup.  Every
 * operation is fp32 with the reference's association, so the score equals the CPU's bit for bit.
 * Sequences with more k-mers than SEG run in tiles of SEG blocks with the tile's last column parked in global scratch.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "abea_hmm.h"

#define NINF_F (-__builtin_inff())

static __device__ __forceinline__ float hmm_logsum(const float* __restrict__ tbl, float a, float b) {
    const float mx = (a > b) ? a : b, mn = (a < b) ? a : b;          /* ESL_MAX / ESL_MIN, logsum.h:19-20 */
    if (mn == NINF_F) return mx;
    const float diff = __fsub_rn(mx, mn);
    if (diff >= 15.7f) return mx;
    return __fadd_rn(mx, tbl[(int)__fmul_rn(diff, 1000.f)]);
}
/* lane b <- lane b-1 within the segment; the segment's first lane <- -inf (the start block, hmm.c:613-625) */
template <int SEG> static __device__ __forceinline__ float from_left(float v) {
    if (SEG == 64)
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(NINF_F), __float_as_int(v), 0x138, 0xf, 0xf, false));   /* wave_shr:1 */
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(NINF_F), __float_as_int(v), 0x111, 0xf, 0xf, false));       /* row_shr:1 */
}
static __device__ __forceinline__ uint32_t cpg_rank(const char* s, int k) {   /* hmm.c:30-61: alphabet A,C,G,M,T */
    uint32_t r = 0;
    for (int i = 0; i < k; ++i) {
        const char c = s[i];
        r = r * 5u + (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'M' ? 3u : c == 'T' ? 4u : 0u);
    }
    return r;
}

template <int SEG> static __device__ __forceinline__
void hmm_forward(int first_job, int n_jobs, const abea_hmm_job* __restrict__ jobs, const char* __restrict__ seqs,
                 const float* __restrict__ evw, const abea_model_t* __restrict__ cpgmodel, int k,
                 const float* tbl, const float* __restrict__ flank, float* col_scratch, float* __restrict__ out) {
    constexpr int PER_WAVE = 64 / SEG;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane64 = threadIdx.x & 63;
    const int lane = lane64 & (SEG - 1);                              /* k-mer block within the tile */
    const int j = wave * PER_WAVE + lane64 / SEG;
    const bool live = j < n_jobs;
    abea_hmm_job jb;
    if (live) jb = jobs[first_job + j];
    else { jb.seq_len = k; jb.n_events = 0; jb.ev_off = 0; jb.col_off = 0; jb.seq_off = 0; jb.rc = 0; jb.flags = 0; jb.out_idx = 0; }
    const int n_k = live ? jb.seq_len - k + 1 : 0;
    const int n_ev = live ? jb.n_events : 0;
    const float* __restrict__ ev = evw + jb.ev_off;
    float* col = col_scratch + jb.col_off;
    float end = NINF_F;
    const int n_tiles = (n_k + SEG - 1) / SEG;
    /* every lane of the wave runs the longest schedule of its wave's jobs (DPP needs all lanes at the same step) */
    int my_tiles = n_tiles;
    for (int off = SEG; off < 64; off <<= 1) my_tiles = max(my_tiles, __shfl_xor(my_tiles, off, 64));
    for (int t = 0; t < my_tiles; ++t) {
        const int b = t * SEG + lane;                                 /* k-mer index of this lane */
        const bool has = b < n_k;
        float c0 = 0.f, gp_mean = 0.f, gp_stdv = 1.f;
        if (has) {
            const char* s = seqs + jb.seq_off + (jb.rc == 0 ? b : jb.seq_len - b - k);   /* hmm.c:383-397 */
            const abea_model_t m = cpgmodel[cpg_rank(s, k)];
            gp_mean = __fadd_rn(__fmul_rn(jb.scale, m.level_mean), jb.shift);       /* hmm.c:92-93 */
            gp_stdv = __fmul_rn(m.level_stdv, jb.var);                              /* hmm.c:94 */
            c0 = __fsub_rn(-0.918938f, __fadd_rn(m.level_log_stdv, jb.log_var));   /* hmm.c:64-70,101-103 */
        }
        const bool last_block = b == n_k - 1;
        const int tile_w = max(0, min(SEG, n_k - t * SEG));
        int steps = (t < n_tiles) ? n_ev + tile_w - 1 : 0;
        for (int off = SEG; off < 64; off <<= 1) steps = max(steps, __shfl_xor(steps, off, 64));
        float LM = NINF_F, LB = NINF_F, LK = NINF_F;                  /* this lane's row r-1 (row 0 is -inf, hmm.c:613-625) */
        float PM = NINF_F, PB = NINF_F, PK = NINF_F;                  /* ... and row r-2 */
        for (int d = 1; d <= steps; ++d) {
            /* the left neighbour's row r (its last step) and row r-1 (the step before) */
            float nLM = from_left<SEG>(LM), nLB = from_left<SEG>(LB), nLK = from_left<SEG>(LK);
            float nPM = from_left<SEG>(PM), nPB = from_left<SEG>(PB), nPK = from_left<SEG>(PK);
            const int r = d - lane;                                   /* 1-based row of this lane on this diagonal */
            const bool act = has && r >= 1 && r <= n_ev;
            if (lane == 0 && t > 0 && act) {                          /* left neighbour lives in the previous tile */
                nLM = col[3 * r]; nLB = col[3 * r + 1]; nLK = col[3 * r + 2];
                nPM = col[3 * (r - 1)]; nPB = col[3 * (r - 1) + 1]; nPK = col[3 * (r - 1) + 2];
            }
            if (act) {
                const float a = __fdiv_rn(__fsub_rn(ev[r - 1], gp_mean), gp_stdv);      /* hmm.c:68 */
                const float lp_em = __fadd_rn(c0, __fmul_rn(__fmul_rn(-0.5f, a), a));   /* hmm.c:69 */
                /* MATCH (hmm.c:436-451): terms in the reference's order; logsum(x, -inf) = x */
                float s = __fadd_rn(jb.lp_mm_self, LM);
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_mm_next, nPM));
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_bm_self, LB));
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_bm_next, nPB));
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_km, nPK));
                /* event_idx == e_start <=> row 1 (the event window is uploaded in row order) */
                const bool soft = (b == 0) && (r == 1 || (jb.flags & 1u));
                s = hmm_logsum(tbl, s, soft ? __fadd_rn(0.0f, flank[r - 1]) : NINF_F);
                const float M = __fadd_rn(s, lp_em);
                /* BAD_EVENT (hmm.c:453-460) */
                const float B = __fadd_rn(hmm_logsum(tbl, __fadd_rn(jb.lp_mb, LM), __fadd_rn(jb.lp_bb, LB)), 0.0f);
                /* KMER_SKIP (hmm.c:462-469): same row, previous block */
                float q = __fadd_rn(jb.lp_mk, nLM);                   /* logsum(-inf, x) = x */
                q = hmm_logsum(tbl, q, __fadd_rn(jb.lp_bk, nLB));
                q = hmm_logsum(tbl, q, __fadd_rn(jb.lp_kk, nLK));
                const float K = __fadd_rn(q, 0.0f);
                PM = LM; PB = LB; PK = LK;
                LM = M; LB = B; LK = K;
                if (last_block && ((jb.flags & 2u) || r == n_ev)) {   /* hmm.c:474-486 */
                    const float pf = flank[n_ev - r];                 /* post_flank[r-1] = pre_flank[n_ev-r] */
                    end = hmm_logsum(tbl, end, __fadd_rn(__fadd_rn(0.0f, M), pf));
                    end = hmm_logsum(tbl, end, __fadd_rn(__fadd_rn(0.0f, B), pf));
                    end = hmm_logsum(tbl, end, __fadd_rn(__fadd_rn(0.0f, K), pf));
                }
                if (lane == tile_w - 1 && t + 1 < n_tiles) {          /* park the tile's last column for the next tile */
                    col[3 * r] = M; col[3 * r + 1] = B; col[3 * r + 2] = K;
                }
            }
        }
        if (t + 1 < n_tiles && lane == 0) { col[0] = NINF_F; col[1] = NINF_F; col[2] = NINF_F; }   /* row 0 */
        /* the column is written and read back by this wavefront only (other wavefronts run other jobs, possibly with a
         * different number of tiles: no workgroup barrier here); make the stores visible to the wave's own loads */
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (live && lane == ((n_k - 1) & (SEG - 1))) out[jb.out_idx] = end;
}

/* jobs [0, n16) have at most 16 k-mers and run four to a wave; jobs [n16, n_jobs) run one to a wave */
extern "C" __global__ __launch_bounds__(256)
void abea_hmm_forward_kernel(int n16, int n_jobs, int blocks16, const abea_hmm_job* __restrict__ jobs,
                             const char* __restrict__ seqs, const float* __restrict__ evw,
                             const abea_model_t* __restrict__ cpgmodel, int kmer_size,
                             const float* __restrict__ logsum_tbl, const float* __restrict__ flank,
                             float* col_scratch, float* __restrict__ out) {
    __shared__ float tbl[ABEA_HMM_TBL];
    for (int i = threadIdx.x; i < ABEA_HMM_TBL; i += blockDim.x) tbl[i] = logsum_tbl[i];
    __syncthreads();
    if ((int)blockIdx.x < blocks16) {
        hmm_forward<16>(0, n16, jobs, seqs, evw, cpgmodel, kmer_size, tbl, flank, col_scratch, out);
    } else {
        /* re-base the block index for the wide jobs */
        const int wave = ((int)blockIdx.x - blocks16) * 4 + (threadIdx.x >> 6);
        if (wave >= n_jobs - n16) return;
        /* hmm_forward derives the job from blockIdx: shift the job window instead */
        hmm_forward<64>(n16 - blocks16 * 4, n_jobs - n16 + blocks16 * 4, jobs, seqs, evw, cpgmodel, kmer_size, tbl, flank, col_scratch, out);
    }
}


/* device-resident event tables (abea_hmm_score_batch_device): the event-mean window of every job, in row order
 * (event_idx = e_start + row * stride, hmm.c:432), gathered from the read's AoS event_t table where the chain left it in
 * HBM.  One wavefront per job (a window is a few dozen events). */
extern "C" __global__ __launch_bounds__(256)
void abea_hmm_gather_kernel(int n_jobs, const abea_hmm_job* __restrict__ jobs, const int64_t* __restrict__ table,
                            const int32_t* __restrict__ start_stride, float* __restrict__ windows) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n_jobs) return;
    const abea_hmm_job d = jobs[q];
    const abea_event_t* __restrict__ ev = reinterpret_cast<const abea_event_t*>((uintptr_t)table[q]);
    const int64_t e0 = start_stride[2 * q], stride = start_stride[2 * q + 1];
    float* __restrict__ w = windows + d.ev_off;
    for (int r = threadIdx.x & 63; r < d.n_events; r += 64) w[r] = ev[e0 + (int64_t)r * stride].mean;
}
