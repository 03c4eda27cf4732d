/* abea_hmm.hip — EXPERIMENTAL, NOT BUILT INTO libabea_hip.so, NOT YET RUN ON A GPU (written at the end of round 1 when
 * the round's GPU budget was spent; compile-checked only:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -c abea_hmm.hip).
 *
 * Row N4 of SURVEY §8f: the profile-HMM forward score of call-methylation, profile_hmm_score_r9
 * (reference src/hmm.c:314-735, called from meth.c:473 twice per CpG group: unmethylated and methylated sequence).
 * The CPU statement of the same computation lives with the test infrastructure (tests/test_hmm_oracle.py).
 *
 * Mapping.  The matrix has one row per event and three states (K skip, B bad event, M match) per k-mer block:
 *     M[r][b] <- row r-1 : M,B of block b and M,B,K of block b-1 (+ the soft start in block 0)
 *     B[r][b] <- row r-1 : M,B of block b
 *     K[r][b] <- row r   : M,B,K of block b-1                       (serial along the row)
 * so every cell of an anti-diagonal d = r + b is independent.  One wavefront scores one job, lane = k-mer block, one
 * step per diagonal; a lane keeps its last two (M,B,K) triples in registers and reads its left neighbour's with DPP
 * wave_shr:1 (the start block, all -inf, is the DPP `old` operand).  The table-driven float log-sum (logsum.h:61-71) is
 * NOT associative, so nothing is re-ordered: each state adds its terms in the reference's order and the end state
 * accumulates row by row in the last block's lane.  The 16000-entry table (64 000 B, built on the host with glibc so
 * that it is the reference's table bit for bit) sits in LDS, shared by the four wavefronts of a workgroup.  Every
 * operation is fp32 (+ exact fp32 division), so the result can equal the CPU's bit for bit.
 * Sequences longer than 64 k-mers run in tiles of 64 blocks with the tile's last column parked in global scratch.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/abea.h"

#define ABEA_HMM_TBL 16000
#define NINF_F (-__builtin_inff())

struct abea_hmm_job {                 /* built on the host per (read, CpG group, sequence variant) */
    int64_t event_off;                /* read's event table in `events` */
    int64_t col_off;                  /* 3 * (n_events + 1) floats of column scratch (only used when n_kmers > 64) */
    int32_t seq_off, rc_seq_off;      /* m_seq / m_rc_seq in `seqs` (hmm.c:628-630) */
    int32_t seq_len;
    int32_t e_start, e_stop;          /* event_start_idx / event_stop_idx (inclusive, meth.c:457-462) */
    int32_t stride;                   /* +1 / -1 */
    int32_t rc;                       /* bam_is_rev */
    uint32_t flags;                   /* HAF_ALLOW_PRE_CLIP = 1, HAF_ALLOW_POST_CLIP = 2 (f5cmisc.h:40-41) */
    float scale, shift, var, log_var; /* scalings_t of the read */
    float lp_mk, lp_mb, lp_mm_self, lp_mm_next, lp_bb, lp_bk, lp_bm_next, lp_bm_self, lp_kk, lp_km;   /* hmm.c:240-310,
                                         computed on the host from events_per_base with glibc log() */
    int32_t out_idx;
    int32_t pad;
};

static __device__ __forceinline__ float hmm_logsum(const float* __restrict__ tbl, float a, float b) {
    const float mx = (a > b) ? a : b, mn = (a < b) ? a : b;          /* ESL_MAX / ESL_MIN, logsum.h:19-20 */
    const float diff = __fsub_rn(mx, mn);
    if (mn == NINF_F || diff >= 15.7f) return mx;
    return __fadd_rn(mx, tbl[(int)__fmul_rn(diff, 1000.f)]);
}
static __device__ __forceinline__ float from_left(float v) {         /* lane b <- lane b-1; lane 0 <- -inf (start block) */
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(NINF_F), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
static __device__ __forceinline__ uint32_t cpg_rank(const char* s, int k) {   /* hmm.c:30-61 */
    uint32_t r = 0;
    for (int i = 0; i < k; ++i) {
        const char c = s[i];
        r = r * 5u + (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'M' ? 3u : c == 'T' ? 4u : 0u);
    }
    return r;
}

extern "C" __global__ __launch_bounds__(256)
void abea_hmm_forward_kernel(int n_jobs, const abea_hmm_job* __restrict__ jobs, const char* __restrict__ seqs,
                             const abea_event_t* __restrict__ events, const abea_model_t* __restrict__ cpgmodel,
                             int kmer_size, const float* __restrict__ logsum_tbl /* 16000 */,
                             const float* __restrict__ flank /* flank[i], i <= max events: pre_flank; post_flank[i] =
                                                                flank[n_events - 1 - i] (hmm.c:141-233) */,
                             float* col_scratch, float* __restrict__ out) {
    __shared__ float tbl[ABEA_HMM_TBL];
    for (int i = threadIdx.x; i < ABEA_HMM_TBL; i += blockDim.x) tbl[i] = logsum_tbl[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_jobs) return;                                          /* whole wavefront */
    const abea_hmm_job jb = jobs[j];
    const int k = kmer_size;
    const int n_k = jb.seq_len - k + 1;
    const int n_ev = (jb.e_stop > jb.e_start ? jb.e_stop - jb.e_start : jb.e_start - jb.e_stop) + 1;
    const abea_event_t* __restrict__ ev = events + jb.event_off;
    float* col = col_scratch + jb.col_off;
    float end = NINF_F;
    const int n_tiles = (n_k + 63) / 64;
    for (int t = 0; t < n_tiles; ++t) {
        const int b = t * 64 + lane;                                  /* k-mer index of this lane */
        const bool has = b < n_k;
        float c0 = 0.f, gp_mean = 0.f, gp_stdv = 1.f;
        if (has) {
            const char* s = jb.rc == 0 ? seqs + jb.seq_off + b : seqs + jb.rc_seq_off + jb.seq_len - b - k;   /* hmm.c:383-397 */
            const abea_model_t m = cpgmodel[cpg_rank(s, k)];
            gp_mean = __fadd_rn(__fmul_rn(jb.scale, m.level_mean), jb.shift);       /* hmm.c:92-95 */
            gp_stdv = __fmul_rn(m.level_stdv, jb.var);
            c0 = __fsub_rn(-0.918938f, __fadd_rn(m.level_log_stdv, jb.log_var));   /* hmm.c:64-70,103 */
        }
        const bool last_block = b == n_k - 1;
        const int tile_w = min(64, n_k - t * 64);
        float LM = NINF_F, LB = NINF_F, LK = NINF_F;                  /* this lane's row r-1 (row 0 is -inf, hmm.c:613-625) */
        float PM = NINF_F, PB = NINF_F, PK = NINF_F;                  /* ... and row r-2 */
        for (int d = 1; d <= n_ev + tile_w - 1; ++d) {
            /* the left neighbour's row r (its last step) and row r-1 (the step before) */
            float nLM = from_left(LM), nLB = from_left(LB), nLK = from_left(LK);
            float nPM = from_left(PM), nPB = from_left(PB), nPK = from_left(PK);
            const int r = d - lane;                                   /* 1-based row of this lane on this diagonal */
            const bool act = has && r >= 1 && r <= n_ev;
            if (lane == 0 && t > 0 && act) {                          /* left neighbour lives in the previous tile */
                nLM = col[3 * r]; nLB = col[3 * r + 1]; nLK = col[3 * r + 2];
                nPM = col[3 * (r - 1)]; nPB = col[3 * (r - 1) + 1]; nPK = col[3 * (r - 1) + 2];
            }
            if (act) {
                const int e = jb.e_start + (r - 1) * jb.stride;
                const float a = __fdiv_rn(__fsub_rn(ev[e].mean, gp_mean), gp_stdv);
                const float lp_em = __fadd_rn(c0, __fmul_rn(__fmul_rn(-0.5f, a), a));
                /* MATCH (hmm.c:436-451): terms in the reference's order; logsum(x, -inf) = x */
                float s = __fadd_rn(jb.lp_mm_self, LM);
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_mm_next, nPM));
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_bm_self, LB));
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_bm_next, nPB));
                s = hmm_logsum(tbl, s, __fadd_rn(jb.lp_km, nPK));
                const bool soft = (b == 0) && (e == jb.e_start || (jb.flags & 1u));
                s = hmm_logsum(tbl, s, soft ? __fadd_rn(0.0f, flank[r - 1]) : NINF_F);
                const float M = __fadd_rn(s, lp_em);
                /* BAD_EVENT (hmm.c:453-460) */
                const float B = __fadd_rn(hmm_logsum(tbl, __fadd_rn(jb.lp_mb, LM), __fadd_rn(jb.lp_bb, LB)), 0.0f);
                /* KMER_SKIP (hmm.c:462-469): same row, previous block */
                float q = hmm_logsum(tbl, NINF_F, __fadd_rn(jb.lp_mk, nLM));
                q = hmm_logsum(tbl, q, __fadd_rn(jb.lp_bk, nLB));
                q = hmm_logsum(tbl, q, __fadd_rn(jb.lp_kk, nLK));
                const float K = __fadd_rn(q, 0.0f);
                PM = LM; PB = LB; PK = LK;
                LM = M; LB = B; LK = K;
                if (last_block && ((jb.flags & 2u) || r == n_ev)) {   /* hmm.c:474-486 */
                    const float pf = flank[n_ev - r];                 /* post_flank[r-1] */
                    end = hmm_logsum(tbl, end, __fadd_rn(__fadd_rn(0.0f, M), pf));
                    end = hmm_logsum(tbl, end, __fadd_rn(__fadd_rn(0.0f, B), pf));
                    end = hmm_logsum(tbl, end, __fadd_rn(__fadd_rn(0.0f, K), pf));
                }
                if (lane == tile_w - 1 && t + 1 < n_tiles) {          /* park the tile's last column for the next tile */
                    col[3 * r] = M; col[3 * r + 1] = B; col[3 * r + 2] = K;
                }
            }
        }
        if (t + 1 < n_tiles) {
            if (lane == 0) { col[0] = NINF_F; col[1] = NINF_F; col[2] = NINF_F; }   /* row 0 */
            /* the column is written and read back by this wavefront only (other wavefronts of the workgroup run other
             * jobs, possibly with a different number of tiles: no workgroup barrier here) */
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    const int last_lane = (n_k - 1) & 63;
    if (lane == last_lane) out[jb.out_idx] = end;
}
