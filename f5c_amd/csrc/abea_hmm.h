/* abea_hmm.h — device job descriptor of the profile-HMM kernel (abea_hmm.hip), shared with its host driver (abea_hmm.cpp). */
#ifndef ABEA_HMM_H
#define ABEA_HMM_H
#include <stdint.h>
#include "../../include/abea.h"

#define ABEA_HMM_TBL 16000                /* p7_LOGSUM_TBL, logsum.h:17 */

typedef struct {
    int64_t ev_off;                       /* the job's event means in row order (event_start_idx, then +stride ...) */
    int64_t col_off;                      /* 3 * (n_events + 1) floats of column scratch (only used when n_kmers > 64) */
    int32_t seq_off;                      /* the sequence the k-mers are read from: m_seq (rc == 0) or m_rc_seq (hmm.c:383-397) */
    int32_t seq_len;
    int32_t n_events;
    int32_t rc;
    uint32_t flags;                       /* HAF_ALLOW_PRE_CLIP = 1, HAF_ALLOW_POST_CLIP = 2 (f5cmisc.h:40-41) */
    int32_t out_idx;
    float scale, shift, var, log_var;     /* scalings_t of the read */
    float lp_mk, lp_mb, lp_mm_self, lp_mm_next, lp_bb, lp_bk, lp_bm_next, lp_bm_self, lp_kk, lp_km;   /* hmm.c:240-310 */
} abea_hmm_job;

#endif
