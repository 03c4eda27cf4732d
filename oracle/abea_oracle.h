/* oracle/abea_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of f5c's adaptive banded event alignment (ABEA), used as the
 * checker for the HIP path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (f5c_amd/libabea_hip.so) never links, loads or calls it.
 *
 * Parity pin: checked against the reference's own known-answer fixtures
 *   test/ecoli_2kb_region/single_read/{read1.fasta,read1.events.exp,
 *   adaptive.exp,read1.scalings.exp} (committed as tests/golden/single_read.npz)
 * The genuine reference object is NOT built here: src/align.c includes f5c.h,
 * which includes htslib headers this image lacks, so it is unbuildable without
 * stand-in headers (see DESIGN.md "Oracle").
 */
#ifndef ABEA_ORACLE_H
#define ABEA_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t start; float length; float mean; float stdv; } orc_event_t;   /* f5c.h:129-136 */
typedef struct { float level_mean; float level_stdv; float level_log_stdv; } orc_model_t; /* f5c.h:147-155 (CACHED_LOG) */
typedef struct { int32_t ref_pos; int32_t read_pos; } orc_pair_t;                        /* f5c.h:181-184 */
typedef struct { float scale; float shift; float var; float log_var; } orc_scalings_t;   /* f5c.h:158-172 */

/* per-read diagnostics the reference computes but does not return (align.c:415-416,424-445,526) */
typedef struct {
    double  sum_emission;      /* Σ lp over the path, reverse path order, double */
    int32_t n_aligned;         /* pairs emitted before QC */
    int32_t best_event;        /* curr_event_idx after the end scan */
    float   max_score;         /* best end score (−inf if none) */
    int32_t max_gap;
    int32_t spanned;
    int32_t oob;               /* 1 if the traceback would have read outside the trace buffer (UB in the reference) */
    int32_t pad;
} orc_diag_t;

uint32_t orc_kmer_rank(const char* s, uint32_t k);                       /* align.c:19-47 */
orc_scalings_t orc_estimate_scalings(const char* seq, int32_t seq_len, const orc_model_t* model,
                                     uint32_t k, const orc_event_t* ev, size_t n_events); /* align.c:58-106 */
int32_t orc_align(orc_pair_t* out, const char* seq, int32_t seq_len,
                  const orc_event_t* ev, size_t n_events, const orc_model_t* model,
                  uint32_t k, float scale, float shift, orc_diag_t* diag);                /* align.c:180-559 */

/* align_single guard + align (f5c.c:811-830): nsample<=0 or E/L >= 15.0f -> 0 */
int32_t orc_align_single(orc_pair_t* out, const char* seq, int32_t seq_len,
                         const orc_event_t* ev, size_t n_events, int64_t nsample,
                         const orc_model_t* model, uint32_t k, float scale, float shift, orc_diag_t* diag);

/* batch driver shaped like pthread_db(core, db, align_single) (f5c.c:575-679):
 * static block partition + atomic work stealing over n_threads pthreads.
 * Flattened batch: read chars at read_ptr[i] (NUL terminated), events at event_ptr[i],
 * output pairs at pair_ptr[i] (capacity n_events[i]+read_len[i]). */
void orc_align_batch(int32_t n_reads, const char* reads, const int64_t* read_ptr, const int32_t* read_len,
                     const orc_event_t* events, const int64_t* event_ptr, const int32_t* n_events,
                     const orc_scalings_t* scalings, const orc_model_t* model, uint32_t k,
                     orc_pair_t* pairs, const int64_t* pair_ptr, int32_t* n_pairs, orc_diag_t* diags,
                     int32_t n_threads);

/* Event detection: raw pA samples -> event table.  getevents (events.c:562-582), i.e.
 * detect_events on the whole signal (the trim result there is discarded), DNA parameters
 * (events.c:52-56).  Returns the number of events written (<= nsample); out must hold nsample entries. */
size_t orc_getevents(size_t nsample, const float* raw_pa, orc_event_t* out);
/* the same with the molecule type: rna != 0 selects event_detection_rna (events.c:59-65); table in detection order */
size_t orc_getevents_rna(size_t nsample, const float* raw_pa, orc_event_t* out, int rna);
void orc_reverse_events(orc_event_t* events, size_t n_events);          /* f5c.c:711-719 */
/* ADC -> pA conversion of event_single (f5c.c:692-696), in place on a float copy of the int16 samples */
void orc_raw_to_pa(float* raw, size_t nsample, float offset, float range, float digitisation);

/* glibc allocator tuning for the multi-threaded CPU baseline (see abea_oracle.c) */
void orc_malloc_tuning(int on);

/* postalign + recalibrate_model (align.c:561-773, f5c.c:736-807) : "next" row N1 */
typedef struct { int32_t start; int32_t stop; } orc_index_pair_t;        /* f5c.h:187-190 */
int32_t orc_scaling_single(const orc_pair_t* pairs, int32_t n_pairs, const char* seq, int32_t seq_len,
                           const orc_event_t* ev, size_t n_events, const orc_model_t* model, uint32_t k,
                           orc_scalings_t* scalings /*in/out*/, orc_index_pair_t* base_to_event_map /*K*/,
                           double* events_per_base, int32_t* read_stat_flag);
/* resquiggle text of one read = the per-read body of output_db_rsq() (src/resquiggle.c:319-449); see abea_oracle.c */
long orc_rsq_format(char* out, size_t cap, int fmt, const char* read_id, int32_t read_len, uint32_t kmer_size,
                    orc_index_pair_t* map, const orc_event_t* event, long nsample, float sc_scale, float sc_shift, int rna);

/* profile-HMM forward score (src/hmm.c:314-735), row N4; pinned to single_read/meth_input.exp + meth.exp (tests/test_hmm_pin.py).  hmm_flags: 1 = HAF_ALLOW_PRE_CLIP,
 * 2 = HAF_ALLOW_POST_CLIP (f5cmisc.h:40-41).  cpgmodel: 5^k entries over the alphabet A,C,G,M,T. */
void orc_flogsum_init(void);
const float* orc_flogsum_table(void);
uint32_t orc_cpg_kmer_rank(const char* str, uint32_t k);
float orc_profile_hmm_score(const char* m_seq, const char* m_rc_seq, const orc_event_t* event, orc_scalings_t scaling,
                            const orc_model_t* cpgmodel, uint32_t kmer_size, uint32_t event_start_idx,
                            uint32_t event_stop_idx, int8_t event_stride, uint8_t rc, double events_per_base,
                            uint32_t hmm_flags);

/* analysis hook (tools/walk_drift.py): orc_align() on this thread copies every band's lower-left k-mer index
 * (band_lower_left[b].kmer_idx, align.c:270-322) into buf; NULL = off */
void orc_debug_band_llk(int32_t* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
