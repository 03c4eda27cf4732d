/* include/abea.h — C ABI of libabea_hip.so: MI355X-native adaptive banded event alignment.
 *
 * Drop-in boundary for the GPU branch of f5c's align_db() (reference src/f5c.c:833-845).
 * Every entry point names the reference interface it replaces.  Plain pointers and sizes
 * only; no C++/STL/htslib/torch types cross this boundary.  All functions return 0 on
 * success and a negative ABEA_E* code on failure (abea_last_error() has the message);
 * the f5c-facing shim (f5c_amd/csrc/f5c_shim.h) turns a failure into the reference's
 * print-and-exit convention (src/error.h:38-92, src/f5cmisc.cuh:78-97).
 *
 * There is no CPU fallback inside this library: a read is either aligned on the GPU or
 * reported as skipped (n_pairs = 0 by the reference's own guards).
 */
#ifndef ABEA_H
#define ABEA_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABEA_BANDWIDTH 100          /* ALN_BANDWIDTH, src/f5c.h:34 (compile-time in the reference too) */
#define ABEA_MAX_KMER_SIZE 9        /* MAX_KMER_SIZE, src/f5c.h:30 */

#define ABEA_OK            0
#define ABEA_EINVAL       -1        /* bad argument */
#define ABEA_EHIP         -2        /* HIP runtime error */
#define ABEA_ENOMEM       -3        /* a single read does not fit the device arena */
#define ABEA_ENODEV       -4        /* no usable gfx950 device */
#define ABEA_EBUSY        -5        /* submitted host batches are in flight / no free lane */

/* ---- POD mirrors of the f5c data contract (layout-identical, static_asserted in abea_capi.cpp) ---- */
typedef struct { uint64_t start; float length; float mean; float stdv; } abea_event_t;          /* event_t     src/f5c.h:129-136 */
typedef struct { float level_mean; float level_stdv; float level_log_stdv; } abea_model_t;       /* model_t     src/f5c.h:147-155 */
typedef struct { float scale; float shift; float var; float log_var; } abea_scalings_t;          /* scalings_t  src/f5c.h:158-172 */
typedef struct { int32_t ref_pos; int32_t read_pos; } abea_pair_t;                               /* AlignedPair src/f5c.h:181-184 */
typedef struct { int32_t start; int32_t stop; } abea_index_pair_t;                               /* index_pair_t src/f5c.h:187-190 */

/* Per-read quantities align() computes but f5c does not return (src/align.c:415-445,526-535);
 * optional output used by the parity tests for the "scores within 1e-4" check. */
typedef struct {
    double  sum_emission;   /* Σ log-emission along the path, reverse path order (align.c:476) */
    int32_t n_aligned;      /* pairs emitted before QC (align.c:480) */
    int32_t best_event;     /* end event chosen by the end-point scan (align.c:442) */
    float   max_score;      /* its score (align.c:441); -inf if no end cell was in band */
    int32_t max_gap;        /* align.c:497 */
    int32_t spanned;        /* align.c:529-530 */
    int32_t flags;          /* ABEA_RF_* */
    int32_t pad;            /* diagnostic, not part of the reference: trace groups the traceback had to re-load whole */
} abea_read_diag;
#define ABEA_RF_SKIPPED   0x1   /* failed the align_single guards (f5c.c:813-814) or shorter than k: n_pairs = 0 */
#define ABEA_RF_NO_END    0x2   /* no in-band end cell (max_score == -inf): n_pairs = 0 */
#define ABEA_RF_QC_FAIL   0x4   /* align.c:534-543 */

/* THREADING CONTRACT.  A context owns shared mutable state (device arena, stream slots with pinned staging, worker pool):
 * it serves ONE call at a time, from any host thread (the thread may change between calls, as f5c's pthread_processor
 * does, src/meth_main.c:668-689).  The library enforces it: every entry that touches that state holds a per-context
 * mutex, so a second thread calling into the same context BLOCKS until the first call returns (it never corrupts the
 * arena).  Callers that want concurrency — e.g. the methylation stage under pthread_db — use one context per thread, or
 * abea_align_batch_host_submit()/_wait() below, the one sanctioned way to have several batches in flight on one context. */
typedef struct abea_ctx abea_ctx;   /* opaque; owns the device arena, the model copy, a stream (cuda_data_t, src/f5c.h:356-386) */

typedef struct {
    int32_t  device_id;         /* opt.cuda_dev_id            src/f5c.h:124 */
    uint32_t kmer_size;         /* core->kmer_size            src/f5c.h:399-ish; 4^k model entries */
    const abea_model_t* model;  /* core->model (host)         src/f5c.c:286 */
    float    mem_frac;          /* opt.cuda_mem_frac          fraction of free device memory for the arena (0 -> 0.9, MEM_FACTOR f5cmisc.cuh:48) */
    uint64_t max_arena_bytes;   /* 0 = no cap; otherwise cap the arena (tests/bench keep room for the batch itself) */
    int32_t  verbosity;         /* opt.verbosity */
    int32_t  reserved;
} abea_cfg;

/* Replaces init_cuda(core_t*)   src/f5c.cu:23-202  (device select, model H2D, one-shot arena). */
int  abea_init(abea_ctx** ctx, const abea_cfg* cfg);
/* The same over several GPUs of one node (BASELINE north_star: "batches of reads shard embarrassingly across the 8
 * GPUs of one node (per-GPU hipStreams ...)"; the reference has only --cuda-dev-id, one device per process,
 * docs/f5c.1:271): a context that owns one device context per entry of device_ids.  abea_align_batch_host() then
 * splits every batch over the devices (longest-processing-time-first on the band count E+K), one host
 * thread and one set of streams per device, each writing straight into the caller's per-read buffers; no data crosses
 * between devices.  cfg->device_id is ignored; a device may be listed twice (two contexts on one GPU).  The
 * device-resident entries (abea_align_batch_device, abea_detect_events_device) need a single-device context. */
int  abea_init_multi(abea_ctx** ctx, const abea_cfg* cfg, const int32_t* device_ids, int32_t n_devices);
int32_t abea_device_count(abea_ctx* ctx);
/* Replaces free_cuda(core_t*)   src/f5c.cu:204-234. */
void abea_free(abea_ctx* ctx);
/* Message of the last failure on the calling thread. */
const char* abea_last_error(void);

/* ---- host batch: the db_t fields align_cuda() reads and writes (src/f5c.cu:647-1061) ---- */
typedef struct {
    int32_t n_reads;                       /* db->n_bam_rec */
    const char* const* read;               /* db->read[i]   NUL-terminated, upper case */
    const int32_t* read_len;               /* db->read_len[i] */
    const abea_event_t* const* events;     /* db->et[i].event */
    const uint64_t* n_events;              /* db->et[i].n */
    const abea_scalings_t* scalings;       /* db->scalings[i] (scale, shift used) */
    const int64_t* n_samples;              /* db->sig[i]->nsample ; NULL = all reads good */
    abea_pair_t* const* pairs;             /* db->event_align_pairs[i], caller-allocated, capacity n_events+read_len (f5c.c:724);
                                              may be NULL when base_to_event_map is given (pair lists stay on the device) */
    int32_t* n_pairs;                      /* db->n_event_align_pairs[i] */
    abea_read_diag* diag;                  /* optional [n_reads], may be NULL */
    /* ---- optional: scaling_db() fused with the alignment (src/f5c.c:736-807 scaling_single = postalign +
     *      recalibrate_model, src/align.c:561-773; row N1).  The pair lists have no other consumer in process_db
     *      (src/f5c.c:907-960: align_db, then pthread_db(scaling_single), then meth_single reads only base_to_event_map /
     *      scalings / events_per_base).  The wavefront that aligned a read also recalibrates it (last phase of the alignment
     *      kernel).  What crosses PCIe downwards: the 2-bit walk (0.4 B per event; expanded by the host workers into the pair
     *      lists when pairs != NULL), one event-count byte per k-mer from which the host rebuilds base_to_event_map (its entries
     *      tile the path's events in k order), and 36 B of scalars per read.  All NULL = alignment only. ---- */
    abea_index_pair_t* const* base_to_event_map;   /* db->base_to_event_map[i]: caller-allocated, read_len-k+1 entries;
                                                      written only for reads with n_pairs > 0 (the reference leaves NULL otherwise) */
    abea_scalings_t* scalings_out;         /* [n_reads] db->scalings[i] after recalibration: shift, scale, var and log_var = (float)log(var)
                                              (double var, glibc log: align.c:758-760, CACHED_LOG) of a recalibrated read; the input
                                              copied unchanged otherwise ; may alias `scalings` */
    double*  events_per_base;              /* [n_reads] db->events_per_base[i] */
    int32_t* read_stat_flag;               /* [n_reads] in/out: ABEA_FAILED_* bits OR-ed in (src/f5c.h:66-68) */
    int32_t* n_event_alignment;            /* [n_reads] db->n_event_alignment[i] */
    int32_t  min_num_events_to_rescale;    /* opt.min_num_events_to_rescale; 0 -> 200 */
    int32_t  flags;                        /* ABEA_HB_* */
} abea_host_batch;
/* base_to_event_map[i] is an OUTPUT pointer: the library malloc()s the map of every read that aligned (read_len - k + 1 entries,
 * where scaling_single -> postalign mallocs it, f5c.c:746) and stores NULL for the others (f5c.c:787); the caller free()s.  On
 * failure everything allocated by the call is released and the entries are NULL. */
#define ABEA_HB_MALLOC_MAPS 0x1

/* Replaces align_cuda(core_t*, db_t*)  src/f5c.cu:647-1061 : flatten + H2D + kernels + D2H.
 * The batch is cut into chunks of reads, longest reads first; chunks rotate through a few stream slots so that the host
 * loops (flatten: event means + sequences into pinned memory; un-flatten: results into the caller's per-read buffers),
 * both PCIe directions and the kernels of neighbouring chunks overlap.  What crosses PCIe downwards is the traceback
 * walk itself, 2 bits per step; the (k-mer, event) pairs are expanded from it on the host straight into
 * db->event_align_pairs[i] (the role of the reversal loop at f5c.cu:1003-1030).  ABEA_HOST_PAIRS=device in the
 * environment moves the expansion back to the GPU (pair lists compacted on the device, then copied down).
 * Results do not depend on the chunking, the mode or the number of devices. */
int abea_align_batch_host(abea_ctx* ctx, const abea_host_batch* batch);

/* ---- several host batches in flight on one context (f5c's default -K 512 / -B 2M batches are latency-bound: one lasts
 * as long as its longest read and fills an eighth of the GPU's wave slots, so a drop-in at default flags leaves the
 * device idle; the reference overlaps only I/O with processing, src/meth_main.c:668-689) ----
 * abea_align_batch_host_submit() starts the batch on a free LANE — a disjoint share of the context's stream slots,
 * device arena and worker threads — on a thread of the library, and returns a ticket at once;
 * abea_align_batch_host_wait() blocks until that batch's outputs are complete and returns its status.  `batch` is copied;
 * every array it points to (inputs AND outputs) must stay valid and untouched until the wait returns.  Up to
 * abea_set_inflight() batches (default 2, at most ABEA_MAX_INFLIGHT) can be in flight; with all lanes busy submit returns
 * ABEA_EBUSY.  Results are bit-identical to abea_align_batch_host().  While batches are in flight the synchronous entries
 * of the same context return ABEA_EBUSY.  Works on multi-device contexts (every device runs lane l of each batch).
 * abea_get_stats() reports the batch of the last successful wait.  A ticket is redeemed once: a second wait on it returns
 * ABEA_EBUSY while the first is still blocked and ABEA_EINVAL afterwards. */
#define ABEA_MAX_INFLIGHT 8
int abea_set_inflight(abea_ctx* ctx, int32_t n_lanes);      /* only while nothing is in flight; a lane gets 1/n of slots, arena, threads */
int abea_align_batch_host_submit(abea_ctx* ctx, const abea_host_batch* batch, int32_t* ticket);
int abea_align_batch_host_wait(abea_ctx* ctx, int32_t ticket);

/* ---- device-resident flattened batch: the layout of the reference's device arrays (src/f5c.cu:672-690) ---- */
typedef struct {
    int32_t n_reads;
    /* index arrays: HOST pointers (the reference keeps host copies too: cuda_data_t *_host, f5c.h:357-366) */
    const int64_t* read_ptr;     /* offset of read i in `reads` (chars)                 */
    const int32_t* read_len;
    const int64_t* event_ptr;    /* offset of read i in `events`                        */
    const int32_t* n_events;
    const int64_t* pair_ptr;     /* offset of read i in `pairs`; capacity n_events+read_len */
    const abea_scalings_t* scalings;   /* HOST pointer, [n_reads] */
    /* bulk arrays: DEVICE pointers */
    const char* reads;           /* flattened sequences, Σ(read_len+1)                  */
    const abea_event_t* events;  /* flattened AoS event tables, Σ n_events              */
    abea_pair_t* pairs;          /* out                                                  */
    int32_t* n_pairs;            /* out [n_reads]                                        */
    abea_read_diag* diag;        /* out [n_reads], optional (NULL)                       */
    /* ---- optional: scaling_single() on the device (src/f5c.c:736-807 = postalign + recalibrate_model,
     *      src/align.c:561-773), run right after the alignment while the pairs are still in HBM.
     *      All NULL = skipped.  Row N1 of SURVEY §8f. ---- */
    const int64_t* kmer_ptr;               /* HOST: offset of read i in base_to_event_map (read_len-k+1 entries each) */
    abea_index_pair_t* base_to_event_map;  /* DEVICE out: db->base_to_event_map[i][k] */
    abea_scalings_t* scalings_io;          /* DEVICE in/out [n_reads]: estimated scalings in, recalibrated
                                              shift/scale/var out when calibrated (log_var is left to the host) */
    double* events_per_base;               /* DEVICE out [n_reads]: db->events_per_base[i] */
    int32_t* read_stat_flag;               /* DEVICE in/out [n_reads]: FAILED_* bits OR-ed in (src/f5c.h:66-68) */
    int32_t* n_event_alignment;            /* DEVICE out [n_reads]: db->n_event_alignment[i] */
    int32_t  min_num_events_to_rescale;    /* opt.min_num_events_to_rescale (src/f5c.c:1185: 200); 0 -> 200 */
    int32_t  reserved;
} abea_device_batch;
#define ABEA_FAILED_CALIBRATION 0x001      /* src/f5c.h:66 */
#define ABEA_FAILED_ALIGNMENT   0x002      /* src/f5c.h:67 */
#define ABEA_FAILED_QUALITY_CHK 0x004      /* src/f5c.h:68 */

/* Same computation with inputs/outputs already in HBM (what bench.py times). Synchronous. */
int abea_align_batch_device(abea_ctx* ctx, const abea_device_batch* batch);

/* ---- raw-signal batch: the front half of event_single() on the device (src/f5c.c:682-712; row N2) ----
 * ADC samples -> pA -> getevents() (src/events.c:562-582) -> estimate_scalings_using_mom() (src/align.c:58-106).
 * Index / scaling arrays are HOST pointers, bulk arrays DEVICE pointers.  Events of read i are written at
 * events[event_ptr[i] ...] up to event_cap[i] entries; n_events[i] is the true count (> cap means truncated: the caller
 * must re-run that read with a larger table, a truncated table is not a valid input of the alignment).
 * rna == 0: the detector runs with event_detection_defaults (src/events.c:52-58: windows 3 / 6, thresholds 1.4 / 9.0, peak
 * height 0.2).  rna != 0 (opt.flag & F5C_RNA, f5c.c:698-702): event_detection_rna (events.c:59-65,575-577: windows 7 / 14,
 * thresholds 2.5 / 9.0, peak height 1.0), the scalings are estimated on the table in detection order and the table is then
 * written reversed, 3'->5', as event_single() does (f5c.c:711-719) — ready for the alignment with the RNA k-mer model. */
typedef struct {
    int32_t n_reads;
    const int64_t* sig_ptr;        /* HOST: offset of read i in `signal` (samples); multiples of 8 are fastest (16-byte loads) */
    const int32_t* n_samples;      /* HOST: db->sig[i]->nsample */
    const float*   scaling;        /* HOST [n_reads][3]: offset, range, digitisation (signal_t, src/f5c.h:276-286) */
    const int64_t* event_ptr;      /* HOST */
    const int32_t* event_cap;      /* HOST */
    const int64_t* read_ptr;       /* HOST: offset of read i in `reads` (for the scalings); may be NULL with scalings */
    const int32_t* read_len;       /* HOST */
    const int16_t* signal;         /* DEVICE: raw ADC samples */
    const char*    reads;          /* DEVICE: flattened sequences (NULL = no scalings) */
    abea_event_t*  events;         /* DEVICE out */
    int32_t*       n_events;       /* DEVICE out [n_reads] */
    abea_scalings_t* scalings;     /* DEVICE out [n_reads] (scale, shift, var = 1), optional */
    int32_t        rna;            /* opt.flag & F5C_RNA */
    int32_t        reserved;
} abea_signal_batch;
int abea_detect_events_device(abea_ctx* ctx, const abea_signal_batch* batch);

/* ---- event_db on host buffers (row N2): pthread_db(core, db, event_single), src/f5c.c:682-734 ----
 * f5c keeps a read's signal as FLOAT ADC counts (signal_t.rawptr, src/f5c.h:276-286; the slow5 / fast5 readers widen int16)
 * and event_single() turns them into pA in place, runs getevents(), estimates the scalings and — RNA — reverses the table.
 * Here the flatten loop narrows the counts back to int16 (2 bytes per sample over PCIe; a sample that is not an integral
 * int16 value is refused with ABEA_EINVAL: it cannot be an ADC count), the detector of abea_detect_events_device runs on
 * the device chunk by chunk, and each read's table comes back as a malloc()ed event_t array — where getevents()
 * (src/events.c:562-582) would have put it; the caller releases it with free() as free_db_tmp does.  Reads with
 * n_samples <= 0 get events = NULL, n_events = 0 (f5c.c:727-731).  A read whose table overflows the first guess of its
 * size is redone with the exact size inside the call.  On failure no table is handed back (all events[i] are NULL), but with
 * signal_to_pa_in_place the signals of the chunks that had completed are already in pA: the batch cannot simply be re-submitted. */
typedef struct {
    int32_t n_reads;                       /* db->n_bam_rec */
    float* const* rawptr;                  /* db->sig[i]->rawptr: ADC counts as float; rewritten to pA when signal_to_pa_in_place */
    const int64_t* n_samples;              /* db->sig[i]->nsample */
    const float* offset;                   /* db->sig[i]->offset        [n_reads] */
    const float* range;                    /* db->sig[i]->range         [n_reads] */
    const float* digitisation;             /* db->sig[i]->digitisation  [n_reads] */
    const char* const* read;               /* db->read[i]; NULL with scalings == NULL */
    const int32_t* read_len;               /* db->read_len[i] */
    int32_t rna;                           /* core->opt.flag & F5C_RNA */
    int32_t signal_to_pa_in_place;         /* != 0: rawptr[j] = (rawptr[j] + offset) * (range / digitisation) as f5c.c:693-696 leaves it */
    abea_event_t** events;                 /* out [n_reads]: db->et[i].event, malloc()ed by the library */
    uint64_t* n_events;                    /* out [n_reads]: db->et[i].n */
    abea_scalings_t* scalings;             /* out [n_reads]: db->scalings[i] = estimate_scalings_using_mom(); may be NULL */
} abea_events_host_batch;
int abea_events_batch_host(abea_ctx* ctx, const abea_events_host_batch* batch);

/* ---- the chain event_db -> align_db -> scaling_db on host buffers (row N3): process_db_rsq, src/resquiggle.c:283-315
 * (process_db starts with the same three steps, src/f5c.c:907-936) ----
 * One call: raw signals and sequences in; event tables, (optionally) pair lists, base_to_event_map, recalibrated scalings,
 * events_per_base, read_stat_flag, n_event_alignment out.  Per-read buffers are malloc()ed by the library exactly where
 * the reference mallocs them — et[i].event (getevents), event_align_pairs[i] (f5c.c:722-725, n_events + read_len entries),
 * base_to_event_map[i] (scaling_single -> postalign, f5c.c:746; NULL for a read that did not align) — and released by the
 * caller with free().  The alignment stage is the chunk pipeline of abea_align_batch_host with scaling_single fused. */
typedef struct {
    int32_t n_reads;
    float* const* rawptr; const int64_t* n_samples;          /* as abea_events_host_batch */
    const float* offset; const float* range; const float* digitisation;
    const char* const* read; const int32_t* read_len;
    int32_t rna, signal_to_pa_in_place;
    abea_event_t** events; uint64_t* n_events;               /* out: db->et[i] */
    abea_scalings_t* scalings;                               /* out: db->scalings[i] after scaling_single (recalibrated when it could be) */
    abea_scalings_t* scalings_estimated;                     /* out, optional: the method-of-moments estimate before it */
    abea_pair_t** pairs;                                     /* out, optional (NULL = pair lists not returned): db->event_align_pairs[i] */
    int32_t* n_pairs;                                        /* out: db->n_event_align_pairs[i] */
    abea_read_diag* diag;                                    /* out, optional */
    abea_index_pair_t** base_to_event_map;                   /* out: db->base_to_event_map[i] */
    double* events_per_base;                                 /* out: db->events_per_base[i] */
    int32_t* read_stat_flag;                                 /* in/out: ABEA_FAILED_* bits OR-ed in */
    int32_t* n_event_alignment;                              /* out */
    int32_t min_num_events_to_rescale;                       /* 0 -> 200 */
    int32_t reserved;
} abea_process_batch;
int abea_process_batch_host(abea_ctx* ctx, const abea_process_batch* batch);

/* ---- resquiggle output of one read (row N3): the per-read body of output_db_rsq() (src/resquiggle.c:319-449) ----
 * fmt 0 = TSV (one line per k-mer: read_id, k-mer index, first sample, one-past-last sample or "."), fmt 1 = PAF
 * (one line; the ss:Z: string encodes matched samples "n,", skipped samples "nI", k-mers without events "nD").
 * `base_to_event_map` is db->base_to_event_map[i] (host copy; reversed IN PLACE when rna != 0, as the reference does),
 * `events` = db->et[i].event, `n_samples` = db->sig[i]->nsample.  scale / shift are what the reference prints in the
 * sc:f: / sh:f: tags: db->scalings->scale, i.e. the FIRST read's of the batch (resquiggle.c:443-444).
 * snprintf-like: writes at most cap-1 bytes plus a NUL and returns the length of the full text; < 0 when the map is
 * inconsistent (the reference asserts / exits there).  Host-only. */
int64_t abea_rsq_format(char* out, size_t cap, int fmt, const char* read_id, int32_t read_len, uint32_t kmer_size,
                        abea_index_pair_t* base_to_event_map, const abea_event_t* events, int64_t n_samples,
                        float scale, float shift, int rna);

/* The loop of output_db_rsq() over a batch (src/resquiggle.c:319-449): every read whose read_stat_flag is clear, in batch
 * order, through abea_rsq_format; as in the reference the sc:f: / sh:f: tags carry the FIRST read's scalings
 * (db->scalings->scale).  The caller's maps are left untouched (an RNA map is reversed on a copy, so a size query followed
 * by the real call prints the same text).  snprintf-like (returns the full length, writes at most
 * cap-1 bytes + NUL; out may be NULL with cap 0); n_printed (optional) = reads printed. */
int64_t abea_rsq_format_batch(char* out, size_t cap, int fmt, int32_t n_reads, const char* const* read_id,
                              const int32_t* read_len, uint32_t kmer_size, abea_index_pair_t* const* base_to_event_map,
                              const abea_event_t* const* events, const int64_t* n_samples, const abea_scalings_t* scalings,
                              const int32_t* read_stat_flag, int rna, int32_t* n_printed);

/* ---- profile-HMM forward scores (row N4): batches of profile_hmm_score() calls, src/hmm.c:689-735 ----
 * One job = one call as calculate_methylation_for_read issues them (src/meth.c:473-474: the unmethylated and the
 * methylated sequence of a CpG group against one read's events).  Argument meaning as in the reference; `strand` is
 * ignored there (hmm.c:79 sets it to 0) and absent here.  The score is the table-driven (ESL_LOG_SUM) float forward
 * log-probability, bit for bit.  Host pointers in, scores[n_jobs] out.  cpgmodel has 5^kmer_size entries (alphabet
 * A,C,G,M,T, hmm.c:30-61), level_log_stdv cached as in model.c:179. */
typedef struct {
    const char* m_seq;                 /* HMMInputSequence: NUL-terminated, may hold 'M' (methylated C) */
    const char* m_rc_seq;              /* its reverse complement (meth.c:366-400), same length */
    const abea_event_t* events;        /* db->et[i].event of the read */
    abea_scalings_t scaling;           /* db->scalings[i] (scale, shift, var, log_var all used: hmm.c:92-103) */
    uint32_t event_start_idx, event_stop_idx;    /* inclusive; stop < start on the reverse strand (meth.c:457-462) */
    int8_t   event_stride;             /* +1 / -1 */
    uint8_t  rc;                       /* bam_is_rev */
    uint16_t pad;
    uint32_t hmm_flags;                /* HAF_ALLOW_PRE_CLIP = 1, HAF_ALLOW_POST_CLIP = 2 (f5cmisc.h:40-41) */
    double   events_per_base;          /* db->events_per_base[i] */
} abea_hmm_job_t;
int abea_hmm_score_batch_host(abea_ctx* ctx, const abea_hmm_job_t* jobs, int32_t n_jobs, const abea_model_t* cpgmodel,
                              uint32_t kmer_size, float* scores);
/* The same with the event tables still in HBM after the chain (abea_detect_events_device -> abea_align_batch_device):
 * every job's `events` is the DEVICE address of its read's table (events + event_ptr[i]); the windows are gathered on
 * the device.  Sequences, scalings and events_per_base are host values as above; scores[] is a host array. */
int abea_hmm_score_batch_device(abea_ctx* ctx, const abea_hmm_job_t* jobs, int32_t n_jobs, const abea_model_t* cpgmodel,
                                uint32_t kmer_size, float* scores);

/* ---- timing / accounting of the last batch (core_t timing fields, src/f5c.h:457-466) ---- */
typedef struct {
    double pre_ms, fill_ms, trace_ms;     /* HIP-event kernel times on the library's stream, summed over sub-batches
                                             (fill = the fused fill + traceback kernel, which since round 4 also runs
                                             scaling_single for its read when that is requested; trace_ms = 0, kept for
                                             the layout) */
    double h2d_ms, d2h_ms, host_ms;       /* host batch only */
    double event_ms;                      /* abea_detect_events_device: kernel time of the last call */
    double hmm_ms;                        /* abea_hmm_score_batch_host: kernel time of the last call */
    double total_ms;                      /* wall time of the call */
    int64_t n_reads_gpu, n_reads_skipped, n_sub_batches;
    int64_t sum_events, sum_bands, sum_pairs;
    int64_t fill_launches;                /* number of fill-kernel launches (== n_sub_batches) */
    uint64_t arena_bytes;
    uint64_t bytes_ref;                   /* algorithmic bytes A_ref of the reads run */
    uint64_t bytes_min;                   /* strict floor A_min */
    uint64_t bytes_moved;                 /* bytes this implementation reads+writes in HBM by construction */
    /* host batch only */
    double flatten_ms, unflatten_ms;      /* wall time of the two host loops (they overlap the copies and kernels) */
    double wait_ms;                       /* time the calling thread spent waiting for the GPU */
    uint64_t h2d_bytes, d2h_bytes;        /* PCIe traffic of the call */
    int32_t n_devices, host_threads;
    double plan_ms, setup_ms;             /* caller-thread time that is neither loop nor wait: per-chunk layout (plan) and the
                                             guards + ordering + chunk carving before the first chunk (setup) */
    double gpu_busy_ms;                   /* host batch / raw-signal entries: length of the union of the chunks' kernel intervals on the
                                             GPU's clock (first kernel start .. last kernel end per chunk): total_ms - gpu_busy_ms = time
                                             the device had nothing of this call to run */
} abea_stats;
int abea_get_stats(abea_ctx* ctx, abea_stats* out);
/* Multi-device context: the share of the last host batch that ran on device_ids[device]; abea_get_stats() gives the
 * whole batch (counts summed, times = the slowest device). */
int abea_get_device_stats(abea_ctx* ctx, int32_t device, abea_stats* out);
/* The split abea_align_batch_host() applies on a multi-device context, exposed for callers that shard themselves
 * (one process per GPU): reads in descending weight (band count n_events + n_kmers + 2; <= 0 for reads the guards skip)
 * go to the currently lightest of n_bins bins, ties to the lowest bin.  Host-only, needs no context. */
int abea_lpt_split(const int64_t* weight, int32_t n, int32_t n_bins, int32_t* bin_of);
/* The un-flatten step of the host entry on its own (host-only): the traceback walk as it crosses PCIe — 2 bits per
 * step, 16 steps per word, step 0 = the end cell (last_kmer, end_event); 0 = diagonal, 1 = up (event only), 2 = left
 * (k-mer only) — expanded into the ascending (ref_pos, read_pos) list align() returns (src/align.c:452-513). */
int abea_expand_walk_codes(const uint32_t* codes, int32_t n_steps, int32_t last_kmer, int32_t end_event, abea_pair_t* out);
/* The same walk expanded straight into db->base_to_event_map[i] (postalign, src/align.c:571-596; last_kmer + 1 entries):
 * what the host entry does when scaling_single is fused — the map never crosses PCIe either.  Host-only. */
int abea_expand_walk_codes_to_map(const uint32_t* codes, int32_t n_steps, int32_t last_kmer, int32_t end_event,
                                  abea_index_pair_t* map);
/* And from the form in which the fused scaling_single phase hands the map to the host entry: the number of events of every
 * k-mer's entry (stop - start + 1; 0 = {-1, -1}).  The entries tile the events of the path in k order, the last non-empty one
 * ending at end_event, so the counts determine the map.  A count of 255 means "255 or more" and is refused here (the host entry
 * rebuilds such a read's map from the walk).  Host-only. */
int abea_expand_kmer_counts_to_map(const uint8_t* count, int32_t n_kmers, int32_t end_event, abea_index_pair_t* map);
/* The flatten loop of the host entry on its own (host-only): means[e] = events[e].mean for e < n_events — 4 of event_t's 24
 * bytes, the only field align() reads (src/align.c:131).  means must be 16-byte aligned (non-temporal stores);
 * prefetch_bytes = distance of the software prefetch ahead of the loads (0 = none), hint 0 / 1 / 2 = nta / t0 / t2. */
int abea_flatten_event_means(const abea_event_t* events, int32_t n_events, float* means, int32_t prefetch_bytes, int32_t hint);
/* The chunk plan abea_align_batch_host() uses for a batch on an arena of arena_bytes (host-only; pairs returned, no
 * fused scaling): chunk_of[i] = launch-order number of the chunk read i goes into, -1 for reads skipped by the
 * align_single guards (src/f5c.c:813-814).  Reads go longest first; a chunk holds >= 1024 reads and >= 24 M events (the
 * first two a quarter / half of that), at most 16384 reads, and fits an eighth of the arena; ABEA_HOST_CHUNK_* and
 * ABEA_HOST_SLOTS in the environment override the numbers. */
int abea_host_plan_chunks(const int32_t* read_len, const int32_t* n_events, int32_t n_reads, uint32_t kmer_size,
                          uint64_t arena_bytes, int32_t* chunk_of, int32_t* n_chunks);

/* The worker-thread / CPU-affinity plan of the host entry (host-only, pure): per DEVICE context
 * threads = (usable_cpus - 2) / n_devices clamped to [1, 16] (ABEA_HOST_THREADS overrides, per device), bound to the CPUs of
 * the device's NUMA node that the process may run on when the machine has several nodes and that set is at least as large
 * — when ABEA_HOST_NUMA=1.  (The run-time DEFAULT, round 5, is ABEA_HOST_NUMA=spread: the workers alternate over the NUMA nodes, which
 * took the box-to-box spread out of the flatten loop — 214-219 ms per 100 k reads where an unbound pool measured 231-333; =0 leaves
 * them unbound.  This report covers the =1 mode only.)  The =1 binding is opt-in, and this report and the run-time path read the SAME switch: the host
 * loops read the caller's event tables (24 B per event) from wherever the caller placed them and write 4 B per event of
 * staging, so binding a device's workers to its node pays only if the caller's buffers are on that node too (measured:
 * DESIGN.md §5); without ABEA_HOST_NUMA=1 every bind list is "".  cpulists use the sysfs format ("0-63,128-191"); allowed_cpulist NULL or "" = all.
 * bind_cpulists (may be NULL) receives n_devices strings of cap_each bytes, "" = not bound.  At run time the inputs come
 * from sched_getaffinity, the cgroup CPU quota, /sys/bus/pci/devices/<bdf>/numa_node and /sys/devices/system/node. */
int abea_host_plan_threads(int32_t usable_cpus, const char* allowed_cpulist, int32_t n_devices, const int32_t* device_numa_node,
                           int32_t n_nodes, const char* const* node_cpulist, int32_t* threads_per_device, char* bind_cpulists,
                           size_t cap_each);

/* What the host<->device link of the context's GPU delivers, measured the way the pipelines use it (pinned host memory; `bytes`
 * per transfer, 0 = 256 MiB; `reps` transfers per measurement, 0 = 4).  GB/s (1e9 bytes) into out[0..7):
 *   [0] host->device, hipMemcpyAsync            [1] device->host, hipMemcpyAsync        [2] device->host, abea_copy_out_kernel
 *   [3], [4] host->device (copy engine) and device->host (kernel) while both run at once — what the chunk pipelines do
 *   [5], [6] the same with both directions on the copy engines
 *   [7] host->device by the copy kernel (loads from pinned host memory); [8], [9] host->device by kernel and device->host by copy
 *   engine at once (written when n_out >= 10).
 * The raw-signal entries move 2 B per sample up and 24 B per event down (event_db, src/f5c.c:682-734): bench.py sets their
 * measured rates against these ceilings.  Diagnostic; nothing in the library depends on it. */
int abea_link_probe(abea_ctx* ctx, uint64_t bytes, int32_t reps, double* out, int32_t n_out);

/* Library / device introspection: "gfx950", CU count; used by tests to assert the native path ran. */
int abea_device_info(abea_ctx* ctx, char* arch, size_t arch_len, int32_t* n_cu, uint64_t* arena_bytes);

/* Hardware self-test of the cross-lane primitives the fill kernel relies on (DPP wave shifts,
 * readlane/writelane); returns 0 when the semantics are as assumed. */
int abea_selftest(abea_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* ABEA_H */
