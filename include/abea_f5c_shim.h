/* include/abea_f5c_shim.h — align_db()-shaped shim over the C ABI (include/abea.h).
 *
 * What an f5c maintainer links instead of src/f5c.cu + src/align.cu (or their .hip twins): three
 * void functions with the reference's own names' shape and error convention
 *     init_cuda(core_t*)            src/f5c.cu:23    -> abea_f5c_init(abea_f5c_core*)
 *     align_cuda(core_t*, db_t*)    src/f5c.cu:647   -> abea_f5c_align(abea_f5c_core*, abea_f5c_db*)
 *     free_cuda(core_t*)            src/f5c.cu:204   -> abea_f5c_free(abea_f5c_core*)
 * operating on POD views that name exactly the core_t / db_t fields the reference's GPU path touches
 * (SURVEY.md §8b).  INTEGRATION.md shows the ~25-line glue that fills the views inside f5c.
 * Errors: like the reference (src/error.h:38-92, f5cmisc.cuh:78-97) the shim prints to stderr and
 * exit(EXIT_FAILURE)s; it never returns a status.
 */
#ifndef ABEA_F5C_SHIM_H
#define ABEA_F5C_SHIM_H
#include "abea.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {                       /* event_table, src/f5c.h:139-144 */
    size_t n; size_t start; size_t end; abea_event_t* event;
} abea_f5c_event_table;

typedef struct {                       /* the slice of core_t used by init_cuda/align_cuda/free_cuda */
    const abea_model_t* model;         /* core->model                         f5c.h:424 */
    uint32_t kmer_size;                /* core->kmer_size                     f5c.h:426 */
    int32_t  cuda_dev_id;              /* core->opt.cuda_dev_id               f5c.h:124 */
    float    cuda_mem_frac;            /* core->opt.cuda_mem_frac             f5c.h:125 */
    int32_t  verbosity;                /* core->opt.verbosity */
    void*    cuda;                     /* core->cuda (opaque; owned by the shim) f5c.h:455 */
    /* timing accumulators, seconds, same meaning as core_t's (f5c.h:457-466) */
    double align_kernel_time, align_pre_kernel_time, align_core_kernel_time, align_post_kernel_time;
    double align_cuda_memcpy, align_cuda_preprocess, align_cuda_postprocess, align_cuda_total_kernel;
    /* one process, several GPUs (the reference has only --cuda-dev-id, one device per process, docs/f5c.1:271):
     * n_cuda_devs > 1 makes abea_f5c_init build a multi-device context over cuda_dev_ids[] and every batch is
     * split over them inside abea_f5c_align; 0 = the single device cuda_dev_id */
    const int32_t* cuda_dev_ids;
    int32_t n_cuda_devs;
    int32_t min_num_events_to_rescale; /* core->opt.min_num_events_to_rescale (abea_f5c_align_scale only) */
    int32_t rna;                       /* core->opt.flag & F5C_RNA (abea_f5c_event_db / abea_f5c_process only) */
    int32_t inflight;                  /* lanes for abea_f5c_align_submit (1..ABEA_MAX_INFLIGHT); 0 = the library's default, 2 */
    double event_time;                 /* core->event_time: seconds spent in abea_f5c_event_db / the event stage of abea_f5c_process */
} abea_f5c_core;

typedef struct {                       /* the slice of db_t align_cuda reads/writes, src/f5c.h:290-352 */
    int32_t n_bam_rec;
    char** read;                       /* db->read[i] */
    int32_t* read_len;
    const int64_t* nsample;            /* db->sig[i]->nsample gathered into an array; NULL = all good */
    abea_f5c_event_table* et;          /* db->et */
    abea_scalings_t* scalings;         /* db->scalings */
    abea_pair_t** event_align_pairs;   /* db->event_align_pairs (caller-allocated, f5c.c:724) */
    int32_t* n_event_align_pairs;      /* db->n_event_align_pairs */
    int64_t sum_bases;                 /* db->sum_bases (statistics only) */
    /* abea_f5c_align_scale only: what scaling_single writes (f5c.c:736-807) */
    abea_index_pair_t** base_to_event_map; /* db->base_to_event_map: entry i is malloc()ed here when read i aligned, NULL otherwise */
    double* events_per_base;           /* db->events_per_base */
    int32_t* read_stat_flag;           /* db->read_stat_flag (FAILED_* bits are OR-ed in) */
    int32_t* n_event_alignment;        /* db->n_event_alignment */
    /* abea_f5c_event_db / abea_f5c_process only: the signal_t fields event_single reads (src/f5c.h:276-286), gathered */
    float** rawptr;                    /* db->sig[i]->rawptr (ADC counts as float; left as pA like f5c.c:693-696) */
    const float* offset;               /* db->sig[i]->offset */
    const float* range;                /* db->sig[i]->range */
    const float* digitisation;         /* db->sig[i]->digitisation */
} abea_f5c_db;

void abea_f5c_init(abea_f5c_core* core);
void abea_f5c_align(abea_f5c_core* core, abea_f5c_db* db);
/* align_db() + pthread_db(scaling_single) in one call (process_db, f5c.c:924-936): the pair lists stay on the device,
 * base_to_event_map / recalibrated db->scalings / events_per_base / read_stat_flag come back.  db->event_align_pairs
 * may be NULL (or given, then it is filled as well, e.g. for --print-banded-aln). */
void abea_f5c_align_scale(abea_f5c_core* core, abea_f5c_db* db);
/* abea_f5c_align split in two, so that a caller can keep several process_db batches in flight — f5c's default -K 512 /
 * -B 2M batches are latency-bound on a GPU.  submit returns a handle at once; wait blocks until
 * db->event_align_pairs / n_event_align_pairs are complete.  db must stay valid and untouched between.  At most
 * core->inflight handles (set before abea_f5c_init; 0 = the library's default of 2, at most ABEA_MAX_INFLIGHT) may be
 * outstanding: a submit beyond that is misuse and exits like every other error here — wait for a handle first.  Each
 * lane owns 1/inflight of the device arena and of the worker threads, so a read that fits the synchronous call can be
 * too long for a lane (ABEA_ENOMEM, exit); INTEGRATION.md gives the arithmetic. */
void* abea_f5c_align_submit(abea_f5c_core* core, abea_f5c_db* db);
void abea_f5c_align_wait(abea_f5c_core* core, void* handle);
/* event_db: pthread_db(core, db, event_single) (f5c.c:682-734) on the GPU.  Fills db->et[i] (malloc()ed tables, n = 0 /
 * event = NULL for reads without signal), db->scalings[i] (method of moments), leaves db->rawptr[i] in pA and malloc()s
 * db->event_align_pairs[i] (n + read_len entries, f5c.c:722-725) when that array is given. */
void abea_f5c_event_db(abea_f5c_core* core, abea_f5c_db* db);
/* process_db_rsq (resquiggle.c:283-315) = event_db -> align_db -> scaling_db in one call; every output of the three. */
void abea_f5c_process(abea_f5c_core* core, abea_f5c_db* db);
void abea_f5c_free(abea_f5c_core* core);

#ifdef __cplusplus
}
#endif
#endif
