/* integration/abea_glue.c — copy to f5c's src/abea_glue.c and build with `make abea=1 ABEA_ROOT=<this repo>` after applying
 * integration/f5c_makefile.patch (it drops f5c.cu / f5c_gpuonly.cu / align.cu from the build: this file defines the three symbols
 * they provide — init_cuda src/f5c.h:577, free_cuda src/f5c.h:580, align_cuda src/f5cmisc.h:124 — on top of libabea_hip.so).
 * INTEGRATION.md quotes the two parts below verbatim; tests/test_integration_glue.py compiles THIS file against the reference's
 * real headers and checks the quotes. */
// src/abea_glue.c — f5c side of the MI355X ABEA library (replaces f5c.cu)
#include "f5c.h"
#include "f5cmisc.h"            // align_cuda prototype (f5cmisc.h:124)
#include "abea_f5c_shim.h"      // from this repo's include/

static abea_f5c_core g_core;    // f5c has a single core_t per process (meth_main.c:619)

void init_cuda(core_t* core) {                       // prototype src/f5c.h:577
    memset(&g_core, 0, sizeof g_core);
    g_core.model = (const abea_model_t*)core->model; // model_t == abea_model_t (12 B, CACHED_LOG)
    g_core.kmer_size = core->kmer_size;
    g_core.cuda_dev_id = core->opt.cuda_dev_id;      // or: g_core.cuda_dev_ids / n_cuda_devs for one process on several GPUs
    g_core.cuda_mem_frac = core->opt.cuda_mem_frac;
    g_core.verbosity = core->opt.verbosity;
    g_core.min_num_events_to_rescale = core->opt.min_num_events_to_rescale;
    abea_f5c_init(&g_core);                          // prints + exit(EXIT_FAILURE) on error, like CUDA_CHK
    core->cuda = (cuda_data_t*)g_core.cuda;
}

void align_cuda(core_t* core, db_t* db) {            // prototype src/f5cmisc.h:124
    int32_t n = db->n_bam_rec;
    int64_t* nsample = (int64_t*)malloc(sizeof(int64_t) * n);
    for (int32_t i = 0; i < n; i++) nsample[i] = db->sig[i]->nsample;
    abea_f5c_db v;
    memset(&v, 0, sizeof v);
    v.n_bam_rec = n;           v.read = db->read;              v.read_len = db->read_len;
    v.nsample = nsample;       v.et = (abea_f5c_event_table*)db->et;   // same 4-field layout, f5c.h:139
    v.scalings = (abea_scalings_t*)db->scalings;               // scalings_t == abea_scalings_t (16 B)
    v.event_align_pairs = (abea_pair_t**)db->event_align_pairs;
    v.n_event_align_pairs = db->n_event_align_pairs;
    v.sum_bases = db->sum_bases;
    abea_f5c_align(&g_core, &v);
    core->align_kernel_time      = g_core.align_kernel_time;       // reported by meth_main.c:749-796
    core->align_pre_kernel_time  = g_core.align_pre_kernel_time;
    core->align_core_kernel_time = g_core.align_core_kernel_time;
    core->align_post_kernel_time = g_core.align_post_kernel_time;
    core->align_cuda_memcpy      = g_core.align_cuda_memcpy;
    core->align_cuda_preprocess  = g_core.align_cuda_preprocess;
    core->align_cuda_postprocess = g_core.align_cuda_postprocess;
    free(nsample);
}

void free_cuda(core_t* core) { abea_f5c_free(&g_core); core->cuda = NULL; }   // prototype src/f5c.h:580

// src/abea_glue.c, continued — the resquiggle chain (resquiggle.c:289-307) in one call
void process_db_rsq_gpu(core_t* core, db_t* db) {    // instead of pthread_db(event_single) + align_db + pthread_db(scaling_single)
    int32_t n = db->n_bam_rec;
    float** raw = (float**)malloc(sizeof(float*) * n);
    float *off = (float*)malloc(4 * n), *rng = (float*)malloc(4 * n), *dig = (float*)malloc(4 * n);
    int64_t* nsample = (int64_t*)malloc(8 * n);
    for (int32_t i = 0; i < n; i++) {
        raw[i] = db->sig[i]->rawptr;  nsample[i] = db->sig[i]->nsample;
        off[i] = db->sig[i]->offset;  rng[i] = db->sig[i]->range;  dig[i] = db->sig[i]->digitisation;
    }
    abea_f5c_db v;
    memset(&v, 0, sizeof v);
    v.n_bam_rec = n;           v.read = db->read;              v.read_len = db->read_len;
    v.nsample = nsample;       v.et = (abea_f5c_event_table*)db->et;
    v.scalings = (abea_scalings_t*)db->scalings;
    v.event_align_pairs = (abea_pair_t**)db->event_align_pairs;   // entries are malloc()ed inside (f5c.c:722-725)
    v.n_event_align_pairs = db->n_event_align_pairs;
    v.sum_bases = db->sum_bases;
    v.rawptr = raw; v.offset = off; v.range = rng; v.digitisation = dig;
    v.base_to_event_map = (abea_index_pair_t**)db->base_to_event_map; v.events_per_base = db->events_per_base;
    v.read_stat_flag = db->read_stat_flag; v.n_event_alignment = db->n_event_alignment;
    g_core.rna = (core->opt.flag & F5C_RNA) != 0;
    abea_f5c_process(&g_core, &v);                   // db->et[i], event_align_pairs[i], base_to_event_map[i] are malloc()ed
    core->event_time += g_core.event_time;           //   inside, exactly where event_single / scaling_single malloc them
    g_core.event_time = 0;
    free(raw); free(off); free(rng); free(dig); free(nsample);
}
