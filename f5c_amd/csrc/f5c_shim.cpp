/* f5c_shim.cpp — align_db()-shaped host code over the C ABI; see include/abea_f5c_shim.h.
 * Pure C++ (no HIP): everything device-side stays behind abea_*.  */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/abea_f5c_shim.h"

/* reference convention: message + exit (src/error.h:44-56) */
#define SHIM_DIE(what) do { fprintf(stderr, "[%s::ERROR]\033[1;31m %s: %s\033[0m\n", __func__, what, abea_last_error()); \
                            exit(EXIT_FAILURE); } while (0)

extern "C" void abea_f5c_init(abea_f5c_core* core) {
    abea_cfg cfg;
    cfg.device_id = core->cuda_dev_id;
    cfg.kmer_size = core->kmer_size;
    cfg.model = core->model;
    cfg.mem_frac = core->cuda_mem_frac;
    cfg.max_arena_bytes = 0;
    cfg.verbosity = core->verbosity;
    cfg.reserved = 0;
    abea_ctx* ctx = nullptr;
    if (core->n_cuda_devs > 1 && core->cuda_dev_ids) {
        if (abea_init_multi(&ctx, &cfg, core->cuda_dev_ids, core->n_cuda_devs) != ABEA_OK) SHIM_DIE("abea_init_multi");
    } else if (abea_init(&ctx, &cfg) != ABEA_OK) SHIM_DIE("abea_init");
    core->cuda = ctx;
    if (core->inflight > 0 && abea_set_inflight(ctx, core->inflight) != ABEA_OK) SHIM_DIE("abea_set_inflight");
    core->event_time = 0;
    core->align_kernel_time = core->align_pre_kernel_time = core->align_core_kernel_time = 0;
    core->align_post_kernel_time = core->align_cuda_memcpy = core->align_cuda_preprocess = 0;
    core->align_cuda_postprocess = core->align_cuda_total_kernel = 0;
}

/* what a submitted batch keeps alive until its wait: the pointer / count arrays the host batch points into */
struct shim_pending {
    std::vector<const abea_event_t*> ev;
    std::vector<uint64_t> n_ev;
    abea_f5c_db* db = nullptr;
    int32_t ticket = -1;
    bool scale = false;
};

static void shim_fill(abea_f5c_core* core, abea_f5c_db* db, bool scale, const char* who, std::vector<const abea_event_t*>& ev,
                      std::vector<uint64_t>& n_ev, abea_host_batch& hb);
static void shim_finish(abea_f5c_core* core, abea_f5c_db* db, bool scale, const char* who);

static void shim_run(abea_f5c_core* core, abea_f5c_db* db, bool scale, const char* who) {
    std::vector<const abea_event_t*> ev;
    std::vector<uint64_t> n_ev;
    abea_host_batch hb;
    shim_fill(core, db, scale, who, ev, n_ev, hb);
    if (abea_align_batch_host((abea_ctx*)core->cuda, &hb) != ABEA_OK) {
        fprintf(stderr, "[%s::ERROR]\033[1;31m abea_align_batch_host: %s\033[0m\n", who, abea_last_error());
        exit(EXIT_FAILURE);
    }
    shim_finish(core, db, scale, who);
}

static void shim_fill(abea_f5c_core* core, abea_f5c_db* db, bool scale, const char* who, std::vector<const abea_event_t*>& ev,
                      std::vector<uint64_t>& n_ev, abea_host_batch& hb) {
    const int32_t n = db->n_bam_rec;
    ev.resize((size_t)n); n_ev.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) { ev[(size_t)i] = db->et[i].event; n_ev[(size_t)i] = db->et[i].n; }
    memset(&hb, 0, sizeof hb);
    hb.n_reads = n;
    hb.read = db->read;
    hb.read_len = db->read_len;
    hb.events = ev.data();
    hb.n_events = n_ev.data();
    hb.scalings = db->scalings;
    hb.n_samples = db->nsample;
    hb.pairs = db->event_align_pairs;
    hb.n_pairs = db->n_event_align_pairs;
    hb.diag = nullptr;
    if (scale) {
        /* scaling_single malloc()s base_to_event_map[i] for aligned reads (f5c.c:746) and leaves NULL otherwise (f5c.c:787): the
         * library does the same from its un-flatten workers (ABEA_HB_MALLOC_MAPS; rounds 2-4 malloc()ed one per read here, on the
         * caller's thread, and released the failed ones afterwards) */
        hb.flags |= ABEA_HB_MALLOC_MAPS;
        hb.base_to_event_map = db->base_to_event_map;
        hb.scalings_out = db->scalings;              /* recalibrated in place, like recalibrate_model(&db->scalings[i]) */
        hb.events_per_base = db->events_per_base;
        hb.read_stat_flag = db->read_stat_flag;
        hb.n_event_alignment = db->n_event_alignment;
        hb.min_num_events_to_rescale = core->min_num_events_to_rescale;
    }
}

static void shim_finish(abea_f5c_core* core, abea_f5c_db* db, bool scale, const char* who) {
    (void)db; (void)scale;
    abea_stats st;
    abea_get_stats((abea_ctx*)core->cuda, &st);
    /* the copies overlap the kernels and the host loops (h2d_ms / d2h_ms stay 0); flatten = the reference's
     * "cpu preprocess", un-flatten = its "cpu postprocess" (meth_main.c:769,784) */
    core->align_pre_kernel_time += st.pre_ms * 1e-3;
    core->align_core_kernel_time += st.fill_ms * 1e-3;      /* fused fill + traceback */
    core->align_post_kernel_time += st.trace_ms * 1e-3;     /* scaling_single kernel, when fused */
    core->align_kernel_time += (st.pre_ms + st.fill_ms + st.trace_ms) * 1e-3;
    core->align_cuda_total_kernel += (st.pre_ms + st.fill_ms + st.trace_ms) * 1e-3;
    core->align_cuda_memcpy += (st.h2d_ms + st.d2h_ms) * 1e-3;
    core->align_cuda_preprocess += st.flatten_ms * 1e-3;
    core->align_cuda_postprocess += st.unflatten_ms * 1e-3;
    if (core->verbosity > 1)                                  /* f5c.cu:1052 "Load : CPU x entries, GPU y entries" */
        fprintf(stderr, "[%s] Load : CPU 0 entries (0.0M bases), GPU %lld entries (%.1fM bases) on %d device(s), %lld skipped by guards\n",
                who, (long long)st.n_reads_gpu, (double)db->sum_bases / 1e6, st.n_devices, (long long)st.n_reads_skipped);
}

extern "C" void abea_f5c_align(abea_f5c_core* core, abea_f5c_db* db) { shim_run(core, db, false, __func__); }

extern "C" void abea_f5c_align_scale(abea_f5c_core* core, abea_f5c_db* db) {
    if (!db->base_to_event_map || !db->events_per_base || !db->read_stat_flag || !db->n_event_alignment) {
        fprintf(stderr, "[%s::ERROR] the db view lacks the scaling_single outputs\n", __func__);
        exit(EXIT_FAILURE);
    }
    shim_run(core, db, true, __func__);
}

/* Two process_db batches in flight (src/meth_main.c:668-689 overlaps only I/O with processing): submit starts the batch
 * on a free lane of the context and returns; wait blocks until db's outputs are complete.  db (and everything it points
 * to) must stay untouched in between.  At most ABEA_MAX_INFLIGHT handles outstanding; same results as abea_f5c_align. */
extern "C" void* abea_f5c_align_submit(abea_f5c_core* core, abea_f5c_db* db) {
    shim_pending* p = new shim_pending();
    p->db = db;
    abea_host_batch hb;
    shim_fill(core, db, false, __func__, p->ev, p->n_ev, hb);
    if (abea_align_batch_host_submit((abea_ctx*)core->cuda, &hb, &p->ticket) != ABEA_OK) SHIM_DIE("abea_align_batch_host_submit");
    return p;
}

extern "C" void abea_f5c_align_wait(abea_f5c_core* core, void* handle) {
    shim_pending* p = (shim_pending*)handle;
    if (!p) return;
    if (abea_align_batch_host_wait((abea_ctx*)core->cuda, p->ticket) != ABEA_OK) SHIM_DIE("abea_align_batch_host_wait");
    shim_finish(core, p->db, false, __func__);
    delete p;
}

static void shim_need_signal(const abea_f5c_db* db, const char* who) {
    if (!db->rawptr || !db->nsample || !db->offset || !db->range || !db->digitisation) {
        fprintf(stderr, "[%s::ERROR] the db view lacks the signal_t fields\n", who);
        exit(EXIT_FAILURE);
    }
}

extern "C" void abea_f5c_event_db(abea_f5c_core* core, abea_f5c_db* db) {
    shim_need_signal(db, __func__);
    const int32_t n = db->n_bam_rec;
    std::vector<abea_event_t*> ev((size_t)n);
    std::vector<uint64_t> n_ev((size_t)n);
    abea_events_host_batch eb;
    memset(&eb, 0, sizeof eb);
    eb.n_reads = n; eb.rawptr = db->rawptr; eb.n_samples = db->nsample; eb.offset = db->offset; eb.range = db->range;
    eb.digitisation = db->digitisation; eb.read = db->read; eb.read_len = db->read_len; eb.rna = core->rna;
    eb.signal_to_pa_in_place = 1;
    eb.events = ev.data(); eb.n_events = n_ev.data(); eb.scalings = db->scalings;
    if (abea_events_batch_host((abea_ctx*)core->cuda, &eb) != ABEA_OK) SHIM_DIE("abea_events_batch_host");
    for (int32_t i = 0; i < n; ++i) {
        db->et[i].event = ev[(size_t)i]; db->et[i].n = (size_t)n_ev[(size_t)i];
        db->et[i].start = 0; db->et[i].end = (size_t)n_ev[(size_t)i];           /* events.c:576-580 */
        if (db->event_align_pairs) {                                            /* f5c.c:722-731 */
            db->event_align_pairs[i] = db->nsample[i] > 0 ? (abea_pair_t*)malloc(sizeof(abea_pair_t) * ((size_t)n_ev[(size_t)i] + (size_t)db->read_len[i])) : nullptr;
            if (db->nsample[i] > 0 && !db->event_align_pairs[i]) { fprintf(stderr, "[%s::ERROR] malloc failed\n", __func__); exit(EXIT_FAILURE); }
        }
    }
    abea_stats st;
    abea_get_stats((abea_ctx*)core->cuda, &st);
    core->event_time += st.total_ms * 1e-3;
}

extern "C" void abea_f5c_process(abea_f5c_core* core, abea_f5c_db* db) {
    shim_need_signal(db, __func__);
    if (!db->base_to_event_map || !db->events_per_base || !db->read_stat_flag || !db->n_event_alignment) {
        fprintf(stderr, "[%s::ERROR] the db view lacks the scaling_single outputs\n", __func__);
        exit(EXIT_FAILURE);
    }
    const int32_t n = db->n_bam_rec;
    std::vector<abea_event_t*> ev((size_t)n);
    std::vector<uint64_t> n_ev((size_t)n);
    abea_process_batch pb;
    memset(&pb, 0, sizeof pb);
    pb.n_reads = n; pb.rawptr = db->rawptr; pb.n_samples = db->nsample; pb.offset = db->offset; pb.range = db->range;
    pb.digitisation = db->digitisation; pb.read = db->read; pb.read_len = db->read_len; pb.rna = core->rna;
    pb.signal_to_pa_in_place = 1;
    pb.events = ev.data(); pb.n_events = n_ev.data(); pb.scalings = db->scalings;
    pb.pairs = db->event_align_pairs; pb.n_pairs = db->n_event_align_pairs;
    pb.base_to_event_map = db->base_to_event_map; pb.events_per_base = db->events_per_base;
    pb.read_stat_flag = db->read_stat_flag; pb.n_event_alignment = db->n_event_alignment;
    pb.min_num_events_to_rescale = core->min_num_events_to_rescale;
    if (abea_process_batch_host((abea_ctx*)core->cuda, &pb) != ABEA_OK) SHIM_DIE("abea_process_batch_host");
    for (int32_t i = 0; i < n; ++i) {
        db->et[i].event = ev[(size_t)i]; db->et[i].n = (size_t)n_ev[(size_t)i];
        db->et[i].start = 0; db->et[i].end = (size_t)n_ev[(size_t)i];
    }
    abea_stats st;
    abea_get_stats((abea_ctx*)core->cuda, &st);
    core->event_time += st.event_ms * 1e-3;
    shim_finish(core, db, false, __func__);              /* the maps of failed reads were released by the library */
}

extern "C" void abea_f5c_free(abea_f5c_core* core) {
    abea_free((abea_ctx*)core->cuda);
    core->cuda = nullptr;
}
