/* f5c_shim.cpp — align_db()-shaped host code over the C ABI; see include/abea_f5c_shim.h.
 * Pure C++ (no HIP): everything device-side stays behind abea_*.  */
#include <cstdio>
#include <cstdlib>
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
    if (abea_init(&ctx, &cfg) != ABEA_OK) SHIM_DIE("abea_init");
    core->cuda = ctx;
    core->align_kernel_time = core->align_pre_kernel_time = core->align_core_kernel_time = 0;
    core->align_post_kernel_time = core->align_cuda_memcpy = core->align_cuda_preprocess = 0;
    core->align_cuda_postprocess = core->align_cuda_total_kernel = 0;
}

extern "C" void abea_f5c_align(abea_f5c_core* core, abea_f5c_db* db) {
    const int32_t n = db->n_bam_rec;
    std::vector<const abea_event_t*> ev((size_t)n);
    std::vector<uint64_t> n_ev((size_t)n);
    for (int32_t i = 0; i < n; ++i) { ev[(size_t)i] = db->et[i].event; n_ev[(size_t)i] = db->et[i].n; }
    abea_host_batch hb;
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
    if (abea_align_batch_host((abea_ctx*)core->cuda, &hb) != ABEA_OK) SHIM_DIE("abea_align_batch_host");
    abea_stats st;
    abea_get_stats((abea_ctx*)core->cuda, &st);
    /* the host entry pipelines chunks: align-pre is timed together with the fused kernel (fill_ms), and the copies
     * overlap the kernels and the host loops, so pre / memcpy are reported as 0 and host_ms is split evenly */
    core->align_pre_kernel_time += st.pre_ms * 1e-3;
    core->align_core_kernel_time += st.fill_ms * 1e-3;      /* fused fill + traceback */
    core->align_post_kernel_time += st.trace_ms * 1e-3;
    core->align_kernel_time += (st.pre_ms + st.fill_ms + st.trace_ms) * 1e-3;
    core->align_cuda_total_kernel += (st.pre_ms + st.fill_ms + st.trace_ms) * 1e-3;
    core->align_cuda_memcpy += (st.h2d_ms + st.d2h_ms) * 1e-3;
    core->align_cuda_preprocess += st.host_ms * 0.5e-3;
    core->align_cuda_postprocess += st.host_ms * 0.5e-3;
    if (core->verbosity > 1)                                  /* f5c.cu:1052 "Load : CPU x entries, GPU y entries" */
        fprintf(stderr, "[%s] Load : CPU 0 entries (0.0M bases), GPU %lld entries (%.1fM bases), %lld skipped by guards\n",
                __func__, (long long)st.n_reads_gpu, (double)db->sum_bases / 1e6, (long long)st.n_reads_skipped);
}

extern "C" void abea_f5c_free(abea_f5c_core* core) {
    abea_free((abea_ctx*)core->cuda);
    core->cuda = nullptr;
}
