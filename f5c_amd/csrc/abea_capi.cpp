/* abea_capi.cpp — host side of libabea_hip.so (include/abea.h).
 *
 * Plays the role of the reference's init_cuda / align_cuda / free_cuda (src/f5c.cu) for the
 * MI355X path: owns a one-shot device arena, builds per-read descriptors (per-read log
 * constants via glibc), orders reads longest-first so the hardware workgroup
 * dispatcher load-balances 1..50 kb mixes, splits a batch into arena-sized sub-batches and
 * launches the kernels of abea_kernels.hip (align-pre, the fused fill + traceback + expansion kernel, optionally
 * scaling_single) on the library's own stream, timed with HIP events on that stream.  The host-buffer entry cuts the
 * batch into chunks that rotate through eight slot streams so that host copies, PCIe and kernels overlap; the raw-signal
 * entry runs event detection (row N2).  No CPU alignment fallback exists in this library.
 */
#include <numeric>
#include "abea_internal.h"

static_assert(sizeof(abea_event_t) == 24, "event_t layout (f5c.h:129)");
static_assert(sizeof(abea_model_t) == 12, "model_t layout (f5c.h:147)");
static_assert(sizeof(abea_scalings_t) == 16, "scalings_t layout (f5c.h:158)");
static_assert(sizeof(abea_pair_t) == 8, "AlignedPair layout (f5c.h:181)");
static_assert(sizeof(abea_index_pair_t) == 8, "index_pair_t layout (f5c.h:187)");
static_assert(sizeof(abea_read_diag) == 40, "abea_read_diag layout");
static_assert(sizeof(abea_kpar_t) == 16, "kpar layout");
static_assert(sizeof(abea_read_desc) % 16 == 0, "desc alignment");

extern "C" {
__global__ void abea_ev_sums_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                    const int64_t*, double*, double*, const int32_t*);
__global__ void abea_ev_psum_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                    const int64_t*, const int32_t*, double*, uint32_t*);
__global__ void abea_ev_pscan_kernel(int, const int32_t*, const int32_t*, const int64_t*, double*, const uint32_t*, int32_t*);
__global__ void abea_ev_pwrite_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                      const int64_t*, const int64_t*, const int32_t*, const double*, const int32_t*,
                                      double*, double*);
__global__ void abea_ev_tstat_kernel(int, const int32_t*, const int32_t*, const int64_t*, const int32_t*, const double*,
                                     const double*, float*, float*, int, const int32_t*);
__global__ void abea_ev_detect_kernel(int, const int32_t*, const int32_t*, const int64_t*, const float*, const float*,
                                      const int64_t*, const int32_t*, int32_t*, int32_t*, const int32_t*, int);
__global__ void abea_ev_spec_kernel(int, const int32_t*, const int32_t*, const int64_t*, const float*, const float*,
                                    const int64_t*, const int32_t*, uint16_t*, int32_t*, int);
__global__ void abea_ev_fix_kernel(int, const int32_t*, const int32_t*, const int64_t*, const float*, const float*,
                                   const int64_t*, const int32_t*, int32_t*, int32_t*, int32_t*, int);
__global__ void abea_ev_scan_kernel(int, const int32_t*, const int32_t*, const int64_t*, int32_t*, int32_t*);
__global__ void abea_ev_gather_kernel(int, const int32_t*, const int32_t*, const int64_t*, const int32_t*,
                                      const uint16_t*, const int32_t*, const int32_t*, const int64_t*, const int32_t*,
                                      int32_t*);
__global__ void abea_ev_scan2_kernel(int, const int32_t*, const int32_t*, const int64_t*, int32_t*, const uint16_t*, const int32_t*,
                                     const uint32_t*, int32_t*, int32_t*);
__global__ void abea_ev_create_kernel(int, const int32_t*, const int32_t*, const int64_t*, const double*, const double*,
                                      const int64_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                                      abea_event_t*, const int64_t*, float*, int, const int32_t*);
__global__ void abea_ev_spec2_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                     const int64_t*, const int32_t*, uint16_t*, int32_t*, uint32_t*);
__global__ void abea_ev_spec2_rna_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                         const int64_t*, const int32_t*, uint16_t*, int32_t*, uint32_t*);
__global__ void abea_ev_fix2_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                    const int64_t*, const int32_t*, int32_t*, int32_t*, int32_t*, int);
__global__ void abea_ev_create3_kernel(int, const int32_t*, const int16_t*, const int64_t*, const int32_t*, const float*,
                                       const int64_t*, const int32_t*, const uint16_t*, const int32_t*, const int64_t*,
                                       const int32_t*, const int32_t*, const int32_t*, abea_event_t*, const int64_t*, float*,
                                       const int32_t*, int);
__global__ void abea_ev_scalings_kernel(int, const int32_t*, const int64_t*, const int32_t*, const float*, const int32_t*,
                                        const int32_t*, const char*, const int64_t*, const int32_t*, const abea_model_t*, int,
                                        abea_scalings_t*, float*, const int64_t*, const int32_t*);
}

/* ------------------------------------------------------------------ errors */
static thread_local char g_err[512] = "";
int abea_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
extern "C" const char* abea_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ context */
/* every failure after `new abea_ctx` goes through abea_free, which releases whatever was created so far */
#define INIT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { abea_free(c); \
    return abea_fail(ABEA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

extern "C" int abea_init(abea_ctx** out, const abea_cfg* cfg) {
    if (!out || !cfg || !cfg->model) return abea_fail(ABEA_EINVAL, "abea_init: null argument");
    if (cfg->kmer_size < 1 || cfg->kmer_size > ABEA_MAX_KMER_SIZE)
        return abea_fail(ABEA_EINVAL, "abea_init: kmer_size %u outside [1,%d]", cfg->kmer_size, ABEA_MAX_KMER_SIZE);
    /* The host entry keeps up to 8 chunks in flight on as many streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
     * (default 4) and kernels of streams that share a queue run one after the other (measured: 422 -> 370 ms per 100 k reads
     * with 16 queues, profiles/r05/hw_queues_ab.txt).  The runtime reads the variable when it initialises, i.e. at the first HIP
     * call of the process: if that is the call below (f5c: init_cuda is the first thing that touches the device) this takes
     * effect; a process that has already used HIP must export it itself.  Never overrides the caller's value, and
     * ABEA_KEEP_HW_QUEUES=1 (any value) keeps the library's hands off the environment altogether: setenv is not safe against other
     * threads calling getenv/setenv at the same moment and changes the runtime's configuration for every HIP user of the process —
     * a host application that runs threads before init_cuda's replacement should export the variable itself and set the opt-out. */
    if (!getenv("ABEA_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return abea_fail(ABEA_ENODEV, "abea_init: no HIP device visible (this library has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= n_dev)
        return abea_fail(ABEA_EINVAL, "abea_init: device_id %d but %d device(s)", cfg->device_id, n_dev);
    HIP_TRY(hipSetDevice(cfg->device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return abea_fail(ABEA_ENODEV, "abea_init: device %d is %s; the kernels are built for gfx950 only",
                    cfg->device_id, prop.gcnArchName);
    abea_ctx* c = new abea_ctx();
    memset(&c->stats, 0, sizeof c->stats);
    c->device = cfg->device_id;
    c->n_cu = prop.multiProcessorCount;
    c->numa_node = abea_device_numa_node(cfg->device_id);
    snprintf(c->arch, sizeof c->arch, "%s", prop.gcnArchName);
    c->k = cfg->kmer_size;
    c->verbosity = cfg->verbosity;
    INIT_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& e : c->ev) INIT_TRY(hipEventCreate(&e));
    const size_t n_model = (size_t)1 << (2 * cfg->kmer_size);
    INIT_TRY(hipMalloc(&c->d_model, n_model * sizeof(abea_model_t)));
    INIT_TRY(hipMemcpy(c->d_model, cfg->model, n_model * sizeof(abea_model_t), hipMemcpyHostToDevice));
    {   /* the model-only terms of recalibrate_model's normal equations (align.c:697-706), once per model entry, with the
         * reference's own double expressions (this file is built -ffp-contract=off; IEEE division on both sides) */
        std::vector<double> mt(n_model * 3);
        for (size_t r = 0; r < n_model; ++r) {
            const double level_mean = cfg->model[r].level_mean, level_stdv = cfg->model[r].level_stdv;
            const double inv_var = 1. / (level_stdv * level_stdv);
            const double mu = level_mean;
            mt[3 * r] = inv_var; mt[3 * r + 1] = mu * inv_var; mt[3 * r + 2] = mu * mu * inv_var;
        }
        INIT_TRY(hipMalloc(&c->d_mterms, mt.size() * sizeof(double)));
        INIT_TRY(hipMemcpy(c->d_mterms, mt.data(), mt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    /* one-shot arena (f5c.cu:110-199 sizes its arrays once from free memory x MEM_FACTOR) */
    size_t free_b = 0, total_b = 0;
    INIT_TRY(hipMemGetInfo(&free_b, &total_b));
    double frac = (cfg->mem_frac > 0.f && cfg->mem_frac <= 1.f) ? cfg->mem_frac : 0.9;
    size_t want = (size_t)((double)free_b * frac);
    if (cfg->max_arena_bytes && want > cfg->max_arena_bytes) want = (size_t)cfg->max_arena_bytes;
    want = want / 4096 * 4096;
    if (want < ((size_t)16 << 20)) { abea_free(c); return abea_fail(ABEA_ENOMEM, "abea_init: only %zu bytes free on device", free_b); }
    INIT_TRY(hipMalloc(&c->arena, want));
    c->arena_bytes = want;
    if (c->verbosity > 0)
        fprintf(stderr, "[abea_init] device %d: %s, %d CUs, arena %.2f GiB of %.2f GiB free\n", c->device, c->arch, c->n_cu,
                want / 1073741824.0, free_b / 1073741824.0);
    *out = c;
    return ABEA_OK;
}

/* One f5c process driving several GPUs (north_star: batches shard across the GPUs of a node behind align_db): a parent
 * context owning one child context per listed device.  The host entry splits each batch over the children
 * (abea_host.cpp); the device-resident entries need a single-device context.  A device may be listed more than once
 * (two contexts on one GPU: how the dispatch is tested on a 1-GPU box). */
extern "C" int abea_init_multi(abea_ctx** out, const abea_cfg* cfg, const int32_t* device_ids, int32_t n_devices) {
    if (!out || !cfg || !device_ids || n_devices < 1) return abea_fail(ABEA_EINVAL, "abea_init_multi: bad argument");
    if (n_devices == 1) { abea_cfg one = *cfg; one.device_id = device_ids[0]; return abea_init(out, &one); }
    abea_ctx* parent = new abea_ctx();
    memset(&parent->stats, 0, sizeof parent->stats);
    parent->device = -1;
    parent->k = cfg->kmer_size;
    parent->verbosity = cfg->verbosity;
    for (int32_t i = 0; i < n_devices; ++i) {
        abea_cfg one = *cfg;
        one.device_id = device_ids[i];
        /* contexts sharing a device share its free memory: split the arena fraction between them */
        int32_t share = 0;
        for (int32_t j = 0; j < n_devices; ++j) share += device_ids[j] == device_ids[i];
        const float frac = (cfg->mem_frac > 0.f && cfg->mem_frac <= 1.f) ? cfg->mem_frac : 0.9f;
        int32_t left = 0;                       /* contexts on this device still to be created, this one included */
        for (int32_t j = i; j < n_devices; ++j) left += device_ids[j] == device_ids[i];
        one.mem_frac = share > 1 ? frac / (float)left : frac;     /* each takes 1/left of what is still free */
        abea_ctx* child = nullptr;
        const int rc = abea_init(&child, &one);
        if (rc != ABEA_OK) { abea_free(parent); return rc; }
        parent->children.push_back(child);
    }
    snprintf(parent->arch, sizeof parent->arch, "%s", parent->children[0]->arch);
    parent->n_cu = parent->children[0]->n_cu;
    for (abea_ctx* ch : parent->children) parent->arena_bytes += ch->arena_bytes;
    *out = parent;
    return ABEA_OK;
}

extern "C" int32_t abea_device_count(abea_ctx* c) { return !c ? 0 : c->children.empty() ? 1 : (int32_t)c->children.size(); }

extern "C" void abea_free(abea_ctx* c) {
    if (!c) return;
    abea_host_join_async(c);                    /* submitted batches still running use the children: let them finish first */
    for (abea_ctx* ch : c->children) abea_free(ch);
    c->children.clear();
    if (c->device >= 0) {
        hipSetDevice(c->device);
        if (c->stream) hipStreamSynchronize(c->stream);
        abea_host_release(c);
        abea_chain_release(c);
        abea_hmm_release(c);
        hipFree(c->d_model); hipFree(c->d_mterms); hipFree(c->arena);
        hipHostFree(c->h_desc);
        for (auto& e : c->ev) if (e) hipEventDestroy(e);
        if (c->stream) hipStreamDestroy(c->stream);
    } else {
        abea_host_release(c);
    }
    delete c;
}

extern "C" int abea_device_info(abea_ctx* c, char* arch, size_t arch_len, int32_t* n_cu, uint64_t* arena_bytes) {
    if (!c) return abea_fail(ABEA_EINVAL, "null ctx");
    if (arch && arch_len) snprintf(arch, arch_len, "%s", c->arch);
    if (n_cu) *n_cu = c->n_cu;
    if (arena_bytes) *arena_bytes = c->arena_bytes;
    return ABEA_OK;
}

extern "C" int abea_get_stats(abea_ctx* c, abea_stats* out) {
    if (!c || !out) return abea_fail(ABEA_EINVAL, "null argument");
    *out = c->stats;
    return ABEA_OK;
}

extern "C" int abea_selftest(abea_ctx* c) {
    if (!c) return abea_fail(ABEA_EINVAL, "null ctx");
    if (!c->children.empty()) { for (abea_ctx* ch : c->children) { int rc = abea_selftest(ch); if (rc) return rc; } return ABEA_OK; }
    ABEA_API_ENTER(c, "abea_selftest");
    HIP_TRY(hipSetDevice(c->device));
    int* d = (int*)c->arena;
    int h[320];
    hipLaunchKernelGGL(abea_selftest_kernel, dim3(1), dim3(64), 0, c->stream, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int l = 0; l < 64; ++l) {
        if (h[l] != l - 1) return abea_fail(ABEA_EHIP, "selftest: wave_shr:1 lane %d got %d", l, h[l]);
        if (h[64 + l] != (l == 63 ? -1 : l + 1)) return abea_fail(ABEA_EHIP, "selftest: wave_shl:1 lane %d got %d", l, h[64 + l]);
        if (h[128 + l] != 147) return abea_fail(ABEA_EHIP, "selftest: readlane lane %d got %d", l, h[128 + l]);
        if (h[192 + l] != (l == 49 ? 777 : l)) return abea_fail(ABEA_EHIP, "selftest: writelane lane %d got %d", l, h[192 + l]);
        if (h[256 + l] != 1) return abea_fail(ABEA_EHIP, "selftest: fp64-reciprocal quotient differs from a/b at lane %d", l);
    }
    return ABEA_OK;
}

/* ------------------------------------------------------------------ batch planning */
plan_offsets plan_advance(const plan_read& r, sub_layout& lay, abea_stats& st) {
    plan_offsets o = {0, 0, 0, 0};
    if (!r.run) return o;
    const int64_t n_groups = (r.n_bands + ABEA_GROUP - 1) / ABEA_GROUP;
    o.kpar_off = (int64_t)lay.n_kpar;   lay.n_kpar += (size_t)r.K;
    o.evm_off = (int64_t)lay.n_evm;     lay.n_evm += align_up((size_t)r.E + 64, 4);
    o.code_off = (int64_t)lay.n_code;   lay.n_code += align_up((size_t)(r.E + r.K) / 16 + 2, 4);
    o.trace_off = (int64_t)lay.n_trace; lay.n_trace += (size_t)n_groups * 64;
    st.sum_events += r.E; st.sum_bands += r.n_bands;
    /* SURVEY §8d algorithmic bytes; P is added after the run from n_pairs */
    const uint64_t Bn = (uint64_t)r.n_bands;
    st.bytes_ref += 24ull * r.E + (uint64_t)(r.L + 1) + 40 + 108ull * Bn + 4;
    st.bytes_min += 4ull * r.E + (uint64_t)(r.L + 3) / 4 + 40 + 25ull * Bn + (Bn + 7) / 8 + 4;
    /* this implementation: pre (24E + L + 12K model gather -> 16K + 4E), fill (16K + 4E in, 32 B/band trace out),
     * post (trace blocks on the path ~ 32 B/band worst case, codes, 16K+4E again, 8P out) */
    st.bytes_moved += 24ull * r.E + r.L + 12ull * r.K + 2 * (16ull * r.K + 4ull * r.E) + 64ull * Bn +
                      16ull * r.K + 4ull * r.E;
    return o;
}

void plan_desc_fill(abea_read_desc& d, const plan_read& r, const abea_scalings_t& sc, const plan_offsets& o) {
    memset(&d, 0, sizeof d);
    d.out_idx = r.idx;
    d.read_len = r.L; d.n_events = r.E; d.n_kmers = r.K;
    if (!r.run) { d.n_groups = 0; return; }
    d.n_groups = (int32_t)((r.n_bands + ABEA_GROUP - 1) / ABEA_GROUP);
    d.scale = sc.scale; d.shift = sc.shift;
    d.kpar_off = o.kpar_off; d.evm_off = o.evm_off; d.code_off = o.code_off; d.trace_off = o.trace_off;
}

void plan_desc_layout(abea_read_desc& d, const plan_read& r, const abea_scalings_t& sc, sub_layout& lay, abea_stats& st) {
    plan_desc_fill(d, r, sc, plan_advance(r, lay, st));
}

void plan_desc_consts(abea_read_desc& d) {
    if (d.n_groups == 0) return;
    /* align.c:207-216, doubles, glibc */
    volatile double eps = 1e-10, trim_p = 0.01;
    double events_per_kmer = (double)(size_t)d.n_events / (size_t)d.n_kmers;
    double p_stay = 1 - (1 / (events_per_kmer + 1));
    d.lp_skip = log(eps);
    d.lp_stay = log(p_stay);
    d.lp_step = log(1.0 - exp(d.lp_skip) - exp(d.lp_stay));
    d.lp_trim = log(trim_p);
}

int ensure_pinned(void** p, size_t* cap, size_t need) {
    if (*cap >= need) return ABEA_OK;
    if (*p) hipHostFree(*p);
    *p = nullptr; *cap = 0;
    size_t n = align_up(need + need / 4, 4096);
    if (hipHostMalloc(p, n, hipHostMallocDefault) != hipSuccess) return abea_fail(ABEA_EHIP, "hipHostMalloc(%zu) failed", n);
    *cap = n;
    return ABEA_OK;
}

extern "C" int abea_align_batch_device(abea_ctx* c, const abea_device_batch* B) {
    if (!c || !B) return abea_fail(ABEA_EINVAL, "null argument");
    if (!c->children.empty()) return abea_fail(ABEA_EINVAL, "abea_align_batch_device needs a single-device context");
    ABEA_API_ENTER(c, "abea_align_batch_device");
    const int32_t n = B->n_reads;
    if (n < 0) return abea_fail(ABEA_EINVAL, "n_reads < 0");
    const double t_start = abea_now_ms();
    abea_stats st; memset(&st, 0, sizeof st);
    st.arena_bytes = c->arena_bytes;
    if (n == 0) { st.total_ms = 0; c->stats = st; return ABEA_OK; }
    if (!B->read_ptr || !B->read_len || !B->event_ptr || !B->n_events || !B->pair_ptr || !B->scalings ||
        !B->reads || !B->events || !B->pairs || !B->n_pairs)
        return abea_fail(ABEA_EINVAL, "abea_align_batch_device: null array");
    if (B->base_to_event_map && (!B->kmer_ptr || !B->scalings_io || !B->events_per_base || !B->read_stat_flag ||
                                 !B->n_event_alignment))
        return abea_fail(ABEA_EINVAL, "abea_align_batch_device: scaling outputs requested but some are null");
    HIP_TRY(hipSetDevice(c->device));      /* the caller's thread changes per batch (f5c.cu:692-694) */

    /* ---- guards + ordering ---- */
    std::vector<plan_read> reads((size_t)n);
    std::vector<int32_t> order; order.reserve((size_t)n);
    std::vector<int32_t> skipped;
    for (int32_t i = 0; i < n; ++i) {
        plan_read& r = reads[(size_t)i];
        r = make_plan(i, B->read_len[i], B->n_events[i], c->k);
        if (r.run && r.n_bands > ABEA_MAX_BANDS)
            return abea_fail(ABEA_EINVAL, "read %d has %lld bands; the limit is %lld", i, (long long)r.n_bands, (long long)ABEA_MAX_BANDS);
        if (r.run) order.push_back(i); else skipped.push_back(i);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return reads[(size_t)a].n_bands > reads[(size_t)b].n_bands; });
    st.n_reads_skipped = (int64_t)skipped.size();
    st.n_reads_gpu = (int64_t)order.size();

    /* skipped reads are appended last: they ride along in the last sub-batch as n_groups == 0 descriptors */
    std::vector<int32_t> seq; seq.reserve((size_t)n);
    seq.insert(seq.end(), order.begin(), order.end());
    seq.insert(seq.end(), skipped.begin(), skipped.end());

    size_t pos = 0;
    while (pos < seq.size()) {
        /* ---- carve a sub-batch that fits the arena ---- */
        size_t end = pos, bytes = 4096;
        while (end < seq.size()) {
            const plan_read& r = reads[(size_t)seq[end]];
            const size_t need = (r.run ? scratch_bytes(r) : sizeof(abea_read_desc)) + 4;
            if (bytes + need + 65536 + 256 > c->arena_bytes) break;
            bytes += need; ++end;
        }
        if (end == pos)
            return abea_fail(ABEA_ENOMEM, "read %d (L=%d, E=%d) needs more scratch than the %zu-byte arena",
                        seq[pos], reads[(size_t)seq[pos]].L, reads[(size_t)seq[pos]].E, c->arena_bytes);
        const size_t m = end - pos;
        int rc = ensure_pinned((void**)&c->h_desc, &c->h_desc_cap, m * sizeof(abea_read_desc));
        if (rc) return rc;

        /* ---- arena layout: [desc][kpar][evm][codes][trace] ---- */
        sub_layout lay;
        for (size_t j = 0; j < m; ++j) {
            const plan_read& r = reads[(size_t)seq[pos + j]];
            abea_read_desc& d = c->h_desc[j];
            plan_desc_layout(d, r, B->scalings[r.idx], lay, st);
            d.read_off = B->read_ptr[r.idx]; d.event_off = B->event_ptr[r.idx]; d.pair_off = B->pair_ptr[r.idx];
            d.kmer_off = B->kmer_ptr ? B->kmer_ptr[r.idx] : 0;
        }
        /* the per-read log-probabilities (four libm calls each: 25 ms for 100 k reads on one thread) on the worker pool */
        abea_parallel_for(c, (int64_t)m, 512, [&](int64_t lo, int64_t hi) { for (int64_t j = lo; j < hi; ++j) plan_desc_consts(c->h_desc[j]); });
        const size_t n_kpar = lay.n_kpar, n_evm = lay.n_evm, n_code = lay.n_code, n_trace = lay.n_trace;
        uint8_t* p = c->arena;
        abea_read_desc* d_desc = (abea_read_desc*)p;        p += align_up(m * sizeof(abea_read_desc), 256);
        abea_kpar_t* d_kpar = (abea_kpar_t*)p;              p += align_up(n_kpar * sizeof(abea_kpar_t), 256);
        uint32_t* d_krank = (uint32_t*)p;                   p += align_up(n_kpar * 4, 256);     /* k-mer ranks for phase 4 (fused scaling only) */
        float* d_evm = (float*)p;                           p += align_up(n_evm * 4 + 512, 256);
        uint32_t* d_codes = (uint32_t*)p;                   p += align_up(n_code * 4, 256);
        uint4* d_trace = (uint4*)p;                         p += n_trace * sizeof(uint4);
        if ((size_t)(p - c->arena) > c->arena_bytes)
            return abea_fail(ABEA_ENOMEM, "internal: sub-batch layout %zu exceeds arena %zu", (size_t)(p - c->arena), c->arena_bytes);

        HIP_TRY(hipMemcpyAsync(d_desc, c->h_desc, m * sizeof(abea_read_desc), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        hipLaunchKernelGGL(abea_pre_kernel, dim3((unsigned)m), dim3(256), 0, c->stream,
                           d_desc, B->reads, B->events, c->d_model, (int)c->k, d_kpar, d_evm,
                           B->base_to_event_map ? d_krank : (uint32_t*)nullptr);
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        abea_fused_scaling fs;                               /* row N1: scaling_single as the last phase of the kernel */
        memset(&fs, 0, sizeof fs);
        if (B->base_to_event_map) {
            fs.reads = B->reads; fs.model = c->d_model; fs.b2e = B->base_to_event_map; fs.sc_io = B->scalings_io;
            fs.epb = B->events_per_base; fs.flag_io = B->read_stat_flag; fs.nalign = B->n_event_alignment;
            fs.kmer_size = (int32_t)c->k;
            fs.min_rescale = B->min_num_events_to_rescale > 0 ? B->min_num_events_to_rescale : 200;
            fs.krank = d_krank; fs.mterms = c->d_mterms;
        }
        hipLaunchKernelGGL(abea_align_kernel, dim3((unsigned)m), dim3(64), 0, c->stream,
                           d_desc, d_evm, d_kpar, d_trace, d_codes, B->pairs, B->n_pairs, B->diag,
                           (unsigned long long*)nullptr, (int64_t*)nullptr, fs);
        HIP_TRY(hipEventRecord(c->ev[2], c->stream));
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));            /* h_desc and the arena are reused by the next sub-batch */
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); st.pre_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); st.fill_ms += ms;   /* fused fill + traceback (+ scaling_single) */
        st.n_sub_batches += 1; st.fill_launches += 1;
        pos = end;
    }
    st.total_ms = abea_now_ms() - t_start;
    c->stats = st;
    if (c->verbosity > 1)
        fprintf(stderr, "[abea] %lld reads on GPU, %lld skipped, %lld sub-batch(es): pre %.3f ms fill %.3f ms post %.3f ms\n",
                (long long)st.n_reads_gpu, (long long)st.n_reads_skipped, (long long)st.n_sub_batches,
                st.pre_ms, st.fill_ms, st.trace_ms);
    return ABEA_OK;
}

/* ------------------------------------------------------------------ raw signal -> events (row N2) */
extern "C" int abea_detect_events_device(abea_ctx* c, const abea_signal_batch* B) {
    if (!c || !B) return abea_fail(ABEA_EINVAL, "null argument");
    if (!c->children.empty()) return abea_fail(ABEA_EINVAL, "abea_detect_events_device needs a single-device context");
    ABEA_API_ENTER(c, "abea_detect_events_device");
    return abea_detect_events_locked(c, B);
}

/* scratch of ONE pass of abea_detect_events_on over these reads (the arithmetic of its wave carving: 64 reads per wave in
 * descending sample count, every array interleaved over the wave's lanes and as long as its longest read) */
size_t abea_detect_scratch_bytes(const int32_t* n_samples, const int32_t* event_cap, const int32_t* n_kmers, int32_t n) {
    if (n <= 0) return 0;
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n_samples[a] > n_samples[b]; });
    const size_t N = (size_t)n, n_waves = (N + 63) / 64;
    const size_t EV_SEG = 512, EV_FIXCAP = 48, SEG_BYTES = 64 * (EV_SEG * 2 + EV_FIXCAP * 4 + 12 * 4 + 2 * 8 + 4 * 4);
    size_t bytes = N * (4 + 8 + 4 + 12 + 8 + 4 + 8 + 4 + 4 + 4) + n_waves * 56 + 8192 + 4096 + 8192;
    for (size_t w = 0; w < n_waves; ++w) {
        const int32_t len = n_samples[order[w * 64]] + 1;
        int32_t cap = 1, wk = 1;
        for (size_t q = w * 64; q < std::min(N, (w + 1) * 64); ++q) {
            cap = std::max(cap, event_cap[order[q]]);
            if (n_kmers) wk = std::max(wk, n_kmers[order[q]]);
        }
        const size_t nseg = std::max<size_t>(1, ((size_t)std::max(len - 1, 0) + EV_SEG - 1) / EV_SEG);
        bytes += (size_t)std::max(len, 1) * 64 * 24 + (size_t)cap * 64 * 8 + (size_t)wk * 64 * 4 + nseg * SEG_BYTES;
    }
    return bytes + (1u << 20);
}

/* the entry's body; the caller holds the context (abea_process.cpp chains it behind a host-side flatten) */
int abea_detect_events_locked(abea_ctx* c, const abea_signal_batch* B) {
    abea_ev_exec X;
    X.stream = c->stream; X.scratch = c->arena; X.scratch_bytes = c->arena_bytes;
    X.h_pinned = (void**)&c->h_desc; X.h_cap = &c->h_desc_cap; X.async = false; X.e0 = c->ev[0]; X.e1 = c->ev[1];
    return abea_detect_events_on(c, B, X);
}

/* The detector's kernels for one signal batch on X.stream with scratch [X.scratch, +X.scratch_bytes) and the pinned index
 * staging *X.h_pinned.  Synchronous form (X.async == false): sub-batches of waves as the scratch allows, a stream
 * synchronisation after each, kernel time into c->stats.event_ms.  Asynchronous form (the chunk pipeline of
 * abea_process.cpp): everything must fit the scratch at once (ABEA_ENOMEM otherwise), nothing is waited for, and the
 * pinned staging must stay untouched until the stream has consumed it. */
int abea_detect_events_on(abea_ctx* c, const abea_signal_batch* B, const abea_ev_exec& X) {
    const int32_t n = B->n_reads;
    if (n < 0) return abea_fail(ABEA_EINVAL, "n_reads < 0");
    if (n == 0) return ABEA_OK;
    if (!B->sig_ptr || !B->n_samples || !B->scaling || !B->event_ptr || !B->event_cap || !B->signal || !B->events ||
        !B->n_events)
        return abea_fail(ABEA_EINVAL, "abea_detect_events_device: null array");
    if (B->scalings && (!B->reads || !B->read_ptr || !B->read_len))
        return abea_fail(ABEA_EINVAL, "abea_detect_events_device: scalings need the read sequences");
    HIP_TRY(hipSetDevice(c->device));
    const int rna = B->rna ? 1 : 0;            /* getevents(nsample, rawptr, rna), events.c:562-582; F5C_RNA, f5c.c:698-702 */
    /* lane-per-read passes: order reads by length so that the 64 reads of a wavefront finish together */
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return B->n_samples[a] > B->n_samples[b]; });
    const size_t N = (size_t)n;
    const int n_waves_all = (n + 63) / 64;
    if (!X.async) c->stats.event_ms = 0;
    int w0 = 0;
    while (w0 < n_waves_all) {
        /* ---- carve waves whose interleaved scratch fits the arena: per sample S,Q fp64 + two float t-statistics
         *      (24 B: touched by flagged reads only since the fused common path), per event slot a peak position + a mean
         *      (8 B) ---- */
        const size_t idx_bytes = N * (4 + 8 + 4 + 12 + 8 + 4 + 8 + 4 + 4 + 4) + (size_t)n_waves_all * 56 + 8192;
        if (idx_bytes + (1u << 20) > X.scratch_bytes) return abea_fail(ABEA_ENOMEM, "arena too small for %d index records", n);
        const size_t budget = X.scratch_bytes - idx_bytes - 4096;
        std::vector<int64_t> wave_base, peak_base, kmer_base, seg_base; std::vector<int32_t> wave_len, wave_cap, wave_k, wave_nseg;
        size_t entries = 0, pentries = 0, kentries = 0, segs = 0;
        const size_t EV_SEG = 512, EV_FIXCAP = 48, SEG_BYTES = 64 * (EV_SEG * 2 + EV_FIXCAP * 4 + 12 * 4 + 2 * 8 + 4 * 4);   /* abea_kernels.hip */
        int w1 = w0;
        while (w1 < n_waves_all) {
            const int32_t len = B->n_samples[order[(size_t)w1 * 64]] + 1;      /* longest read of the wave, +1 for S[n] */
            int32_t cap = 1, wk = 1;
            for (int q = w1 * 64; q < std::min(n, (w1 + 1) * 64); ++q) {
                cap = std::max(cap, B->event_cap[order[(size_t)q]]);
                if (B->scalings) wk = std::max(wk, B->read_len[order[(size_t)q]] - (int32_t)c->k + 1);
            }
            const size_t need = (size_t)std::max(len, 1) * 64, pneed = (size_t)cap * 64, kneed = (size_t)wk * 64;
            const size_t nseg = std::max<size_t>(1, ((size_t)std::max(len - 1, 0) + EV_SEG - 1) / EV_SEG);
            if ((entries + need) * 24 + (pentries + pneed) * 8 + (kentries + kneed) * 4 + (segs + nseg) * SEG_BYTES + 4096 > budget) break;
            seg_base.push_back((int64_t)segs); wave_nseg.push_back((int32_t)nseg); segs += nseg;
            wave_base.push_back((int64_t)entries); wave_len.push_back(len);
            peak_base.push_back((int64_t)pentries); wave_cap.push_back(cap);
            kmer_base.push_back((int64_t)kentries); wave_k.push_back(wk);
            entries += need; pentries += pneed; kentries += kneed; ++w1;
        }
        if (w1 == w0) return abea_fail(ABEA_ENOMEM, "a read of %d samples does not fit the %zu-byte arena",
                                  B->n_samples[order[(size_t)w0 * 64]], X.scratch_bytes);
        const int nw = w1 - w0;
        const int r0 = w0 * 64, nr = std::min(n - r0, nw * 64);
        int rc = ensure_pinned(X.h_pinned, X.h_cap, idx_bytes);
        if (rc) return rc;
        uint8_t* h = (uint8_t*)*X.h_pinned; uint8_t* d = X.scratch;
        size_t o = 0;
        auto put = [&](const void* src, size_t sz) { size_t at = o; if (src) memcpy(h + o, src, sz); else memset(h + o, 0, sz);
                                                     o = align_up(o + sz, 16); return at; };
        const size_t o_order = put(order.data() + r0, (size_t)nr * 4), o_sig = put(B->sig_ptr, N * 8);
        const size_t o_ns = put(B->n_samples, N * 4), o_sc = put(B->scaling, N * 12), o_ep = put(B->event_ptr, N * 8);
        const size_t o_ec = put(B->event_cap, N * 4), o_rp = put(B->read_ptr, N * 8), o_rl = put(B->read_len, N * 4);
        const size_t o_wb = put(wave_base.data(), (size_t)nw * 8), o_wl = put(wave_len.data(), (size_t)nw * 4);
        const size_t o_pb = put(peak_base.data(), (size_t)nw * 8), o_wc = put(wave_cap.data(), (size_t)nw * 4);
        const size_t o_kb = put(kmer_base.data(), (size_t)nw * 8), o_wk = put(wave_k.data(), (size_t)nw * 4);   /* the scalings kernel's rows of k-mer levels */
        const size_t o_sb = put(seg_base.data(), (size_t)nw * 8), o_wn = put(wave_nseg.data(), (size_t)nw * 4);
        const size_t o_need = put(nullptr, N * 4);                    /* per-read "segments never met" flags, zeroed */
        const size_t o_need_s = put(nullptr, N * 4);                  /* per-read "prefix sums may round" flags */
        double* dS = (double*)(d + align_up(o, 256));
        double* dQ = dS + entries;                 /* back to back: the kernels address [dS, dQ + entries) as ONE array of {S, Q} pairs */
        float* dT1 = (float*)(dQ + entries);
        float* dT2 = dT1 + entries;
        int32_t* dPk = (int32_t*)(dT2 + entries);
        float* dMean = (float*)(dPk + pentries);
        /* kentries floats behind the means: a row of k-mer levels per read, written and read back by the scalings kernel */
        uint8_t* dSegs = (uint8_t*)(dMean + pentries + kentries);
        dSegs += (256 - ((uintptr_t)dSegs & 255)) & 255;
        uint16_t* dSpec = (uint16_t*)dSegs;
        int32_t* dFix = (int32_t*)(dSpec + segs * EV_SEG * 64);
        int32_t* dRec = dFix + segs * EV_FIXCAP * 64;
        double* dSegSum = (double*)(dRec + segs * 12 * 64);
        uint32_t* dSegExp = (uint32_t*)(dSegSum + segs * 2 * 64);
        HIP_TRY(hipMemcpyAsync(d, h, o, hipMemcpyHostToDevice, X.stream));
        if (X.e0) HIP_TRY(hipEventRecord(X.e0, X.stream));
        /* pass 1: prefix sums by segments where that is provably exact, sequentially otherwise */
        const int max_nseg = *std::max_element(wave_nseg.begin(), wave_nseg.end());
        const dim3 sgrid((unsigned)((max_nseg + 3) / 4), (unsigned)nw);
        const bool seq_only = getenv("ABEA_EV_SEQUENTIAL") != nullptr;
        const char* path_env = getenv("ABEA_EV_PATH");
        const bool arrays = seq_only || (path_env && !strcmp(path_env, "arrays"));     /* rounds 3-5: prefix-sum / t-statistic arrays for every read */
        const int32_t* dOrder = (const int32_t*)(d + o_order); const int64_t* dSig = (const int64_t*)(d + o_sig);
        const int32_t* dNs = (const int32_t*)(d + o_ns); const float* dSc = (const float*)(d + o_sc);
        const int64_t* dWb = (const int64_t*)(d + o_wb); const int32_t* dWl = (const int32_t*)(d + o_wl);
        const int64_t* dPb = (const int64_t*)(d + o_pb); const int32_t* dWc = (const int32_t*)(d + o_wc);
        const int64_t* dSb = (const int64_t*)(d + o_sb); const int32_t* dWn = (const int32_t*)(d + o_wn);
        const int32_t* dEc = (const int32_t*)(d + o_ec); const int64_t* dEp = (const int64_t*)(d + o_ep);
        int32_t* dNeed = (int32_t*)(d + o_need); int32_t* dNeedS = (int32_t*)(d + o_need_s);
        const unsigned tiles = (unsigned)std::min<int64_t>(1024, (wave_len[0] + 3) / 4);   /* grid-stride: any tile count covers the wave */
        const unsigned etiles = (unsigned)std::min<int64_t>(256, (wave_cap[0] + 3) / 4);
        const dim3 fgrid((unsigned)((std::max(max_nseg, 2) - 1 + 3) / 4), (unsigned)nw);
        if (arrays) {
            /* ---- the array form for every read: prefix sums {S, Q} and the two t-statistics go through HBM ---- */
            if (!seq_only) {
                hipLaunchKernelGGL(abea_ev_psum_kernel, sgrid, dim3(256), 0, X.stream, nr, dOrder, B->signal, dSig, dNs, dSc, dSb, dWn,
                                   dSegSum, dSegExp);
                hipLaunchKernelGGL(abea_ev_pscan_kernel, dim3((unsigned)nw), dim3(64), 0, X.stream, nr, dOrder, dNs, dSb, dSegSum,
                                   dSegExp, dNeedS);
                hipLaunchKernelGGL(abea_ev_pwrite_kernel, sgrid, dim3(256), 0, X.stream, nr, dOrder, B->signal, dSig, dNs, dSc, dWb,
                                   dSb, dWn, dSegSum, (const int32_t*)dNeedS, dS, dQ);
            }
            hipLaunchKernelGGL(abea_ev_sums_kernel, dim3((unsigned)nw), dim3(64), 0, X.stream, nr, dOrder, B->signal, dSig, dNs, dSc,
                               dWb, dS, dQ, seq_only ? (const int32_t*)nullptr : (const int32_t*)dNeedS);
            hipLaunchKernelGGL(abea_ev_tstat_kernel, dim3(tiles, (unsigned)nw), dim3(256), 0, X.stream, nr, dOrder, dNs, dWb, dWl,
                               dS, dQ, dT1, dT2, rna, (const int32_t*)nullptr);
            /* pass 3: the automaton over (read, segment) pairs, then the sequential one for reads whose segments never met */
            hipLaunchKernelGGL(abea_ev_spec_kernel, sgrid, dim3(256), 0, X.stream, nr, dOrder, dNs, dWb, dT1, dT2, dSb, dWn, dSpec,
                               dRec, rna);
            if (max_nseg > 1)
                hipLaunchKernelGGL(abea_ev_fix_kernel, fgrid, dim3(256), 0, X.stream, nr, dOrder, dNs, dWb, dT1, dT2, dSb, dWn, dFix,
                                   dRec, dNeed, rna);
            hipLaunchKernelGGL(abea_ev_scan_kernel, dim3((unsigned)nw), dim3(64), 0, X.stream, nr, dOrder, dNs, dSb, dRec,
                               B->n_events);
            hipLaunchKernelGGL(abea_ev_gather_kernel, sgrid, dim3(256), 0, X.stream, nr, dOrder, dNs, dSb, dWn, dSpec, dFix, dRec,
                               dPb, dEc, dPk);
            hipLaunchKernelGGL(abea_ev_detect_kernel, dim3((unsigned)nw), dim3(64), 0, X.stream, nr, dOrder, dNs, dWb, dT1, dT2,
                               dPb, dEc, dPk, B->n_events, seq_only ? (const int32_t*)nullptr : (const int32_t*)dNeed, rna);
            hipLaunchKernelGGL(abea_ev_create_kernel, dim3(etiles, (unsigned)nw), dim3(256), 0, X.stream, nr, dOrder, dNs, dWb, dS,
                               dQ, dPb, dWc, dPk, B->n_events, dEc, B->events, dEp, dMean, rna, (const int32_t*)nullptr);
        } else {
            /* ---- the common path straight from the samples (round 6; abea_kernels.hip) ... ---- */
            hipLaunchKernelGGL(rna ? abea_ev_spec2_rna_kernel : abea_ev_spec2_kernel, dim3((unsigned)max_nseg, (unsigned)nw), dim3(64), 0,
                               X.stream, nr, dOrder, B->signal, dSig, dNs, dSc, dSb, dWn, dSpec, dRec, dSegExp);
            if (max_nseg > 1)
                hipLaunchKernelGGL(abea_ev_fix2_kernel, fgrid, dim3(256), 0, X.stream, nr, dOrder, B->signal, dSig, dNs, dSc, dSb,
                                   dWn, dFix, dRec, dNeed, rna);
            hipLaunchKernelGGL(abea_ev_scan2_kernel, dim3((unsigned)nr), dim3(64), 0, X.stream, nr, dOrder, dNs, dSb, dRec,
                               (const uint16_t*)dSpec, (const int32_t*)dFix, (const uint32_t*)dSegExp, B->n_events, dNeed);
            /* ---- ... and the array form behind it for the reads it flagged (sums that may round, a segment that never met its
             *      replay): sequential prefix sums, t-statistics, the sequential automaton, events from the prefix sums.  With no
             *      read flagged these four launches return at once. ---- */
            hipLaunchKernelGGL(abea_ev_sums_kernel, dim3((unsigned)nw), dim3(64), 0, X.stream, nr, dOrder, B->signal, dSig, dNs, dSc,
                               dWb, dS, dQ, (const int32_t*)dNeed);
            hipLaunchKernelGGL(abea_ev_tstat_kernel, dim3(tiles, (unsigned)nw), dim3(256), 0, X.stream, nr, dOrder, dNs, dWb, dWl,
                               dS, dQ, dT1, dT2, rna, (const int32_t*)dNeed);
            hipLaunchKernelGGL(abea_ev_detect_kernel, dim3((unsigned)nw), dim3(64), 0, X.stream, nr, dOrder, dNs, dWb, dT1, dT2,
                               dPb, dEc, dPk, B->n_events, (const int32_t*)dNeed, rna);
            hipLaunchKernelGGL(abea_ev_create_kernel, dim3(etiles, (unsigned)nw), dim3(256), 0, X.stream, nr, dOrder, dNs, dWb, dS,
                               dQ, dPb, dWc, dPk, B->n_events, dEc, B->events, dEp, dMean, rna, (const int32_t*)dNeed);
            hipLaunchKernelGGL(abea_ev_create3_kernel, dim3((unsigned)nr, (unsigned)((max_nseg + 7) / 8)), dim3(64), 0, X.stream, nr,
                               dOrder, B->signal, dSig, dNs, dSc, dSb, (const int32_t*)dRec, (const uint16_t*)dSpec,
                               (const int32_t*)dFix, dPb, dWc, (const int32_t*)B->n_events, dEc, B->events, dEp, dMean,
                               (const int32_t*)dNeed, rna);
        }
        if (B->scalings)                                     /* one wavefront per read (round 6): grid = reads of this pass */
            hipLaunchKernelGGL(abea_ev_scalings_kernel, dim3((unsigned)nr), dim3(64), 0, X.stream,
                               nr, (const int32_t*)(d + o_order), (const int64_t*)(d + o_pb), (const int32_t*)(d + o_wc), dMean,
                               B->n_events, (const int32_t*)(d + o_ec), B->reads, (const int64_t*)(d + o_rp),
                               (const int32_t*)(d + o_rl), c->d_model, (int)c->k, B->scalings, dMean + pentries,
                               (const int64_t*)(d + o_kb), (const int32_t*)(d + o_wk));
        if (X.e1) HIP_TRY(hipEventRecord(X.e1, X.stream));
        HIP_TRY(hipGetLastError());
        if (X.async) {
            if (w1 < n_waves_all) return abea_fail(ABEA_ENOMEM, "internal: the detector's scratch for %d reads does not fit %zu bytes at once", n, X.scratch_bytes);
            return ABEA_OK;
        }
        HIP_TRY(hipStreamSynchronize(X.stream));
        float ms = 0;
        if (X.e0 && X.e1) { HIP_TRY(hipEventElapsedTime(&ms, X.e0, X.e1)); c->stats.event_ms += ms; }
        w0 = w1;
    }
    return ABEA_OK;
}
