/* abea_internal.h — shared between the host translation units of libabea_hip.so (abea_capi.cpp: context, device
 * entry, event detection; abea_host.cpp: the host-buffer pipeline and the multi-device dispatch).  Internal. */
#ifndef ABEA_INTERNAL_H
#define ABEA_INTERNAL_H
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>
#include "abea_device.h"

extern "C" {
__global__ void abea_selftest_kernel(int* out);
__global__ void abea_pre_kernel(const abea_read_desc*, const char*, const abea_event_t*, const abea_model_t*, int,
                                abea_kpar_t*, float*, uint32_t*);
__global__ void abea_align_kernel(const abea_read_desc*, const float*, const abea_kpar_t*, uint4*, uint32_t*,
                                  abea_pair_t*, int32_t*, abea_read_diag*, unsigned long long*, int64_t*, const abea_fused_scaling);
__global__ void abea_copy_out_kernel(const uint4*, uint4*, size_t);
}

/* ------------------------------------------------------------------ errors */
int abea_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return abea_fail(ABEA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static inline double abea_now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* ------------------------------------------------------------------ context */
struct abea_host_pool;      /* abea_host.cpp: persistent host worker threads */
struct abea_host_async;     /* abea_host.cpp: lanes (slot set + arena share + pool) and the batches in flight */
static const int ABEA_MAX_SLOTS = 16;       /* stream slots per device context (8 are used unless ABEA_HOST_SLOTS asks for more) */
struct abea_host_slot;      /* abea_host.cpp: one chunk in flight (stream, pinned staging, arena share) */
struct abea_chain_slot;     /* abea_chain.cpp: one chunk of the raw-signal pipeline in flight */
struct abea_hmm_state;      /* abea_hmm.cpp: log-sum table, CpG model copy, staging of the profile-HMM entry (row N4) */

struct abea_ctx {
    int device = 0;
    int n_cu = 0;
    char arch[64] = {0};
    uint32_t k = 0;
    int verbosity = 0;
    hipStream_t stream = nullptr;
    abea_model_t* d_model = nullptr;
    double* d_mterms = nullptr;              /* 3 doubles per model entry: the model-only terms of recalibrate_model (abea_fused_scaling.mterms) */
    uint8_t* arena = nullptr;      size_t arena_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    /* pinned staging for descriptors (device entry, event detection) */
    abea_read_desc* h_desc = nullptr; size_t h_desc_cap = 0;
    /* host-buffer entry (abea_host.cpp), created on first use */
    std::vector<abea_host_slot*> slots;
    std::vector<abea_chain_slot*> chain_slots;   /* abea_chain.cpp, created on first use */
    std::mutex slots_mu;
    abea_host_async* async = nullptr;
    int numa_node = -1;                         /* of the device (sysfs), -1 unknown */
    /* one caller at a time per context: every public entry that touches the arena / pool / staging holds this */
    std::mutex api_mu;
    abea_hmm_state* hmm = nullptr;
    /* abea_init_multi: a parent context owns one child per device and no device state of its own */
    std::vector<abea_ctx*> children;
    abea_stats stats;
};

void abea_host_join_async(abea_ctx* c);     /* joins the threads of submitted host batches (abea_host.cpp) */
void abea_host_release(abea_ctx* c);       /* frees pool + slots (abea_host.cpp); called by abea_free */
void abea_hmm_release(abea_ctx* c);        /* abea_hmm.cpp */
void abea_chain_release(abea_ctx* c);      /* abea_chain.cpp */

/* what the raw-signal pipeline works on: the fields of abea_events_host_batch / abea_process_batch (include/abea.h), one view
 * for both entries.  align == false: event_db only (scalings = the method-of-moments estimate, may be NULL; read may be NULL
 * with it).  align == true: event_db -> align_db -> scaling_db; every output array below must be present except pairs / diag /
 * scalings_estimated. */
struct abea_chain_job {
    int32_t n_reads;
    float* const* rawptr; const int64_t* n_samples; const float* offset; const float* range; const float* digitisation;
    const char* const* read; const int32_t* read_len;
    int32_t rna, signal_to_pa_in_place;
    abea_event_t** events; uint64_t* n_events;
    abea_scalings_t* scalings; abea_scalings_t* scalings_estimated;
    bool align;
    abea_pair_t** pairs; int32_t* n_pairs; abea_read_diag* diag; abea_index_pair_t** base_to_event_map;
    double* events_per_base; int32_t* read_stat_flag; int32_t* n_event_alignment; int32_t min_num_events_to_rescale;
};
/* the pipeline on ONE device context over reads mine[0..n_mine) (NULL = all), the caller holds the context (abea_chain.cpp) */
int abea_chain_run(abea_ctx* c, const abea_chain_job* J, const int32_t* mine, int32_t n_mine, abea_stats* st_out);
/* run f(lo, hi) over [0, n) in pieces of `grain` items on the context's persistent worker pool (created on first use;
 * the caller's thread takes part) — abea_host.cpp */
void abea_parallel_for(abea_ctx* c, int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& f);
int abea_default_host_threads();
int abea_pool_threads(abea_ctx* c);          /* abea_host.cpp: width of the device context's worker pool (0 before it exists) */
void abea_host_prepare_pools(abea_ctx* c);   /* abea_host.cpp: every device's worker pool at the width of the context's thread plan */
int abea_host_batches_in_flight(abea_ctx* c);   /* abea_host.cpp: submitted and not yet waited for */
/* the single-caller-per-context contract, enforced: a public entry that uses the arena / pool / staging holds api_mu for
 * its duration and refuses to run while submitted host batches are in flight */
#define ABEA_API_ENTER(c, name) std::lock_guard<std::mutex> api_lock_((c)->api_mu); \
    if (abea_host_batches_in_flight(c)) return abea_fail(ABEA_EBUSY, name ": submitted host batches are still in flight")
/* bodies of public entries for callers that already hold the context (abea_process.cpp chains them) */
int abea_detect_events_locked(abea_ctx* c, const abea_signal_batch* B);            /* abea_capi.cpp */
/* where and how the detector's kernels of one signal batch run (abea_capi.cpp: abea_detect_events_on) */
struct abea_ev_exec {
    hipStream_t stream; uint8_t* scratch; size_t scratch_bytes;
    void** h_pinned; size_t* h_cap;        /* pinned staging of the index records (grown on demand) */
    bool async;                            /* true: one pass, no synchronisation (ABEA_ENOMEM when the scratch is too small) */
    hipEvent_t e0, e1;                     /* optional: recorded before / after the kernels */
};
int abea_detect_events_on(abea_ctx* c, const abea_signal_batch* B, const abea_ev_exec& X);
/* scratch bytes abea_detect_events_on needs for reads of these lengths in ONE pass (upper bound; the same arithmetic) */
size_t abea_detect_scratch_bytes(const int32_t* n_samples, const int32_t* event_cap, const int32_t* n_kmers, int32_t n);
int abea_host_batch_locked(abea_ctx* c, const abea_host_batch* H);                 /* abea_host.cpp */
int abea_device_numa_node(int device);      /* abea_host.cpp: sysfs numa_node of a HIP device's PCI function */

/* ------------------------------------------------------------------ batch planning */
struct plan_read {
    int32_t idx;          /* index in the caller's batch */
    int32_t L, E, K;
    int64_t n_bands;
    bool run;
};

/* the kernels address the trace with 32-bit byte offsets (32 B per band): 2^27 bands = a read of ~40 Mbases */
static const int64_t ABEA_MAX_BANDS = (int64_t)1 << 27;

/* the align_single guard (f5c.c:813-814, E/L < 15.0f in float); reads shorter than k are UB in the reference
 * (size_t underflow, align.c:191) and rejected here */
static inline plan_read make_plan(int32_t idx, int32_t L, int32_t E, uint32_t k) {
    plan_read r;
    r.idx = idx; r.L = L; r.E = E;
    r.K = L - (int32_t)k + 1;
    r.run = E > 0 && r.K >= 1 && ((float)E / (float)L) < 15.0f;
    r.n_bands = (int64_t)E + r.K + 2;
    return r;
}

/* per-read scratch bytes (kpar, evm, codes, trace, desc) */
static inline size_t scratch_bytes(const plan_read& r) {
    const size_t n_groups = (size_t)(r.n_bands + ABEA_GROUP - 1) / ABEA_GROUP;
    return align_up((size_t)r.K * (sizeof(abea_kpar_t) + 4), 16) + align_up((size_t)r.E * 4 + 256, 16) +      /* kpar + the rank array behind it */
           align_up(((size_t)(r.E + r.K) / 16 + 2) * 4, 16) + n_groups * 64 * sizeof(uint4) +
           sizeof(abea_read_desc);
}

/* element counts of one launch's scratch arrays */
struct sub_layout { size_t n_kpar = 0, n_evm = 0, n_code = 0, n_trace = 0; };

/* Descriptor of one read and its place in the launch's scratch; the caller sets read_off/event_off/pair_off/kmer_off.
 * plan_desc = plan_desc_layout (offsets and accounting: integer work, serial because every read's offsets follow the
 * previous read's) + plan_desc_consts (the per-read log-probabilities of align.c:207-216: four glibc log/exp calls, the
 * expensive part, independent per read — the host entry computes them inside its parallel flatten loop). */
struct plan_offsets { int64_t kpar_off, evm_off, code_off, trace_off; };
/* plan_desc_layout = plan_advance (the read's place in the scratch arrays + the accounting: the only part that must run in
 * read order) + plan_desc_fill (the descriptor itself: the host entry writes it from its parallel flatten loop, so that the
 * write misses into the pinned staging block are not taken on the caller's serial path) */
plan_offsets plan_advance(const plan_read& r, sub_layout& lay, abea_stats& st);
void plan_desc_fill(abea_read_desc& d, const plan_read& r, const abea_scalings_t& sc, const plan_offsets& o);
void plan_desc_layout(abea_read_desc& d, const plan_read& r, const abea_scalings_t& sc, sub_layout& lay, abea_stats& st);
void plan_desc_consts(abea_read_desc& d);
static inline void plan_desc(abea_read_desc& d, const plan_read& r, const abea_scalings_t& sc, sub_layout& lay, abea_stats& st) {
    plan_desc_layout(d, r, sc, lay, st);
    plan_desc_consts(d);
}

int ensure_pinned(void** p, size_t* cap, size_t need);
/* workgroups of an abea_copy_out_kernel launch that moves a LARGE block across PCIe (the raw-signal pipeline's tables, when
 * ABEA_CHAIN_TABLE_COPY=kernel, or its signals, ABEA_CHAIN_UP=kernel; abea_link_probe's kernel legs).  32 workgroups keep the link
 * as busy as 512 (16 B x 256 lanes x 32 = 128 KiB in flight: profiles/r06/c_chain_movers_copy_kernel_width_ab.log, link probe at 8 /
 * 16 / 64 / 512) and occupy an eighth of the CUs.  ABEA_CHAIN_COPY_BLOCKS overrides. */
static inline int abea_copy_kernel_blocks() {
    const char* e = getenv("ABEA_CHAIN_COPY_BLOCKS");
    return e ? std::max(1, std::min(4096, atoi(e))) : 32;
}

/* scaling_single's last line for a recalibrated read (align.c:758-760, CACHED_LOG): log_var = log(var) in double, glibc's, on the
 * host.  The kernel hands var down as a double, -1.0 = "not recalibrated" (log_var keeps its input value); a NaN var (NaN sum / n_M)
 * is a recalibrated read whose log_var is log(NaN) = NaN in the reference too — `>= 0` would have left the input value there
 * (round-5 advisor finding).  One helper for both retire paths (abea_host.cpp, abea_chain.cpp). */
static inline void abea_apply_log_var(abea_scalings_t& o, double var_f64) {
    if (!(var_f64 < 0.0)) o.log_var = (float)log(var_f64);
}

/* length of the union of [begin, end) intervals (the chunks' kernel spans on the GPU clock) */
static inline double interval_union_ms(std::vector<std::pair<float, float>>& iv) {
    std::sort(iv.begin(), iv.end());
    double total = 0; float lo = 0, hi = -1;
    for (const auto& x : iv) {
        if (hi < lo || x.first > hi) { if (hi > lo) total += hi - lo; lo = x.first; hi = x.second; }
        else hi = std::max(hi, x.second);
    }
    if (hi > lo) total += hi - lo;
    return total;
}

#endif
