/* abea_hmm.cpp — host driver of row N4: batches of profile_hmm_score() calls (reference src/hmm.c:689-735, call site
 * meth.c:473-474: two per CpG group and read) scored on the GPU.  Builds the job descriptors (per-job transition
 * log-probabilities exactly as calculate_transitions does, hmm.c:240-310), flattens the event windows and sequences
 * into pinned memory, runs abea_hmm_forward_kernel and copies the scores back.  No CPU scoring fallback. */
#include <atomic>
#include <cmath>
#include <numeric>
#include "abea_internal.h"
#include "abea_hmm.h"

extern "C" __global__ void abea_hmm_forward_kernel(int, int, int, const abea_hmm_job*, const char*, const float*,
                                                   const abea_model_t*, int, const float*, const float*, float*, float*);

struct abea_hmm_state {                    /* per context, created on first use */
    float* d_tbl = nullptr;                /* p7_FLogsum table (logsum.h:33-48), built with glibc on the host */
    abea_model_t* d_model = nullptr; size_t model_entries = 0;
    uint8_t* pin = nullptr; size_t pin_cap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

void abea_hmm_release(abea_ctx* c) {
    abea_hmm_state* s = c->hmm;
    if (!s) return;
    hipFree(s->d_tbl); hipFree(s->d_model); hipHostFree(s->pin);
    if (s->e0) hipEventDestroy(s->e0);
    if (s->e1) hipEventDestroy(s->e1);
    delete s;
    c->hmm = nullptr;
}

extern "C" __global__ void abea_hmm_gather_kernel(int, const abea_hmm_job*, const int64_t*, const int32_t*, float*);

static int hmm_score_batch(abea_ctx* c, const abea_hmm_job_t* jobs, int32_t n_jobs, const abea_model_t* cpgmodel,
                           uint32_t kmer_size, float* scores, bool events_on_device);

extern "C" int abea_hmm_score_batch_host(abea_ctx* c, const abea_hmm_job_t* jobs, int32_t n_jobs,
                                         const abea_model_t* cpgmodel, uint32_t kmer_size, float* scores) {
    return hmm_score_batch(c, jobs, n_jobs, cpgmodel, kmer_size, scores, false);
}

/* the same with every job's `events` a DEVICE pointer (the read's table as the chain left it in HBM: events + event_ptr[i] of
 * abea_detect_events_device / abea_device_batch): the event windows are gathered by a kernel instead of the host */
extern "C" int abea_hmm_score_batch_device(abea_ctx* c, const abea_hmm_job_t* jobs, int32_t n_jobs,
                                           const abea_model_t* cpgmodel, uint32_t kmer_size, float* scores) {
    return hmm_score_batch(c, jobs, n_jobs, cpgmodel, kmer_size, scores, true);
}

static int hmm_score_batch(abea_ctx* c, const abea_hmm_job_t* jobs, int32_t n_jobs, const abea_model_t* cpgmodel,
                           uint32_t kmer_size, float* scores, bool events_on_device) {
    if (!c || n_jobs < 0 || (n_jobs && (!jobs || !scores)) || !cpgmodel) return abea_fail(ABEA_EINVAL, "abea_hmm_score_batch_host: null argument");
    if (!c->children.empty()) return abea_fail(ABEA_EINVAL, "abea_hmm_score_batch_host needs a single-device context");
    if (kmer_size < 1 || kmer_size > ABEA_MAX_KMER_SIZE) return abea_fail(ABEA_EINVAL, "kmer_size %u", kmer_size);
    if (n_jobs == 0) return ABEA_OK;
    ABEA_API_ENTER(c, "abea_hmm_score_batch_host");
    const double t_start = abea_now_ms();
    HIP_TRY(hipSetDevice(c->device));
    if (!c->hmm) {
        /* built aside and published only when complete: a failure half-way must not leave a state the next call
         * would take for initialised (round-2 advisor finding) */
        c->hmm = new abea_hmm_state();
        struct undo { abea_ctx* c; bool ok = false; ~undo() { if (!ok) abea_hmm_release(c); } } guard{c};
        std::vector<float> tbl(ABEA_HMM_TBL);
        for (int i = 0; i < ABEA_HMM_TBL; i++) tbl[(size_t)i] = (float)log(1. + exp((double)-i / 1000.f));   /* logsum.h:44 */
        HIP_TRY(hipMalloc(&c->hmm->d_tbl, ABEA_HMM_TBL * sizeof(float)));
        HIP_TRY(hipMemcpy(c->hmm->d_tbl, tbl.data(), ABEA_HMM_TBL * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipEventCreate(&c->hmm->e0)); HIP_TRY(hipEventCreate(&c->hmm->e1));
        guard.ok = true;
    }
    abea_hmm_state* S = c->hmm;
    size_t n_model = 1;
    for (uint32_t i = 0; i < kmer_size; ++i) n_model *= 5;            /* MAX_NUM_KMER_METH: alphabet A,C,G,M,T */
    if (S->model_entries != n_model) {
        hipFree(S->d_model); S->d_model = nullptr; S->model_entries = 0;
        HIP_TRY(hipMalloc(&S->d_model, n_model * sizeof(abea_model_t)));
        S->model_entries = n_model;
    }
    HIP_TRY(hipMemcpyAsync(S->d_model, cpgmodel, n_model * sizeof(abea_model_t), hipMemcpyHostToDevice, c->stream));

    /* ---- jobs with at most 16 k-mers first (four to a wavefront), then the wide ones ---- */
    std::vector<int32_t> order((size_t)n_jobs);
    std::vector<int32_t> seq_len((size_t)n_jobs), n_ev((size_t)n_jobs);
    size_t tot_ev = 0, tot_seq = 0, tot_col = 0;
    int32_t max_ev = 0, n16 = 0;
    std::atomic<int32_t> bad_job{-1};
    abea_parallel_for(c, n_jobs, 2048, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; ++j) {
            const abea_hmm_job_t& J = jobs[j];
            seq_len[(size_t)j] = -1; n_ev[(size_t)j] = 0;
            if (!J.m_seq || !J.m_rc_seq || !J.events) { bad_job.store((int32_t)j); continue; }
            const size_t L = strlen(J.m_seq);
            if (L < kmer_size || L > (1u << 20) ||
                (J.rc && J.event_stride != -1) || (!J.rc && J.event_stride != 1) ||                                 /* hmm.c:331 assert */
                (J.event_stride == 1 && J.event_stop_idx < J.event_start_idx) ||      /* the window would be walked out of the */
                (J.event_stride == -1 && J.event_stop_idx > J.event_start_idx)) {     /* event table (UB in the reference)     */
                bad_job.store((int32_t)j); continue;
            }
            seq_len[(size_t)j] = (int32_t)L;
            n_ev[(size_t)j] = (int32_t)(J.event_stop_idx > J.event_start_idx ? J.event_stop_idx - J.event_start_idx + 1
                                                                              : J.event_start_idx - J.event_stop_idx + 1);   /* hmm.c:649-654 */
        }
    });
    if (bad_job.load() >= 0)
        return abea_fail(ABEA_EINVAL, "job %d: null pointer, sequence shorter than k or longer than 1 Mbase, rc / event_stride mismatch, or event window against its stride", bad_job.load());
    for (int32_t j = 0; j < n_jobs; ++j) {
        max_ev = std::max(max_ev, n_ev[(size_t)j]);
        n16 += ((size_t)seq_len[(size_t)j] - kmer_size + 1) <= 16;
    }
    {
        int32_t a = 0, b = n16;
        for (int32_t j = 0; j < n_jobs; ++j) {
            if ((size_t)seq_len[(size_t)j] - kmer_size + 1 <= 16) order[(size_t)a++] = j; else order[(size_t)b++] = j;
        }
    }
    std::vector<abea_hmm_job> desc((size_t)n_jobs);
    for (int32_t q = 0; q < n_jobs; ++q) {                           /* offsets: serial prefix sums */
        const int32_t j = order[(size_t)q];
        abea_hmm_job& d = desc[(size_t)q];
        memset(&d, 0, sizeof d);
        d.ev_off = (int64_t)tot_ev;   tot_ev += align_up((size_t)n_ev[(size_t)j], 4);
        d.seq_off = (int32_t)tot_seq; tot_seq += align_up((size_t)seq_len[(size_t)j] + 1, 4);
        d.seq_len = seq_len[(size_t)j]; d.n_events = n_ev[(size_t)j];
        const size_t nk = (size_t)d.seq_len - kmer_size + 1;
        d.col_off = (int64_t)tot_col; if (nk > 64) tot_col += 3 * ((size_t)d.n_events + 1);
        d.out_idx = j;
    }
    if (tot_seq > (size_t)INT32_MAX)                                 /* seq_off is 32-bit in the device descriptor */
        return abea_fail(ABEA_EINVAL, "%d HMM jobs hold %zu bytes of sequence; at most 2 GiB per call", n_jobs, tot_seq);
    abea_parallel_for(c, n_jobs, 2048, [&](int64_t lo, int64_t hi) {
        for (int64_t q = lo; q < hi; ++q) {
            const abea_hmm_job_t& J = jobs[order[(size_t)q]];
            abea_hmm_job& d = desc[(size_t)q];
            d.rc = J.rc ? 1 : 0; d.flags = J.hmm_flags;
            d.scale = J.scaling.scale; d.shift = J.scaling.shift; d.var = J.scaling.var; d.log_var = J.scaling.log_var;
            /* calculate_transitions (hmm.c:240-310); hmm.c is compiled as C++ (Makefile:6), so log() of a float is logf */
            float p_stay = 1 - (1 / J.events_per_base);
            float p_skip = 0.0025, p_bad = 0.001, p_bad_self = p_bad, p_skip_self = 0.3;
            float p_mk = p_skip, p_mb = p_bad, p_mm_self = p_stay, p_mm_next = 1.0f - p_mm_self - p_mk - p_mb;
            float p_bb = p_bad_self, p_bk, p_bm_next, p_bm_self;
            p_bk = p_bm_next = p_bm_self = (1.0f - p_bb) / 3;
            float p_kk = p_skip_self, p_km = 1.0f - p_kk;
            d.lp_mk = logf(p_mk); d.lp_mb = logf(p_mb); d.lp_mm_self = logf(p_mm_self); d.lp_mm_next = logf(p_mm_next);
            d.lp_bb = logf(p_bb); d.lp_bk = logf(p_bk); d.lp_bm_next = logf(p_bm_next); d.lp_bm_self = logf(p_bm_self);
            d.lp_kk = logf(p_kk); d.lp_km = logf(p_km);
        }
    });
    /* pre_flank (hmm.c:188-233); post_flank[i] = pre_flank[n_events-1-i] term by term (hmm.c:141-185) */
    std::vector<float> flank((size_t)max_ev + 1);
    flank[0] = log(1 - 0.5);
    if (max_ev >= 1) flank[1] = log(0.5) + -3.0f + log(1 - 0.9);
    for (size_t i = 2; i < flank.size(); ++i) flank[i] = log(0.9) + -3.0f + flank[i - 1];

    /* ---- staging: [desc][flank][seqs][event windows] up, [scores] down ---- */
    size_t o = 0;
    const size_t o_desc = o;  o = align_up(o + (size_t)n_jobs * sizeof(abea_hmm_job), 256);
    const size_t o_flank = o; o = align_up(o + flank.size() * 4, 256);
    const size_t o_seq = o;   o = align_up(o + tot_seq, 256);
    /* device-resident events: per job the table's device address and {first event, stride} instead of the window itself */
    const size_t o_src = o;   if (events_on_device) o = align_up(o + (size_t)n_jobs * 16, 256);
    const size_t o_ev = o;    if (!events_on_device) o = align_up(o + tot_ev * 4, 256);
    const size_t up_bytes = o;
    if (events_on_device) o = align_up(o + tot_ev * 4, 256);           /* the windows exist on the device only */
    const size_t o_out = o;   o = align_up(o + (size_t)n_jobs * 4, 256);
    const size_t o_col = o;   o = align_up(o + tot_col * 4, 256);
    if (o + 4096 > c->arena_bytes) return abea_fail(ABEA_ENOMEM, "%d HMM jobs need %zu bytes, the arena has %zu", n_jobs, o, c->arena_bytes);
    int rc = ensure_pinned((void**)&S->pin, &S->pin_cap, o_col);
    if (rc) return rc;
    memcpy(S->pin + o_desc, desc.data(), (size_t)n_jobs * sizeof(abea_hmm_job));
    memcpy(S->pin + o_flank, flank.data(), flank.size() * 4);
    abea_parallel_for(c, n_jobs, 1024, [&](int64_t lo, int64_t hi) {
        for (int64_t q = lo; q < hi; ++q) {
            const abea_hmm_job_t& J = jobs[order[(size_t)q]];
            const abea_hmm_job& d = desc[(size_t)q];
            memcpy(S->pin + o_seq + d.seq_off, d.rc ? J.m_rc_seq : J.m_seq, (size_t)d.seq_len + 1);
            if (events_on_device) {
                ((int64_t*)(S->pin + o_src))[q] = (int64_t)(uintptr_t)J.events;
                int32_t* ss = (int32_t*)(S->pin + o_src + (size_t)n_jobs * 8) + 2 * q;
                ss[0] = (int32_t)J.event_start_idx; ss[1] = J.event_stride;
                continue;
            }
            float* w = (float*)(S->pin + o_ev) + d.ev_off;
            for (int32_t r = 0; r < d.n_events; ++r)                   /* event_idx = e_start + (row-1)*stride, hmm.c:432 */
                w[r] = J.events[(int64_t)J.event_start_idx + (int64_t)r * J.event_stride].mean;
        }
    });
    uint8_t* dev = c->arena;
    HIP_TRY(hipMemcpyAsync(dev, S->pin, up_bytes, hipMemcpyHostToDevice, c->stream));
    const int blocks16 = (n16 + 15) / 16, blocks64 = (n_jobs - n16 + 3) / 4;
    HIP_TRY(hipEventRecord(S->e0, c->stream));
    if (events_on_device)
        hipLaunchKernelGGL(abea_hmm_gather_kernel, dim3((unsigned)((n_jobs + 3) / 4)), dim3(256), 0, c->stream,
                           (int)n_jobs, (const abea_hmm_job*)(dev + o_desc), (const int64_t*)(dev + o_src),
                           (const int32_t*)(dev + o_src + (size_t)n_jobs * 8), (float*)(dev + o_ev));
    hipLaunchKernelGGL(abea_hmm_forward_kernel, dim3((unsigned)(blocks16 + blocks64)), dim3(256), 0, c->stream,
                       n16, n_jobs, blocks16, (const abea_hmm_job*)(dev + o_desc), (const char*)(dev + o_seq),
                       (const float*)(dev + o_ev), S->d_model, (int)kmer_size, S->d_tbl, (const float*)(dev + o_flank),
                       (float*)(dev + o_col), (float*)(dev + o_out));
    HIP_TRY(hipEventRecord(S->e1, c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(S->pin + o_out, dev + o_out, (size_t)n_jobs * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(scores, S->pin + o_out, (size_t)n_jobs * 4);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, S->e0, S->e1));
    memset(&c->stats, 0, sizeof c->stats);
    c->stats.hmm_ms = ms;
    c->stats.total_ms = abea_now_ms() - t_start;
    c->stats.h2d_bytes = up_bytes; c->stats.d2h_bytes = (uint64_t)n_jobs * 4;
    c->stats.arena_bytes = c->arena_bytes;
    return ABEA_OK;
}
