/* oracle/abea_oracle.c — TEST INFRASTRUCTURE ONLY (see abea_oracle.h).
 *
 * A plain-C restatement of the reference CPU path of f5c's adaptive banded
 * event alignment.  Written from the algorithm description (SURVEY.md §3.2,
 * §9), each function cites the reference lines it follows.  It keeps the
 * reference's cost profile on purpose (per-read malloc, full −inf/0 init pass,
 * scalar cell loop) because bench.py times it as the "port" CPU baseline.
 *
 * Build: gcc -O2 -ffp-contract=off (reference flags Makefile:8 are -O2 with
 * no FMA on x86-64; contraction must stay off so float expressions round
 * exactly like the reference build).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            /* open_memstream */
#endif
#include "abea_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <malloc.h>

#define ORC_BW 100                 /* ALN_BANDWIDTH f5c.h:34 */
#define ORC_FROM_D 0               /* align.c:194-196 */
#define ORC_FROM_U 1
#define ORC_FROM_L 2

/* align.c:19-32 : A0 C1 G2 T3, anything else 0 (the reference also prints a warning) */
static inline uint32_t base_rank(char b) {
    switch (b) {
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default:  return 0;
    }
}

/* align.c:36-47 : first base most significant */
uint32_t orc_kmer_rank(const char* s, uint32_t k) {
    uint32_t r = 0;
    for (uint32_t j = 0; j < k; ++j) r = (r << 2) | base_rank(s[j]);
    return r;
}

/* align.c:58-106 : method-of-moments shift/scale, all accumulations in double, event order */
orc_scalings_t orc_estimate_scalings(const char* seq, int32_t seq_len, const orc_model_t* model,
                                     uint32_t k, const orc_event_t* ev, size_t n_events) {
    orc_scalings_t out;
    memset(&out, 0, sizeof out);
    int32_t n_kmers = seq_len - (int32_t)k + 1;

    double ev_sum = 0.0;
    for (size_t i = 0; i < n_events; ++i) ev_sum += ev[i].mean;

    double km_sum = 0.0, km_sq = 0.0;
    for (int32_t i = 0; i < n_kmers; ++i) {
        double l = model[orc_kmer_rank(seq + i, k)].level_mean;
        km_sum += l;
        km_sq += l * l;
    }
    double shift = ev_sum / n_events - km_sum / n_kmers;

    double ev_sq = 0.0;
    for (size_t i = 0; i < n_events; ++i) {
        ev_sq += (ev[i].mean - shift) * (ev[i].mean - shift);
    }
    double scale = (ev_sq / n_events) / (km_sq / n_kmers);
    out.shift = (float)shift;
    out.scale = (float)scale;
    return out;
}

/* align.c:108-154 : all-float, no FMA; var treated as 1, cached log(stdv) (CACHED_LOG f5c.h:82) */
static inline float emission_lp(float x, float scale, float shift, const orc_model_t* m) {
    float gp_mean = scale * m->level_mean + shift;
    float gp_stdv = m->level_stdv * 1;
    float gp_log_stdv = m->level_log_stdv;
    float log_inv_sqrt_2pi = -0.918938f;
    float a = (x - gp_mean) / gp_stdv;
    return log_inv_sqrt_2pi - gp_log_stdv + (-0.5f * a * a);
}

typedef struct { int e; int k; } ll_t;      /* align.c:265-268 */

/* analysis hook (tools/walk_drift.py): when set, orc_align() copies band_lower_left[b].kmer_idx of every band here */
static __thread int32_t* g_dbg_llk = NULL;
static __thread size_t g_dbg_llk_cap = 0;
void orc_debug_band_llk(int32_t* buf, size_t cap) { g_dbg_llk = buf; g_dbg_llk_cap = cap; }

int32_t orc_align(orc_pair_t* out, const char* seq, int32_t seq_len,
                  const orc_event_t* ev, size_t n_events, const orc_model_t* model,
                  uint32_t k, float scale, float shift, orc_diag_t* diag) {
    const size_t E = n_events;
    const size_t K = (size_t)seq_len - k + 1;                 /* align.c:191 */
    const int W = ORC_BW, HALF = ORC_BW / 2;

    /* align.c:207-216 : transition penalties, double, glibc log/exp */
    double events_per_kmer = (double)E / K;
    double p_stay = 1 - (1 / (events_per_kmer + 1));
    double epsilon = 1e-10;
    double lp_skip = log(epsilon);
    double lp_stay = log(p_stay);
    double lp_step = log(1.0 - exp(lp_skip) - exp(lp_stay));
    double lp_trim = log(0.01);

    const size_t n_bands = (E + 1) + (K + 1);                 /* align.c:219-221 */

    size_t* ranks = (size_t*)malloc(sizeof(size_t) * K);      /* align.c:226-234 */
    for (size_t i = 0; i < K; ++i) ranks[i] = orc_kmer_rank(seq + i, k);

    float*   bands = (float*)malloc(sizeof(float) * n_bands * W);     /* align.c:242-245 */
    uint8_t* trace = (uint8_t*)malloc(sizeof(uint8_t) * n_bands * W);
    ll_t*    ll    = (ll_t*)malloc(sizeof(ll_t) * n_bands);           /* align.c:270-272 */
    for (size_t b = 0; b < n_bands; ++b) {                            /* align.c:247-259 */
        for (int j = 0; j < W; ++j) {
            bands[b * W + j] = -INFINITY;
            trace[b * W + j] = 0;
        }
    }
#define BAND(b, o)  bands[(size_t)(b) * W + (o)]
#define TRACE(b, o) trace[(size_t)(b) * W + (o)]

    /* align.c:277-291 : first two bands */
    ll[0].e = HALF - 1;  ll[0].k = -1 - HALF;
    ll[1].e = ll[0].e + 1;  ll[1].k = ll[0].k;
    BAND(0, (-1) - ll[0].k) = 0.0f;
    {
        int o = ll[1].e - 0;
        BAND(1, o) = lp_trim;
        TRACE(1, o) = ORC_FROM_U;
    }

    /* align.c:300-410 : fill */
    for (size_t b = 2; b < n_bands; ++b) {
        float s_ll = BAND(b - 1, 0), s_ur = BAND(b - 1, W - 1);
        int ll_ob = s_ll == -INFINITY, ur_ob = s_ur == -INFINITY;
        int right = (ll_ob && ur_ob) ? (b % 2 == 1) : (s_ll < s_ur);  /* align.c:309-314 */
        if (right) { ll[b].e = ll[b - 1].e;     ll[b].k = ll[b - 1].k + 1; }
        else       { ll[b].e = ll[b - 1].e + 1; ll[b].k = ll[b - 1].k; }

        int trim_o = (-1) - ll[b].k;                                   /* align.c:324-333 */
        if (trim_o >= 0 && trim_o < W) {
            int64_t e = ll[b].e - trim_o;
            if (e >= 0 && e < (int64_t)E) {
                BAND(b, trim_o) = lp_trim * (e + 1);
                TRACE(b, trim_o) = ORC_FROM_U;
            } else {
                BAND(b, trim_o) = -INFINITY;
            }
        }

        /* align.c:337-346 ; note (int)(K) and (int)(E-1) as the reference's int conversions */
        int kmin = 0 - ll[b].k, kmax = (int)K - ll[b].k;
        int emin = ll[b].e - (int)(E - 1), emax = ll[b].e - (-1);
        int lo = kmin > emin ? kmin : emin;  if (lo < 0) lo = 0;
        int hi = kmax < emax ? kmax : emax;  if (hi > W) hi = W;

        for (int o = lo; o < hi; ++o) {
            int e = ll[b].e - o, kk = ll[b].k + o;
            int o_up   = ll[b - 1].e - (e - 1);                        /* align.c:354-356 */
            int o_left = (kk - 1) - ll[b - 1].k;
            int o_diag = (kk - 1) - ll[b - 2].k;
            float up   = (o_up   >= 0 && o_up   < W) ? BAND(b - 1, o_up)   : -INFINITY;
            float left = (o_left >= 0 && o_left < W) ? BAND(b - 1, o_left) : -INFINITY;
            float diag = (o_diag >= 0 && o_diag < W) ? BAND(b - 2, o_diag) : -INFINITY;

            float lp = emission_lp(ev[e].mean, scale, shift, &model[ranks[kk]]);
            float sd = diag + lp_step + lp;                            /* align.c:382-384 : double sums, one rounding */
            float su = up + lp_stay + lp;
            float sl = left + lp_skip;

            float m = sd;  uint8_t from = ORC_FROM_D;                  /* align.c:386-392 */
            m = su > m ? su : m;
            from = m == su ? ORC_FROM_U : from;
            m = sl > m ? sl : m;
            from = m == sl ? ORC_FROM_L : from;
            BAND(b, o) = m;
            TRACE(b, o) = from;
        }
    }

    /* align.c:424-445 : end-point scan, first strict max */
    float max_score = -INFINITY;
    int cur_e = 0, cur_k = (int)K - 1;
    for (size_t e = 0; e < E; ++e) {
        int b = ((int)e + 1) + (cur_k + 1);
        int o = ll[b].e - (int)e;
        if (o >= 0 && o < W) {
            float s = BAND(b, o) + (E - e) * lp_trim;
            if (s > max_score) { max_score = s; cur_e = (int)e; }
        }
    }
    int best_event = cur_e;

    /* align.c:452-499 : traceback */
    double sum_emission = 0, n_aligned = 0;
    int n_out = 0, gap = 0, max_gap = 0, oob = 0;
    while (cur_k >= 0 && cur_e >= 0) {
        out[n_out].ref_pos = cur_k;
        out[n_out].read_pos = cur_e;
        n_out++;
        float lp = emission_lp(ev[cur_e].mean, scale, shift, &model[orc_kmer_rank(seq + cur_k, k)]);
        sum_emission += lp;
        n_aligned += 1;

        int b = (cur_e + 1) + (cur_k + 1);
        int o = ll[b].e - cur_e;
        /* the reference indexes the flat trace array with an unchecked offset (align.c:177,486);
         * reproduce the flat index when it stays inside the buffer, flag it otherwise */
        int64_t flat = (int64_t)b * W + o;
        if (flat < 0 || flat >= (int64_t)(n_bands * W)) { oob = 1; break; }
        uint8_t from = trace[flat];
        if (from == ORC_FROM_D)      { cur_k -= 1; cur_e -= 1; gap = 0; }
        else if (from == ORC_FROM_U) { cur_e -= 1; gap = 0; }
        else { cur_k -= 1; gap += 1; if (gap > max_gap) max_gap = gap; }
    }

    for (int i = 0, j = n_out - 1; i < j; ++i, --j) {                  /* align.c:503-513 */
        orc_pair_t t = out[i]; out[i] = out[j]; out[j] = t;
    }

    if (g_dbg_llk)
        for (size_t b = 0; b < n_bands && b < g_dbg_llk_cap; ++b) g_dbg_llk[b] = ll[b].k;

    /* align.c:526-543 : QC */
    double avg = sum_emission / n_aligned;
    int spanned = n_out > 0 && out[0].ref_pos == 0 && out[n_out - 1].ref_pos == (int)(K - 1);
    int pre_qc = n_out;
    if (oob || avg < -5.0 || !spanned || max_gap > 50) n_out = 0;

    if (diag) {
        diag->sum_emission = sum_emission;
        diag->n_aligned = pre_qc;
        diag->best_event = best_event;
        diag->max_score = max_score;
        diag->max_gap = max_gap;
        diag->spanned = spanned;
        diag->oob = oob;
        diag->pad = 0;
    }
    free(ranks); free(bands); free(trace); free(ll);
#undef BAND
#undef TRACE
    return n_out;
}

/* f5c.c:811-830 */
int32_t orc_align_single(orc_pair_t* out, const char* seq, int32_t seq_len,
                         const orc_event_t* ev, size_t n_events, int64_t nsample,
                         const orc_model_t* model, uint32_t k, float scale, float shift, orc_diag_t* diag) {
    if (diag) memset(diag, 0, sizeof *diag);
    if (nsample > 0 && (n_events) / (float)(seq_len) < 15.0f && seq_len >= (int32_t)k && n_events > 0) {
        return orc_align(out, seq, seq_len, ev, n_events, model, k, scale, shift, diag);
    }
    return 0;
}

/* Process-level allocator tuning for the CPU baseline (not part of the algorithm): align() mallocs
 * ~12 MB per 8 kb read; with glibc defaults every one is an mmap/munmap + page faults under the
 * process-wide mm lock, which stops scaling beyond ~64 threads.  on=1 keeps the buffers on the
 * per-thread heaps (what MALLOC_MMAP_THRESHOLD_/MALLOC_TRIM_THRESHOLD_ would do for f5c itself). */
void orc_malloc_tuning(int on) {
    if (on) { mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, -1); mallopt(M_TOP_PAD, 64 << 20); }
    else    { mallopt(M_MMAP_THRESHOLD, 128 * 1024); mallopt(M_TRIM_THRESHOLD, 128 * 1024); mallopt(M_TOP_PAD, 0); }
}

/* ---- batch driver: pthread_db-shaped pool (f5c.c:575-679) ---- */
typedef struct {
    int32_t n_reads; const char* reads; const int64_t* read_ptr; const int32_t* read_len;
    const orc_event_t* events; const int64_t* event_ptr; const int32_t* n_events;
    const orc_scalings_t* scalings; const orc_model_t* model; uint32_t k;
    orc_pair_t* pairs; const int64_t* pair_ptr; int32_t* n_pairs; orc_diag_t* diags;
} batch_t;

typedef struct worker_s {
    const batch_t* B;
    int32_t start, end;            /* static block, start advanced atomically (f5c.c:592,612) */
    struct worker_s* all; int32_t n_workers; int32_t id;
} worker_t;

static void do_read(const batch_t* B, int32_t i) {
    B->n_pairs[i] = orc_align_single(B->pairs + B->pair_ptr[i], B->reads + B->read_ptr[i], B->read_len[i],
                                     B->events + B->event_ptr[i], (size_t)B->n_events[i], 1,
                                     B->model, B->k, B->scalings[i].scale, B->scalings[i].shift,
                                     B->diags ? &B->diags[i] : NULL);
}

static void* worker_main(void* arg) {
    worker_t* w = (worker_t*)arg;
    for (;;) {                                   /* own block */
        int32_t i = __sync_fetch_and_add(&w->start, 1);
        if (i >= w->end) break;
        do_read(w->B, i);
    }
    for (;;) {                                   /* steal from the fullest (f5c.c:575-596) */
        int32_t best = -1, left = 0;
        for (int32_t t = 0; t < w->n_workers; ++t) {
            int32_t rem = w->all[t].end - w->all[t].start;
            if (rem > left) { left = rem; best = t; }
        }
        if (best < 0) break;
        int32_t i = __sync_fetch_and_add(&w->all[best].start, 1);
        if (i < w->all[best].end) do_read(w->B, i);
    }
    return NULL;
}

void orc_align_batch(int32_t n_reads, const char* reads, const int64_t* read_ptr, const int32_t* read_len,
                     const orc_event_t* events, const int64_t* event_ptr, const int32_t* n_events,
                     const orc_scalings_t* scalings, const orc_model_t* model, uint32_t k,
                     orc_pair_t* pairs, const int64_t* pair_ptr, int32_t* n_pairs, orc_diag_t* diags,
                     int32_t n_threads) {
    batch_t B = { n_reads, reads, read_ptr, read_len, events, event_ptr, n_events, scalings, model, k,
                  pairs, pair_ptr, n_pairs, diags };
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_reads) n_threads = n_reads > 0 ? n_reads : 1;
    worker_t* ws = (worker_t*)calloc((size_t)n_threads, sizeof(worker_t));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    int32_t step = (n_reads + n_threads - 1) / n_threads;
    for (int32_t t = 0; t < n_threads; ++t) {
        ws[t].B = &B; ws[t].all = ws; ws[t].n_workers = n_threads; ws[t].id = t;
        ws[t].start = t * step < n_reads ? t * step : n_reads;
        ws[t].end = (t + 1) * step < n_reads ? (t + 1) * step : n_reads;
    }
    if (n_threads == 1) { worker_main(&ws[0]); }
    else {
        for (int32_t t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker_main, &ws[t]);
        for (int32_t t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    free(ws); free(th);
}

/* ---- N1: postalign + recalibrate_model + QC flags (align.c:561-773, f5c.c:736-807) ---- */
int32_t orc_scaling_single(const orc_pair_t* pairs, int32_t n_pairs, const char* seq, int32_t seq_len,
                           const orc_event_t* ev, size_t n_events, const orc_model_t* model, uint32_t k,
                           orc_scalings_t* sc, orc_index_pair_t* map, double* events_per_base,
                           int32_t* read_stat_flag) {
    (void)n_events;
    int32_t n_kmers = seq_len - (int32_t)k + 1;
    *events_per_base = 0;
    if (n_pairs <= 0) { *read_stat_flag |= 0x002; return 0; }        /* FAILED_ALIGNMENT f5c.h:67 */

    /* postalign align.c:571-602 */
    for (int32_t i = 0; i < n_kmers; ++i) { map[i].start = -1; map[i].stop = -1; }
    int32_t max_event = 0, min_event = INT32_MAX, prev_event = -1;
    for (int32_t i = 0; i < n_pairs; ++i) {
        int32_t ki = pairs[i].ref_pos, ei = pairs[i].read_pos;
        if (ei != prev_event) {
            if (map[ki].start == -1) map[ki].start = ei;
            map[ki].stop = ei;
        }
        if (ei > max_event) max_event = ei;
        if (ei < min_event) min_event = ei;
        prev_event = ei;
    }
    *events_per_base = (double)(max_event - min_event) / n_kmers;

    /* postalign align.c:606-657 fused with recalibrate_model align.c:677-723:
     * walk k-mers in order, events start..stop, state 'M' when the k-mer rank changed */
    int32_t n_align = 0, n_M = 0, prev_rank = -1;
    double A00 = 0, A01 = 0, A11 = 0, b0 = 0, b1 = 0;
    for (int32_t ki = 0; ki < n_kmers; ++ki) {
        if (map[ki].start == -1) continue;
        for (int32_t ei = map[ki].start; ei <= map[ki].stop; ++ei) {
            int32_t rank = (int32_t)orc_kmer_rank(seq + ki, k);
            if (prev_rank != rank) {
                n_M++;
                double e = ev[ei].mean, mu = model[rank].level_mean, sd = model[rank].level_stdv;
                double inv_var = 1. / (sd * sd);
                A00 += inv_var; A01 += mu * inv_var; A11 += mu * mu * inv_var;
                b0 += e * inv_var; b1 += mu * e * inv_var;
            }
            n_align++;
            prev_rank = rank;
        }
    }
    int calibrated = 0;
    if (n_M >= 200) {                                                 /* f5c.c:1185 min_num_events_to_rescale */
        double A10 = A01;
        double div = A00 * A11 - A01 * A10;
        double x0 = -(A01 * b1 - A11 * b0) / div;
        double x1 = (A00 * b1 - A10 * b0) / div;
        double shift = x0, scale = x1, var = 0.;
        prev_rank = -1;
        for (int32_t ki = 0; ki < n_kmers; ++ki) {                    /* align.c:738-753 */
            if (map[ki].start == -1) continue;
            for (int32_t ei = map[ki].start; ei <= map[ki].stop; ++ei) {
                int32_t rank = (int32_t)orc_kmer_rank(seq + ki, k);
                if (prev_rank != rank) {
                    double e = ev[ei].mean, mu = model[rank].level_mean, sd = model[rank].level_stdv;
                    double yi = (e - shift - scale * mu);
                    var += yi * yi / (sd * sd);
                }
                prev_rank = rank;
            }
        }
        var /= n_M;
        var = sqrt(var);
        sc->shift = shift; sc->scale = scale; sc->var = var; sc->log_var = log(var);
        calibrated = 1;
    }
    if (!calibrated || sc->var > 2.5) { *read_stat_flag |= 0x001; return n_align; }  /* FAILED_CALIBRATION */
    if (*events_per_base > 5.0) { *read_stat_flag |= 0x004; }                         /* FAILED_QUALITY_CHK */
    return n_align;
}


/* ---- N2: event detection, restated from events.c (scrappie-derived t-statistic peak picker).
 * The reference is compiled as C++ (Makefile:6-8), so fabs()/sqrt() on float arguments are the float
 * overloads; written as fabsf/sqrtf here. */
#include <float.h>

void orc_raw_to_pa(float* raw, size_t nsample, float offset, float range, float digitisation) {
    float raw_unit = range / digitisation;                          /* f5c.c:693 */
    for (size_t j = 0; j < nsample; j++) raw[j] = (raw[j] + offset) * raw_unit;
}

/* events.c:324-369 */
static float* ev_tstat(const double* sum, const double* sumsq, size_t n, size_t w) {
    float* t = (float*)calloc(n, sizeof(float));
    const float eta = FLT_MIN;
    const float wf = (float)w;
    if (n < 2 * w || w < 2) return t;
    for (size_t i = 0; i < w; ++i) { t[i] = 0; t[n - i - 1] = 0; }
    for (size_t i = w; i <= n - w; ++i) {
        double sum1 = sum[i], sumsq1 = sumsq[i];
        if (i > w) { sum1 -= sum[i - w]; sumsq1 -= sumsq[i - w]; }
        float sum2 = (float)(sum[i + w] - sum[i]);
        float sumsq2 = (float)(sumsq[i + w] - sumsq[i]);
        float mean1 = sum1 / wf;
        float mean2 = sum2 / wf;
        float combined_var = sumsq1 / wf - mean1 * mean1 + sumsq2 / wf - mean2 * mean2;
        combined_var = fmaxf(combined_var, eta);
        const float delta_mean = mean2 - mean1;
        t[i] = fabsf(delta_mean) / sqrtf(combined_var / wf);
    }
    return t;
}

typedef struct {
    int def_peak_pos; float def_peak_val; const float* signal; size_t signal_length; float threshold;
    size_t window_length; size_t masked_to; int peak_pos; float peak_value; int valid_peak;
} ev_detector;

/* getevents(nsample, rawptr, rna), events.c:562-582: detector parameters by molecule type (events.c:52-65).  The event
 * table comes back in DETECTION order for RNA too; event_single() reverses it after the scalings (f5c.c:711-719):
 * orc_reverse_events.  RNA is UNPINNED: the reference's RNA test sets are downloaded by test/test_eventalign.sh -e, none is
 * in the mount; this is the same code path with the other parameter row. */
size_t orc_getevents_rna(size_t nsample, const float* raw, orc_event_t* out, int rna) {
    const size_t n = nsample;
    const size_t w1 = rna ? 7 : 3, w2 = rna ? 14 : 6;
    const float thr1 = rna ? 2.5f : 1.4f, thr2 = 9.0f;
    if (n == 0) return 0;
    /* events.c:303-313: prefix sums; the square is a float product */
    double* sums = (double*)calloc(n + 1, sizeof(double));
    double* sumsqs = (double*)calloc(n + 1, sizeof(double));
    for (size_t i = 0; i < n; ++i) {
        sums[i + 1] = sums[i] + raw[i];
        sumsqs[i + 1] = sumsqs[i] + raw[i] * raw[i];
    }
    float* t1 = ev_tstat(sums, sumsqs, n, w1);                        /* events.c:52-65 */
    float* t2 = ev_tstat(sums, sumsqs, n, w2);
    ev_detector d[2] = {
        { -1, FLT_MAX, t1, n, thr1, w1, 0, -1, FLT_MAX, 0 },
        { -1, FLT_MAX, t2, n, thr2, w2, 0, -1, FLT_MAX, 0 } };
    const float peak_height = rna ? 1.0f : 0.2f;
    size_t* peaks = (size_t*)calloc(n, sizeof(size_t));
    size_t peak_count = 0;
    for (size_t i = 0; i < n; i++) {                                  /* events.c:380-452 */
        for (unsigned k = 0; k < 2; k++) {
            ev_detector* det = &d[k];
            if (det->masked_to >= i) continue;
            float cur = det->signal[i];
            if (det->peak_pos == det->def_peak_pos) {
                if (cur < det->peak_value) {
                    det->peak_value = cur;
                } else if (cur - det->peak_value > peak_height) {
                    det->peak_value = cur;
                    det->peak_pos = (int)i;
                }
            } else {
                if (cur > det->peak_value) { det->peak_value = cur; det->peak_pos = (int)i; }
                if (k == 0) {
                    if (det->peak_value > det->threshold) {
                        d[1].masked_to = det->peak_pos + det->window_length;
                        d[1].peak_pos = d[1].def_peak_pos;
                        d[1].peak_value = d[1].def_peak_val;
                        d[1].valid_peak = 0;
                    }
                }
                if (det->peak_value - cur > peak_height && det->peak_value > det->threshold) det->valid_peak = 1;
                if (det->valid_peak && (i - det->peak_pos) > det->window_length / 2) {
                    peaks[peak_count++] = det->peak_pos;
                    det->peak_pos = det->def_peak_pos;
                    det->peak_value = cur;
                    det->valid_peak = 0;
                }
            }
        }
    }
    /* events.c:466-513: events between consecutive peaks */
    size_t ne = 1;
    for (size_t i = 0; i < n; ++i) if (peaks[i] > 0 && peaks[i] < n) ne++;
    for (size_t ev = 0; ev < ne; ++ev) {
        size_t start = (ev == 0) ? 0 : peaks[ev - 1];
        size_t end = (ev == ne - 1) ? n : peaks[ev];
        if (ne == 1) { start = 0; end = n; }                          /* no peak: the reference reads peaks[-1] */
        orc_event_t e; memset(&e, 0, sizeof e);
        e.start = (uint64_t)start;
        e.length = (float)(end - start);
        e.mean = (float)(sums[end] - sums[start]) / e.length;
        const float deltasqr = (sumsqs[end] - sumsqs[start]);
        const float var = deltasqr / e.length - e.mean * e.mean;
        e.stdv = sqrtf(fmaxf(var, 0.0f));
        out[ev] = e;
    }
    free(peaks); free(t1); free(t2); free(sums); free(sumsqs);
    return ne;
}

size_t orc_getevents(size_t nsample, const float* raw, orc_event_t* out) { return orc_getevents_rna(nsample, raw, out, 0); }

/* f5c.c:711-719: "If sequencing RNA, reverse the events to be 3'->5'" */
void orc_reverse_events(orc_event_t* events, size_t n_events) {
    for (size_t i = 0; i < n_events / 2; ++i) {
        orc_event_t tmp = events[i];
        events[i] = events[n_events - 1 - i];
        events[n_events - 1 - i] = tmp;
    }
}

/* ------------------------------------------------------------------------------------------------
 * resquiggle output of one read: the per-read body of output_db_rsq(), src/resquiggle.c:319-449, with printf
 * redirected into a memory stream.  fmt 0 = TSV, 1 = PAF.  `map` is reversed in place when rna (resquiggle.c:346-357).
 * scale/shift: the reference prints db->scalings->scale / ->shift (element 0 of the batch array, :443-444).
 * Returns the text length (the text is copied to out, NUL-terminated, if it fits cap) or -1 where the reference would
 * assert / exit. */
long orc_rsq_format(char* out, size_t cap, int fmt, const char* read_id, int32_t read_len, uint32_t kmer_size,
                    orc_index_pair_t* map, const orc_event_t* event, long nsample, float sc_scale, float sc_shift, int rna) {
    char* buf = NULL; size_t blen = 0;
    FILE* fp = open_memstream(&buf, &blen);
    char* ss = (char*)malloc((size_t)read_len * 24 + 64); size_t sl = 0; ss[0] = 0;
    int bad = 0;
    int32_t n_kmers = read_len - (int32_t)kmer_size + 1;
    int64_t signal_start_point = -1, signal_start_point2 = -1, signal_end_point = -1, signal_end_point2 = -1;
    int64_t read_start = -1, read_end = -1;
    int64_t ci = 0, mi = 0, d = 0; int8_t ff = 1; int matches = 0; int64_t count_samples = 0;
    if (rna) {
        for (int j = 0; j < n_kmers / 2; ++j) { orc_index_pair_t tmp = map[j]; map[j] = map[n_kmers - 1 - j]; map[n_kmers - 1 - j] = tmp; }
        for (int j = 0; j < n_kmers; ++j) { int32_t tmp = map[j].start; map[j].start = map[j].stop; map[j].stop = tmp; }
    }
    for (int j = 0; j < n_kmers; j++) {
        int32_t start_event_idx = map[j].start, end_event_idx = map[j].stop;
        if (start_event_idx == -1) {                                          /* :361-368 */
            if (end_event_idx != -1) bad = 1;
            signal_start_point = signal_end_point = -1;
            if (!ff) d++;
        } else {                                                              /* :370-402 */
            if (end_event_idx == -1) { bad = 1; break; }
            signal_start_point = (int64_t)event[start_event_idx].start;
            if (ff) { signal_start_point2 = signal_start_point; read_start = j; ci = signal_start_point; ff = 0; }
            signal_end_point2 = signal_end_point = (int64_t)event[end_event_idx].start + (int)event[end_event_idx].length;
            read_end = j;
            if (fmt) {
                if (d > 0) { sl += (size_t)sprintf(ss + sl, "%dD", (int)d); d = 0; }
                if (j == 0) ci = signal_start_point;
                ci += (mi = signal_start_point - ci);
                if (mi) { sl += (size_t)sprintf(ss + sl, "%dI", (int)mi); count_samples += mi; }
                ci += (mi = signal_end_point - signal_start_point);
                if (mi) { matches++; sl += (size_t)sprintf(ss + sl, "%d,", (int)mi); count_samples += mi; }
            }
        }
        if (fmt == 0) {                                                       /* :406-427 */
            fprintf(fp, "%s\t%d\t", read_id, rna ? n_kmers - j - 1 : j);
            if (signal_start_point < 0) fprintf(fp, ".\t"); else fprintf(fp, "%ld\t", (long)signal_start_point);
            if (signal_end_point < 0) fprintf(fp, "."); else fprintf(fp, "%ld", (long)signal_end_point);
            fprintf(fp, "\n");
            if (signal_start_point >= 0 && signal_end_point >= 0 && signal_end_point <= signal_start_point) { bad = 1; break; }
        }
    }
    if (fmt == 1 && !bad) {                                                   /* :431-447 */
        if (count_samples != (signal_end_point2 - signal_start_point2) || n_kmers <= 0 || signal_start_point2 == -1 ||
            signal_end_point2 == -1) bad = 1;
        else {
            fprintf(fp, "%s\t%ld\t%ld\t%ld\t+\t", read_id, (long)nsample, (long)signal_start_point2, (long)signal_end_point2);
            fprintf(fp, "%s\t%d\t%ld\t%ld\t", read_id, n_kmers, (long)(rna ? n_kmers - read_start : read_start),
                    (long)(rna ? n_kmers - 1 - read_end : read_end + 1));
            fprintf(fp, "%d\t%d\t%d\t", matches, n_kmers, 255);
            fprintf(fp, "sc:f:%f\t", sc_scale);
            fprintf(fp, "sh:f:%f\t", sc_shift);
            fprintf(fp, "ss:Z:%s\n", ss);
        }
    }
    fclose(fp);
    long n = bad ? -1 : (long)blen;
    if (!bad && out && cap > blen) memcpy(out, buf, blen + 1);
    free(buf); free(ss);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * Profile-HMM forward score of one event range against one (possibly methylated) sequence: row N4 of SURVEY §8f.
 * Restates profile_hmm_score -> profile_hmm_score_r9 -> profile_hmm_fill_generic_r9<ProfileHMMForwardOutputR9>
 * (src/hmm.c:314-535, 613-735) as it is compiled in the reference: ESL_LOG_SUM = 1 (f5c.h:88: table-driven float
 * log-sum, logsum.h:40-71), CACHED_LOG, HMM_REVERSE_FIX undefined, USE_EXTERNAL_PARAMS undefined.
 * PINNED: test/ecoli_2kb_region/single_read/meth_input.exp lists the arguments of read1's 90 calls and single_read/meth.exp
 * prints their results; this function reproduces all 90 (89 equal at the printed %.2f, one 0.005 from a rounding boundary):
 * tests/test_hmm_pin.py. */
#define ORC_LOGSUM_TBL 16000
static float orc_flogsum_tbl[ORC_LOGSUM_TBL];
static int orc_flogsum_ready = 0;
void orc_flogsum_init(void) {                                        /* logsum.h:33-48 */
    for (int i = 0; i < ORC_LOGSUM_TBL; i++) orc_flogsum_tbl[i] = log(1. + exp((double)-i / 1000.f));
    orc_flogsum_ready = 1;
}
const float* orc_flogsum_table(void) { if (!orc_flogsum_ready) orc_flogsum_init(); return orc_flogsum_tbl; }
static inline float orc_flogsum(float a, float b) {                  /* logsum.h:61-71 */
    const float max = (a > b) ? a : b, min = (a < b) ? a : b;
    return (min == -INFINITY || (max - min) >= 15.7f) ? max : max + orc_flogsum_tbl[(int)((max - min) * 1000.f)];
}
static inline double orc_add_logs(const double a, const double b) { return orc_flogsum(a, b); }   /* hmm.c:537-541 */

uint32_t orc_cpg_kmer_rank(const char* str, uint32_t k) {            /* hmm.c:30-61: alphabet A,C,G,M,T */
    uint32_t p = 1, r = 0;
    for (uint32_t i = 0; i < k; ++i) {
        char b = str[k - i - 1];
        uint32_t v = (b == 'A') ? 0 : (b == 'C') ? 1 : (b == 'G') ? 2 : (b == 'M') ? 3 : (b == 'T') ? 4 : 0;
        r += v * p; p *= 5;
    }
    return r;
}

float orc_profile_hmm_score(const char* m_seq, const char* m_rc_seq, const orc_event_t* event, orc_scalings_t scaling,
                            const orc_model_t* cpgmodel, uint32_t kmer_size, uint32_t event_start_idx,
                            uint32_t event_stop_idx, int8_t event_stride, uint8_t rc, double events_per_base,
                            uint32_t hmm_flags) {
    if (!orc_flogsum_ready) orc_flogsum_init();
    enum { KSKIP = 0, BAD = 1, MATCH = 2, NST = 3 };                 /* hmm.c:104-111 */
    const uint32_t k = kmer_size;
    const uint32_t n_kmers = (uint32_t)strlen(m_seq) - k + 1;        /* hmm.c:642-656 */
    const uint32_t n_cols = NST * (n_kmers + 2);
    const uint32_t e_start = event_start_idx;
    const uint32_t n_events = (event_stop_idx > e_start) ? event_stop_idx - e_start + 1 : e_start - event_stop_idx + 1;
    const uint32_t n_rows = n_events + 1;
    float* fm = (float*)malloc(sizeof(float) * (size_t)n_rows * n_cols);
    #define FM(r, c) fm[(size_t)(r) * n_cols + (c)]
    for (uint32_t r = 0; r < n_rows; r++) for (uint32_t c = 0; c < n_cols; c++) FM(r, c) = -INFINITY;   /* hmm.c:613-625; the
        reference leaves the other cells uninitialised but never reads them before writing (row 0 and block 0 are set) */
    /* transitions, identical for every k-mer (hmm.c:240-310) */
    float p_stay = 1 - (1 / events_per_base);
    float p_skip = 0.0025, p_bad = 0.001, p_bad_self = p_bad, p_skip_self = 0.3;
    float p_mk = p_skip, p_mb = p_bad, p_mm_self = p_stay, p_mm_next = 1.0f - p_mm_self - p_mk - p_mb;
    float p_bb = p_bad_self, p_bk, p_bm_next, p_bm_self;
    p_bk = p_bm_next = p_bm_self = (1.0f - p_bb) / 3;
    float p_kk = p_skip_self, p_km = 1.0f - p_kk;
    /* hmm.c is compiled as C++ (reference Makefile:6 LANGFLAG = -x c++): log() of a float argument is the float overload,
     * i.e. glibc logf, not (float)log((double)x) */
    const float lp_mk = logf(p_mk), lp_mb = logf(p_mb), lp_mm_self = logf(p_mm_self), lp_mm_next = logf(p_mm_next);
    const float lp_bb = logf(p_bb), lp_bk = logf(p_bk), lp_bm_next = logf(p_bm_next), lp_bm_self = logf(p_bm_self);
    const float lp_kk = logf(p_kk), lp_km = logf(p_km);
    /* k-mer ranks (hmm.c:383-397) */
    uint32_t* ranks = (uint32_t*)malloc(sizeof(uint32_t) * n_kmers);
    const int32_t seq_len = (int32_t)strlen(m_seq);
    for (uint32_t ki = 0; ki < n_kmers; ++ki)
        ranks[ki] = orc_cpg_kmer_rank(rc == 0 ? m_seq + ki : m_rc_seq + seq_len - ki - k, k);
    /* flanks (hmm.c:141-233) */
    float* pre = (float*)calloc(n_events + 1, sizeof(float));
    float* post = (float*)calloc(n_events, sizeof(float));
    pre[0] = log(1 - 0.5);
    pre[1] = log(0.5) + -3.0f + log(1 - 0.9);
    for (uint32_t i = 2; i < n_events + 1; ++i) pre[i] = log(0.9) + -3.0f + pre[i - 1];
    post[n_events - 1] = log(1 - 0.5);
    if (n_events > 1) {
        post[n_events - 2] = log(0.5) + -3.0f + log(1 - 0.9);
        for (int i = (int)n_events - 3; i >= 0; --i) post[i] = log(0.9) + -3.0f + post[i + 1];
    }
    const float lp_sm = 0.0f, lp_ms = 0.0f;
    float lp_end = -INFINITY;
    const uint32_t num_blocks = n_kmers + 2, last_kmer_idx = n_kmers - 1, last_row = n_rows - 1;
    for (uint32_t row = 1; row < n_rows; row++) {                    /* hmm.c:419-496 */
        for (uint32_t block = 1; block < num_blocks - 1; block++) {
            const uint32_t kmer_idx = block - 1, po = NST * (block - 1), co = NST * block;
            const uint32_t event_idx = e_start + (row - 1) * event_stride;
            const orc_model_t* m = &cpgmodel[ranks[kmer_idx]];
            /* log_probability_match_r9 (hmm.c:73-118, CACHED_LOG) */
            const float gp_mean = scaling.scale * m->level_mean + scaling.shift;
            const float gp_stdv = m->level_stdv * scaling.var;
            const float gp_log_stdv = m->level_log_stdv + scaling.log_var;
            const float a = (event[event_idx].mean - gp_mean) / gp_stdv;
            const float lp_em = -0.918938f - gp_log_stdv + (-0.5f * a * a);
            float x[6], sum;
            /* MATCH */
            x[0] = lp_mm_self + FM(row - 1, co + MATCH);
            x[1] = lp_mm_next + FM(row - 1, po + MATCH);
            x[2] = lp_bm_self + FM(row - 1, co + BAD);
            x[3] = lp_bm_next + FM(row - 1, po + BAD);
            x[4] = lp_km + FM(row - 1, po + KSKIP);
            x[5] = (kmer_idx == 0 && (event_idx == e_start || (hmm_flags & 1))) ? lp_sm + pre[row - 1] : -INFINITY;
            sum = x[0]; for (int i = 1; i < 6; ++i) sum = orc_add_logs(sum, x[i]);
            sum += lp_em; FM(row, co + MATCH) = sum;
            /* BAD_EVENT */
            x[0] = lp_mb + FM(row - 1, co + MATCH); x[1] = -INFINITY;
            x[2] = lp_bb + FM(row - 1, co + BAD); x[3] = x[4] = x[5] = -INFINITY;
            sum = x[0]; for (int i = 1; i < 6; ++i) sum = orc_add_logs(sum, x[i]);
            sum += 0.0f; FM(row, co + BAD) = sum;
            /* KMER_SKIP: same row, previous block */
            x[0] = -INFINITY; x[1] = lp_mk + FM(row, po + MATCH); x[2] = -INFINITY;
            x[3] = lp_bk + FM(row, po + BAD); x[4] = lp_kk + FM(row, po + KSKIP); x[5] = -INFINITY;
            sum = x[0]; for (int i = 1; i < 6; ++i) sum = orc_add_logs(sum, x[i]);
            sum += 0.0f; FM(row, co + KSKIP) = sum;
            if (kmer_idx == last_kmer_idx && ((hmm_flags & 2) || row == last_row)) {   /* hmm.c:478-486 */
                const float lp1 = lp_ms + FM(row, co + MATCH) + post[row - 1];
                const float lp2 = lp_ms + FM(row, co + BAD) + post[row - 1];
                const float lp3 = lp_ms + FM(row, co + KSKIP) + post[row - 1];
                lp_end = orc_add_logs(lp_end, lp1); lp_end = orc_add_logs(lp_end, lp2); lp_end = orc_add_logs(lp_end, lp3);
            }
        }
    }
    #undef FM
    free(fm); free(ranks); free(pre); free(post);
    return lp_end;
}
