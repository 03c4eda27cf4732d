/* abea_process.cpp — host-buffer entries for the callers either side of align_db (rows N2 / N3 of SURVEY §8f):
 *
 *   abea_events_batch_host   event_db  = pthread_db(event_single)   src/f5c.c:682-734
 *   abea_process_batch_host  process_db_rsq = event_db -> align_db -> scaling_db   src/resquiggle.c:283-315
 *                            (process_db runs the same three steps first, src/f5c.c:907-936)
 *   abea_rsq_format_batch    output_db_rsq's loop over the batch   src/resquiggle.c:319-449
 *
 * f5c holds a read's raw signal as FLOAT ADC counts (signal_t.rawptr, src/f5c.h:276-286: the slow5 / fast5 readers widen
 * int16) and event_single converts it to pA in place (f5c.c:693-696) before getevents().  The device detector
 * (abea_detect_events_device) takes the int16 counts and does the same float arithmetic itself, so the flatten loop here
 * narrows the floats back (2 bytes per sample over PCIe; a sample that is not an integer in int16 range is refused: it
 * cannot have come from an ADC), optionally writes the pA values into the caller's buffer as the reference does, and the
 * event tables come back as malloc()ed event_t arrays exactly where getevents() would have put them.
 *
 * The event stage works in chunks of reads whose signal + event table fit a share of the device arena; the detector's own
 * scratch comes out of the rest.  The alignment + scaling_single stage is the host pipeline of abea_host.cpp
 * (abea_align_batch_host with the fused outputs), fed with the event tables just produced.  No CPU fallback: every
 * number in the outputs was computed on the GPU.
 */
#include <atomic>
#include <cinttypes>
#include <string>
#include "abea_internal.h"

extern "C" __global__ void abea_ev_compact_kernel(int, const abea_event_t*, const int64_t*, const int64_t*, const int32_t*,
                                                  abea_event_t*);

namespace {

struct arena_view {                  /* the entries called from here take their scratch from [c->arena, +c->arena_bytes) */
    abea_ctx* c; uint8_t* a; size_t n;
    explicit arena_view(abea_ctx* ctx) : c(ctx), a(ctx->arena), n(ctx->arena_bytes) {}
    void shrink(size_t used) { c->arena = a + used; c->arena_bytes = n - used; }
    ~arena_view() { c->arena = a; c->arena_bytes = n; }
};

struct pinned_buf {
    void* p = nullptr; size_t cap = 0;
    int need(size_t n) { return ensure_pinned(&p, &cap, n); }
    ~pinned_buf() { if (p) hipHostFree(p); }
};

int check_events_batch(const abea_events_host_batch* B) {
    if (!B) return abea_fail(ABEA_EINVAL, "null argument");
    if (B->n_reads < 0) return abea_fail(ABEA_EINVAL, "n_reads < 0");
    if (B->n_reads == 0) return ABEA_OK;
    if (!B->rawptr || !B->n_samples || !B->offset || !B->range || !B->digitisation || !B->events || !B->n_events)
        return abea_fail(ABEA_EINVAL, "abea_events_batch_host: null array");
    if (B->scalings && (!B->read || !B->read_len))
        return abea_fail(ABEA_EINVAL, "abea_events_batch_host: scalings need the read sequences");
    return ABEA_OK;
}

/* event_db on one device context; the caller holds the context */
int events_locked(abea_ctx* c, const abea_events_host_batch* B, double* kernel_ms) {
    const int32_t n = B->n_reads;
    HIP_TRY(hipSetDevice(c->device));
    const bool want_sc = B->scalings != nullptr;
    std::vector<int32_t> todo;                          /* reads with a signal, in batch order (f5c.c:684) */
    for (int32_t i = 0; i < n; ++i) { B->events[i] = nullptr; B->n_events[i] = 0; }   /* before any exit: the callers free() these */
    for (int32_t i = 0; i < n; ++i) {
        if (B->n_samples[i] <= 0) continue;              /* f5c.c:727-731: et.n = 0, et.event = NULL */
        if (!B->rawptr[i]) return abea_fail(ABEA_EINVAL, "read %d: null signal", i);
        if (B->n_samples[i] > (int64_t)INT32_MAX - 64) return abea_fail(ABEA_EINVAL, "read %d: %" PRId64 " samples", i, B->n_samples[i]);
        if (want_sc && (!B->read[i] || B->read_len[i] < (int32_t)c->k))
            return abea_fail(ABEA_EINVAL, "read %d: sequence shorter than k", i);
        todo.push_back(i);
    }
    arena_view view(c);
    pinned_buf up, dn;
    /* ---- chunks: signal (2 B per sample) + event table + compacted copy (24 B per slot each) in <= 40 % of the arena, the
     *      detector's scratch (<= 24 B per sample of the waves it runs at once; it sub-batches itself) in the rest ---- */
    const size_t budget = std::min<size_t>(view.n / 5 * 2, (size_t)24 << 30);
    size_t cap_div = 4;                                /* event slots: n/4 + 16 first, the true counts when one overflowed */
    std::vector<int32_t> cap;
    size_t q0 = 0;
    while (q0 < todo.size()) {
        size_t bytes = 1 << 16, q1 = q0;
        while (q1 < todo.size() && q1 - q0 < ((size_t)1 << 20)) {
            const int32_t i = todo[q1];
            const size_t ns = (size_t)B->n_samples[i];
            const size_t need = align_up(ns * 2, 16) + (ns / cap_div + 16) * 48 + (want_sc ? (size_t)B->read_len[i] + 17 : 0) + 128;
            if (bytes + need > budget && q1 > q0) break;
            if (bytes + need > budget) return abea_fail(ABEA_ENOMEM, "read %d (%zu samples) does not fit the %zu-byte arena", i, ns, view.n);
            bytes += need; ++q1;
        }
        const int32_t m = (int32_t)(q1 - q0);
        const int32_t* rd = todo.data() + q0;            /* caller index of chunk read j */
        cap.assign((size_t)m, 0);
        for (int32_t j = 0; j < m; ++j) cap[(size_t)j] = (int32_t)std::min<size_t>((size_t)B->n_samples[rd[j]] / cap_div + 16, INT32_MAX / 2);
        bool recarve = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            /* ---- layout ---- */
            std::vector<int64_t> sig_ptr((size_t)m), ev_ptr((size_t)m), read_ptr((size_t)m), out_ptr((size_t)m);
            std::vector<int32_t> ns32((size_t)m), rl((size_t)m);
            std::vector<float> sc3((size_t)m * 3);
            size_t n_sig = 0, n_slot = 0, n_seq = 0;
            for (int32_t j = 0; j < m; ++j) {
                const int32_t i = rd[j];
                const int64_t ns = B->n_samples[i];
                sig_ptr[(size_t)j] = (int64_t)n_sig; n_sig += (size_t)((ns + 7) / 8 * 8);
                ns32[(size_t)j] = (int32_t)ns;
                ev_ptr[(size_t)j] = (int64_t)n_slot; n_slot += (size_t)cap[(size_t)j];
                sc3[(size_t)j * 3] = B->offset[i]; sc3[(size_t)j * 3 + 1] = B->range[i]; sc3[(size_t)j * 3 + 2] = B->digitisation[i];
                rl[(size_t)j] = want_sc ? B->read_len[i] : (int32_t)c->k;
                read_ptr[(size_t)j] = (int64_t)n_seq; n_seq += (size_t)rl[(size_t)j] + 1;
            }
            const size_t o_sig = 0, o_seq = align_up(n_sig * 2, 256), u_end = align_up(o_seq + n_seq, 256);
            int rc = up.need(std::max(u_end, (size_t)m * 16));
            if (rc) return rc;
            int16_t* h_sig = (int16_t*)((uint8_t*)up.p + o_sig);
            char* h_seq = (char*)up.p + o_seq;
            /* ---- flatten: float ADC counts -> int16 ---- */
            std::atomic<int32_t> bad(-1);
            abea_parallel_for(c, m, 1, [&](int64_t lo, int64_t hi) {
                for (int64_t j = lo; j < hi; ++j) {
                    const int32_t i = rd[j];
                    const int64_t ns = ns32[(size_t)j];
                    int16_t* dst = h_sig + sig_ptr[(size_t)j];
                    const float* src = B->rawptr[i];
                    bool ok = true;
                    for (int64_t t = 0; t < ns; ++t) {
                        const float v = src[t];
                        const int32_t q = (int32_t)v;
                        ok &= (v >= -32768.0f) & (v <= 32767.0f) & ((float)q == v);
                        dst[t] = (int16_t)q;
                    }
                    if (!ok) bad.store(i);
                    for (int64_t t = ns; t < (ns + 7) / 8 * 8; ++t) dst[t] = 0;
                    if (want_sc) memcpy(h_seq + read_ptr[(size_t)j], B->read[i], (size_t)rl[(size_t)j]);
                    else memset(h_seq + read_ptr[(size_t)j], 'A', (size_t)rl[(size_t)j]);
                    h_seq[read_ptr[(size_t)j] + rl[(size_t)j]] = '\0';
                }
            });
            if (bad.load() >= 0) return abea_fail(ABEA_EINVAL, "read %d: a raw sample is not an int16 ADC count (already converted to pA?)", bad.load());
            /* ---- device block at the head of the arena: [signal][sequences][event slots][compacted events][n_events][scalings] ---- */
            uint8_t* p = view.a;
            uint8_t* d_up = p;                              p += u_end;
            abea_event_t* d_ev = (abea_event_t*)p;          p += align_up(n_slot * sizeof(abea_event_t), 256);
            abea_event_t* d_evc = (abea_event_t*)p;         p += align_up(n_slot * sizeof(abea_event_t), 256);
            int32_t* d_ne = (int32_t*)p;                    p += align_up((size_t)m * 4, 256);
            abea_scalings_t* d_sc = (abea_scalings_t*)p;    p += align_up((size_t)m * sizeof(abea_scalings_t), 256);
            int64_t* d_idx = (int64_t*)p;                   p += align_up((size_t)m * 16, 256);
            const size_t used = align_up((size_t)(p - view.a), 4096);
            if (used + ((size_t)64 << 20) > view.n) return abea_fail(ABEA_ENOMEM, "event chunk of %d reads leaves no scratch in the %zu-byte arena", m, view.n);
            view.shrink(used);
            HIP_TRY(hipMemcpyAsync(d_up, up.p, u_end, hipMemcpyHostToDevice, c->stream));
            abea_signal_batch sb;
            memset(&sb, 0, sizeof sb);
            sb.n_reads = m; sb.sig_ptr = sig_ptr.data(); sb.n_samples = ns32.data(); sb.scaling = sc3.data();
            sb.event_ptr = ev_ptr.data(); sb.event_cap = cap.data();
            sb.read_ptr = read_ptr.data(); sb.read_len = rl.data();
            sb.signal = (const int16_t*)(d_up + o_sig); sb.reads = want_sc ? (const char*)(d_up + o_seq) : nullptr;
            sb.events = d_ev; sb.n_events = d_ne; sb.scalings = want_sc ? d_sc : nullptr; sb.rna = B->rna;
            rc = abea_detect_events_locked(c, &sb);
            view.shrink(0);
            if (rc) return rc;
            if (kernel_ms) *kernel_ms += c->stats.event_ms;
            /* ---- counts and scalings down; a table that overflowed its slots is redone with the true counts ---- */
            const size_t o_ne = 0, o_sc = align_up((size_t)m * 4, 256), o_ev = align_up(o_sc + (size_t)m * sizeof(abea_scalings_t), 256);
            rc = dn.need(o_ev + 256);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync((uint8_t*)dn.p + o_ne, d_ne, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
            if (want_sc) HIP_TRY(hipMemcpyAsync((uint8_t*)dn.p + o_sc, d_sc, (size_t)m * sizeof(abea_scalings_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            bool over = false;
            size_t n_ev = 0;
            {
                const int32_t* h_ne = (const int32_t*)((uint8_t*)dn.p + o_ne);
                for (int32_t j = 0; j < m; ++j) {
                    if (h_ne[j] > cap[(size_t)j]) { over = true; cap[(size_t)j] = h_ne[j]; }
                    out_ptr[(size_t)j] = (int64_t)n_ev; n_ev += (size_t)std::max(h_ne[j], 0);
                }
            }
            if (over) {
                if (attempt == 1) return abea_fail(ABEA_EHIP, "internal: event tables overflowed twice");
                size_t need = 1 << 16;
                for (int32_t j = 0; j < m; ++j) need += (size_t)ns32[(size_t)j] * 2 + (size_t)cap[(size_t)j] * 48 + (size_t)rl[(size_t)j] + 160;
                if (need > budget) { recarve = true; break; }        /* cut the chunk again with one slot per sample */
                continue;
            }
            /* ---- compact the tables on the device, one copy down, scatter into malloc()ed event_t arrays (getevents,
             *      events.c:562-582 returns a malloc()ed table; free() it like free_db_tmp does) ---- */
            const std::vector<int32_t> h_ne((const int32_t*)((uint8_t*)dn.p + o_ne), (const int32_t*)((uint8_t*)dn.p + o_ne) + m);
            const std::vector<abea_scalings_t> h_sc((const abea_scalings_t*)((uint8_t*)dn.p + o_sc), (const abea_scalings_t*)((uint8_t*)dn.p + o_sc) + m);
            rc = dn.need(o_ev + n_ev * sizeof(abea_event_t) + 256);      /* may move the block: the small arrays were copied out */
            if (rc) return rc;
            {
                int64_t* h_idx = (int64_t*)up.p;                          /* the staging block is free again */
                for (int32_t j = 0; j < m; ++j) { h_idx[j] = ev_ptr[(size_t)j]; h_idx[m + j] = out_ptr[(size_t)j]; }
                HIP_TRY(hipMemcpyAsync(d_idx, h_idx, (size_t)m * 16, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(abea_ev_compact_kernel, dim3((unsigned)m), dim3(256), 0, c->stream,
                                   (int)m, (const abea_event_t*)d_ev, (const int64_t*)d_idx, (const int64_t*)(d_idx + m),
                                   (const int32_t*)d_ne, d_evc);
                HIP_TRY(hipGetLastError());
                if (n_ev) HIP_TRY(hipMemcpyAsync((uint8_t*)dn.p + o_ev, d_evc, n_ev * sizeof(abea_event_t), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            const abea_event_t* h_ev = (const abea_event_t*)((uint8_t*)dn.p + o_ev);
            std::atomic<bool> oom(false);
            const bool to_pa = B->signal_to_pa_in_place != 0;
            abea_parallel_for(c, m, 1, [&](int64_t lo, int64_t hi) {
                for (int64_t j = lo; j < hi; ++j) {
                    const int32_t i = rd[j];
                    const size_t ne = (size_t)h_ne[(size_t)j];
                    abea_event_t* t = (abea_event_t*)malloc(std::max<size_t>(ne, 1) * sizeof(abea_event_t));
                    if (!t) { oom.store(true); continue; }
                    memcpy(t, h_ev + out_ptr[(size_t)j], ne * sizeof(abea_event_t));
                    B->events[i] = t; B->n_events[i] = ne;
                    if (want_sc) B->scalings[i] = h_sc[(size_t)j];
                    if (to_pa) {                                         /* f5c.c:693-696, the same two float operations */
                        const float raw_unit = B->range[i] / B->digitisation[i], off = B->offset[i];
                        float* src = B->rawptr[i];
                        for (int64_t t2 = 0; t2 < ns32[(size_t)j]; ++t2) src[t2] = (src[t2] + off) * raw_unit;
                    }
                }
            });
            if (oom.load()) return abea_fail(ABEA_ENOMEM, "malloc of an event table failed");
            break;
        }
        if (recarve) { cap_div = 1; continue; }          /* same q0; one slot per sample cannot overflow */
        q0 = q1;
    }
    return ABEA_OK;
}

abea_ctx* event_device(abea_ctx* c) { return c->children.empty() ? c : c->children[0]; }

}  // namespace

extern "C" int abea_events_batch_host(abea_ctx* c, const abea_events_host_batch* B) {
    if (!c) return abea_fail(ABEA_EINVAL, "null argument");
    int rc = check_events_batch(B);
    if (rc || B->n_reads == 0) return rc;
    ABEA_API_ENTER(c, "abea_events_batch_host");
    const double t0 = abea_now_ms();
    double ms = 0;
    rc = events_locked(event_device(c), B, &ms);
    if (rc) {                                            /* nothing half-built is handed back */
        for (int32_t i = 0; i < B->n_reads; ++i) { free(B->events[i]); B->events[i] = nullptr; B->n_events[i] = 0; }
        return rc;
    }
    memset(&c->stats, 0, sizeof c->stats);
    c->stats.event_ms = ms; c->stats.total_ms = abea_now_ms() - t0; c->stats.n_devices = 1;
    return ABEA_OK;
}

extern "C" int abea_process_batch_host(abea_ctx* c, const abea_process_batch* P) {
    if (!c || !P) return abea_fail(ABEA_EINVAL, "null argument");
    const int32_t n = P->n_reads;
    if (n < 0) return abea_fail(ABEA_EINVAL, "n_reads < 0");
    if (n == 0) return ABEA_OK;
    if (!P->rawptr || !P->n_samples || !P->offset || !P->range || !P->digitisation || !P->read || !P->read_len || !P->events ||
        !P->n_events || !P->scalings || !P->n_pairs || !P->base_to_event_map || !P->events_per_base || !P->read_stat_flag ||
        !P->n_event_alignment)
        return abea_fail(ABEA_EINVAL, "abea_process_batch_host: null array");
    ABEA_API_ENTER(c, "abea_process_batch_host");
    const double t0 = abea_now_ms();
    /* ---- event_db (f5c.c:682-734) ---- */
    abea_events_host_batch E;
    memset(&E, 0, sizeof E);
    E.n_reads = n; E.rawptr = P->rawptr; E.n_samples = P->n_samples; E.offset = P->offset; E.range = P->range;
    E.digitisation = P->digitisation; E.read = P->read; E.read_len = P->read_len; E.rna = P->rna;
    E.signal_to_pa_in_place = P->signal_to_pa_in_place; E.events = P->events; E.n_events = P->n_events; E.scalings = P->scalings;
    for (int32_t i = 0; i < n; ++i) {
        if (P->pairs) P->pairs[i] = nullptr;
        P->base_to_event_map[i] = nullptr; P->n_pairs[i] = 0;
        if (P->n_samples[i] <= 0) { abea_scalings_t z; memset(&z, 0, sizeof z); P->scalings[i] = z; }
    }
    double ev_ms = 0;
    auto release = [&]() {
        for (int32_t i = 0; i < n; ++i) {
            free(P->events[i]); P->events[i] = nullptr; P->n_events[i] = 0;
            if (P->pairs) { free(P->pairs[i]); P->pairs[i] = nullptr; }
            free(P->base_to_event_map[i]); P->base_to_event_map[i] = nullptr;
        }
    };
    int rc = events_locked(event_device(c), &E, &ev_ms);
    if (rc) { release(); return rc; }
    if (P->scalings_estimated) memcpy(P->scalings_estimated, P->scalings, (size_t)n * sizeof(abea_scalings_t));
    /* ---- align_db + scaling_db (f5c.c:833-845, 736-807) on the tables just made: what event_single and scaling_single
     *      malloc() per read (f5c.c:722-725, 746) is malloc()ed here ---- */
    std::vector<const abea_event_t*> ev((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        ev[(size_t)i] = P->events[i];
        const int32_t nk = P->read_len[i] - (int32_t)c->k + 1;
        if (P->n_samples[i] <= 0 || P->n_events[i] == 0 || nk <= 0) continue;
        if (P->pairs) P->pairs[i] = (abea_pair_t*)malloc(sizeof(abea_pair_t) * ((size_t)P->n_events[i] + (size_t)P->read_len[i]));
        P->base_to_event_map[i] = (abea_index_pair_t*)malloc(sizeof(abea_index_pair_t) * (size_t)nk);
        if ((P->pairs && !P->pairs[i]) || !P->base_to_event_map[i]) { release(); return abea_fail(ABEA_ENOMEM, "malloc failed for read %d", i); }
    }
    abea_host_batch H;
    memset(&H, 0, sizeof H);
    H.n_reads = n; H.read = P->read; H.read_len = P->read_len; H.events = ev.data(); H.n_events = P->n_events;
    H.scalings = P->scalings; H.n_samples = P->n_samples; H.pairs = P->pairs; H.n_pairs = P->n_pairs; H.diag = P->diag;
    H.base_to_event_map = P->base_to_event_map; H.scalings_out = P->scalings; H.events_per_base = P->events_per_base;
    H.read_stat_flag = P->read_stat_flag; H.n_event_alignment = P->n_event_alignment;
    H.min_num_events_to_rescale = P->min_num_events_to_rescale;
    rc = abea_host_batch_locked(c, &H);
    if (rc) { release(); return rc; }
    for (int32_t i = 0; i < n; ++i)                      /* scaling_single leaves the map NULL for a read that did not align */
        if (P->n_pairs[i] <= 0 && P->base_to_event_map[i]) { free(P->base_to_event_map[i]); P->base_to_event_map[i] = nullptr; }
    c->stats.event_ms = ev_ms;
    c->stats.total_ms = abea_now_ms() - t0;
    return ABEA_OK;
}

/* output_db_rsq (resquiggle.c:319-449): reads whose read_stat_flag is clear are printed in batch order; the sc:f / sh:f tags
 * of the PAF line carry the FIRST read's scalings (db->scalings->scale, resquiggle.c:443-444).  snprintf-like. */
extern "C" int64_t abea_rsq_format_batch(char* out, size_t cap, int fmt, int32_t n_reads, const char* const* read_id,
                                         const int32_t* read_len, uint32_t kmer_size, abea_index_pair_t* const* base_to_event_map,
                                         const abea_event_t* const* events, const int64_t* n_samples, const abea_scalings_t* scalings,
                                         const int32_t* read_stat_flag, int rna, int32_t* n_printed) {
    if (n_reads < 0 || (n_reads && (!read_id || !read_len || !base_to_event_map || !events || !n_samples || !scalings || !read_stat_flag)))
        return ABEA_EINVAL;
    std::string all;
    int32_t printed = 0;
    std::vector<char> buf;
    std::vector<abea_index_pair_t> rmap;
    for (int32_t i = 0; i < n_reads; ++i) {
        if (read_stat_flag[i]) continue;                 /* resquiggle.c:322; the else branch only counts the failures */
        if (!base_to_event_map[i] || !events[i] || !read_id[i]) return ABEA_EINVAL;
        /* a k-mer adds at most its id, three 20-digit numbers and
         * separators to the TSV, fewer bytes to the PAF string */
        const size_t nk = (size_t)std::max(read_len[i] - (int32_t)kmer_size + 1, 1);
        buf.resize(nk * (strlen(read_id[i]) + 72) + 2 * strlen(read_id[i]) + 512);
        abea_index_pair_t* map = base_to_event_map[i];
        if (rna) { rmap.assign(map, map + nk); map = rmap.data(); }   /* reversed in place by the per-read call: on a copy, so
                                                                        * that the size query and the real call see the same map */
        const int64_t len = abea_rsq_format(buf.data(), buf.size(), fmt, read_id[i], read_len[i], kmer_size, map,
                                            events[i], n_samples[i], scalings[0].scale, scalings[0].shift, rna);
        if (len < 0) return len;
        if ((size_t)len >= buf.size()) return ABEA_EINVAL;
        all.append(buf.data(), (size_t)len);
        ++printed;
    }
    if (n_printed) *n_printed = printed;
    if (out && cap) {
        const size_t w = std::min(all.size(), cap - 1);
        memcpy(out, all.data(), w);
        out[w] = '\0';
    }
    return (int64_t)all.size();
}
