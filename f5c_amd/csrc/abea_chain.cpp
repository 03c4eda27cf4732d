/* abea_chain.cpp — the chunk pipeline behind the raw-signal host entries of libabea_hip.so (rows N2 / N3 of SURVEY §8f):
 *
 *   abea_events_batch_host   event_db = pthread_db(event_single)                  src/f5c.c:682-734
 *   abea_process_batch_host  process_db_rsq = event_db -> align_db -> scaling_db  src/resquiggle.c:283-315
 *                            (process_db starts with the same three steps, src/f5c.c:907-936)
 *
 * Round 5.  Rounds 3-4 ran the event stage chunk by chunk with a stream synchronisation after every step (flatten, copy,
 * detector, counts, compaction, table copy, scatter — nothing overlapped), and the chain then handed the 24-byte event tables
 * it had just brought down to the alignment's host entry, whose flatten loop read them again (60 GB of host DRAM per 2.5 G
 * events) and sent the 4-byte means back up.  Now one pipeline serves both entries:
 *   - reads go longest signal first and are cut into chunks; a chunk is the unit of
 *       D  flatten (float ADC counts -> int16, optionally -> pA in place, sequences) -> ONE H2D copy -> the detector's kernels
 *          (abea_detect_events_on) -> counts + method-of-moments scalings down (copy-out kernel)
 *       A  counts known -> compaction + event tables down (copy-out kernel into pinned memory) and — process only —
 *          descriptors up, abea_pre_kernel reading the event means from the tables WHERE THE DETECTOR LEFT THEM IN HBM,
 *          abea_align_kernel with scaling_single fused, result block down
 *       R  retire: malloc()ed event tables, pair lists (optional), base_to_event_map, scalars into the caller's db
 *   - chunks rotate through N slots (own stream, pinned staging, share of the device arena); stage A of chunk c is issued
 *     after stage D of chunk c + 1, so the caller's thread never waits for a detector it has just started; kernels and both
 *     PCIe directions of neighbouring chunks overlap with the host loops;
 *   - host loops run on the context's worker pool (abea_parallel_for).
 * A table that overflows its first-guess capacity (n_samples / 4 + 16 events) is redone after the pipeline has drained from
 * the int16 samples kept in the chunk's pinned staging (the float signal may already have been rewritten to pA).
 * Every number in the outputs is computed on the GPU; there is no CPU fallback.
 */
#include <atomic>
#include <cinttypes>
#include <numeric>
#include <string>
#include <thread>
#include <immintrin.h>
#include "abea_internal.h"

extern "C" __global__ void abea_ev_compact_kernel(int, const abea_event_t*, const int64_t*, const int64_t*, const int32_t*,
                                                  abea_event_t*);
extern "C" __global__ void abea_ev_pack_kernel(int, const abea_event_t*, const int64_t*, const int64_t*, const int32_t*, uint4*);

namespace {

struct pinned_buf {
    void* p = nullptr; size_t cap = 0;
    int need(size_t n) { return ensure_pinned(&p, &cap, n); }
    uint8_t* u8() const { return (uint8_t*)p; }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

/* one chunk in flight */
struct abea_chain_slot {
    hipStream_t stream = nullptr, dstream = nullptr;  /* dstream: the tables' own device-to-host stream (copy-engine mode) */
    hipEvent_t e_cnt = nullptr, e_done = nullptr, t0 = nullptr, t1 = nullptr, t2 = nullptr, t3 = nullptr, t4 = nullptr;
    hipEvent_t e_cmp = nullptr, e_tab = nullptr;     /* compaction done (kernel stream) / tables down (dstream) */
    bool tab_on_dstream = false, packed = false;
    pinned_buf up, cnt, desc, dn, tab;
    void* idx_p = nullptr; size_t idx_cap = 0;       /* grown by abea_detect_events_on */
    bool busy = false, staged = false;
    /* ---- the chunk ---- */
    int32_t m = 0, chunk_no = 0;
    std::vector<int32_t> rd, ns32, cap, rl, nk, ne;  /* caller index, samples, event slots, read length, k-mers, events (0 when redone) */
    std::vector<int64_t> sig_ptr, ev_ptr, read_ptr, out_ptr;
    std::vector<float> sc3;
    std::vector<uint8_t> run;                        /* aligned on the GPU (process mode) */
    std::vector<abea_scalings_t> est;                /* method-of-moments scalings */
    size_t n_sig = 0, n_slot = 0, n_seq = 0, n_ev = 0, n_rec = 0, u_end = 0, o_seq = 0;   /* n_rec: table records incl. per-read padding */
    uint8_t* arena = nullptr; size_t arena_bytes = 0;
    uint8_t* d_up = nullptr; abea_event_t* d_ev = nullptr; abea_event_t* d_evc = nullptr; uint8_t* d_cnt = nullptr;
    int64_t* d_idx = nullptr; uint8_t* d_rest = nullptr; size_t rest_bytes = 0;
    size_t cnt_bytes = 0, o_sc_cnt = 0;
    /* result block offsets (process mode), as abea_host.cpp lays them out */
    size_t o_np = 0, o_diag = 0, o_codes = 0, o_sc = 0, o_epb = 0, o_flag = 0, o_nal = 0, o_var = 0, o_kcnt = 0, dn_copy = 0;
    std::vector<abea_read_desc> descs;               /* host copy: code_off / kmer_off / n_kmers at retire */
};

static void abea_chain_slot_destroy(abea_chain_slot* s);

void abea_chain_release(abea_ctx* c) {
    for (abea_chain_slot* s : c->chain_slots) {
        if (!s) continue;
        abea_chain_slot_destroy(s);
    }
    c->chain_slots.clear();
}

static void abea_chain_slot_destroy(abea_chain_slot* s) {
    if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    if (s->dstream) { hipStreamSynchronize(s->dstream); hipStreamDestroy(s->dstream); }
    for (hipEvent_t e : {s->e_cnt, s->e_done, s->t0, s->t1, s->t2, s->t3, s->t4, s->e_cmp, s->e_tab}) if (e) hipEventDestroy(e);
    s->up.release(); s->cnt.release(); s->desc.release(); s->dn.release(); s->tab.release();
    if (s->idx_p) hipHostFree(s->idx_p);
    delete s;
}

namespace {

/* packed: the tables cross PCIe as 12-byte {start, mean, stdv} records and the retire loop rebuilds event_t from them (half the bytes
 * of the call's dominant transfer); copy_engine: they come down by hipMemcpyAsync on the slot's own device-to-host stream instead of
 * abea_copy_out_kernel.  Defaults = what measured fastest on the MI355X box (profiles/r06/chain_*); ABEA_CHAIN_TABLE_FORMAT=full|packed
 * and ABEA_CHAIN_TABLE_COPY=kernel|engine override (read per call: A/B runs in one process). */
struct chain_opts { size_t chunk_samples; int32_t reads_min, reads_max; int n_slots; size_t cap_div; bool packed, copy_engine, up_kernel; int depth, copy_blocks; };

chain_opts read_chain_opts() {
    chain_opts o;
    o.chunk_samples = (size_t)192 << 20; o.reads_min = 256; o.reads_max = 16384; o.n_slots = 6; o.cap_div = 4;
    if (const char* e = getenv("ABEA_CHAIN_CHUNK_SAMPLES")) o.chunk_samples = std::max<size_t>(1, strtoull(e, nullptr, 10));
    if (const char* e = getenv("ABEA_CHAIN_CHUNK_READS")) o.reads_min = std::max(1, atoi(e));
    if (const char* e = getenv("ABEA_CHAIN_CHUNK_READS_MAX")) o.reads_max = std::max(1, atoi(e));
    if (const char* e = getenv("ABEA_CHAIN_SLOTS")) o.n_slots = std::min(ABEA_MAX_SLOTS, std::max(1, atoi(e)));
    if (const char* e = getenv("ABEA_CHAIN_CAP_DIV")) o.cap_div = (size_t)std::max(1, atoi(e));
    o.packed = true; o.copy_engine = true;
    o.depth = ABEA_MAX_SLOTS;                            /* chunks in flight: as many as there are slots; ABEA_CHAIN_DEPTH=2 is round 5's pairing */
    if (const char* e = getenv("ABEA_CHAIN_DEPTH")) o.depth = std::max(1, atoi(e));
    o.up_kernel = false;                                 /* ABEA_CHAIN_UP=kernel: the signal comes up by the copy kernel (loads from pinned memory) */
    if (const char* e = getenv("ABEA_CHAIN_UP")) o.up_kernel = strcmp(e, "kernel") == 0;
    o.copy_blocks = abea_copy_kernel_blocks();
    if (const char* e = getenv("ABEA_CHAIN_TABLE_FORMAT")) o.packed = strcmp(e, "full") != 0;
    if (const char* e = getenv("ABEA_CHAIN_TABLE_COPY")) o.copy_engine = strcmp(e, "engine") == 0;
    o.reads_max = std::max(o.reads_max, o.reads_min);
    return o;
}

int slot_make_parts(abea_chain_slot* s) {
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s->dstream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->e_cnt, hipEventDisableTiming | hipEventBlockingSync));
    HIP_TRY(hipEventCreateWithFlags(&s->e_done, hipEventDisableTiming | hipEventBlockingSync));
    HIP_TRY(hipEventCreateWithFlags(&s->e_tab, hipEventDisableTiming | hipEventBlockingSync));
    HIP_TRY(hipEventCreateWithFlags(&s->e_cmp, hipEventDisableTiming));
    for (hipEvent_t* e : {&s->t0, &s->t1, &s->t2, &s->t3, &s->t4}) HIP_TRY(hipEventCreate(e));
    return ABEA_OK;
}

/* a slot that could not be built completely is destroyed, never cached half-made (round-5 advisor finding) */
int slot_make(abea_chain_slot** out) {
    abea_chain_slot* s = new abea_chain_slot();
    const int rc = slot_make_parts(s);
    if (rc) { abea_chain_slot_destroy(s); *out = nullptr; return rc; }
    *out = s;
    return ABEA_OK;
}

/* one read the pipeline could not finish: its table overflowed the first-guess capacity */
struct redo_read { int32_t idx; int32_t n_events; std::vector<int16_t> samples; };

struct chain_state {
    abea_ctx* c;
    const abea_chain_job* J;
    chain_opts opt;
    abea_stats st;
    std::vector<redo_read> redo;
    hipEvent_t origin = nullptr;
    std::vector<std::pair<float, float>> spans;        /* kernel span of every retired chunk on the GPU clock */
    bool trace = false; double t_origin = 0;
    std::atomic<bool> oom{false};
    void log(const char* what, int chunk, int m = 0, size_t n = 0) const {
        if (trace) fprintf(stderr, "[abea chain dev %d] %8.2f ms  chunk %2d  %-10s reads %d samples/events %zu\n", c->device, abea_now_ms() - t_origin, chunk, what, m, n);
    }
};

/* What a chunk takes of its slot's share of the arena, accumulated read by read in the order the chunk is carved (descending sample
 * count).  Round 5 charged every read ONE lane of detector scratch; the detector (abea_detect_events_on) interleaves its arrays over
 * whole 64-lane waves, each as long as the wave's longest read, so a chunk that the arena limit closed at a read count that is not
 * a multiple of 64 — or a single very long read — passed the carving and then failed the whole batch in stage D with "internal: the
 * detector needs ..." (round-5 advisor finding).  Now the carving runs the detector's own arithmetic (abea_detect_scratch_bytes:
 * same constants, same wave rule) and the layout of stage_detect / stage_align:
 *   [signal | sequences][event slots][compacted tables][counts | scalings][index records][max(detector scratch, alignment scratch)] */
struct chunk_charge {
    bool align = false;
    size_t n = 0, up = 0, seq = 0, slots = 0, det_waves = 0, aln = 0;
    int32_t wave_cap = 0, wave_k = 0;                 /* maxima of the wave being filled */
    size_t wave_len = 0;
    static size_t wave_bytes(size_t len, int32_t cap, int32_t wk) {
        const size_t EV_SEG = 512, EV_FIXCAP = 48, SEG_BYTES = 64 * (EV_SEG * 2 + EV_FIXCAP * 4 + 12 * 4 + 2 * 8 + 4 * 4);   /* abea_capi.cpp */
        const size_t nseg = std::max<size_t>(1, (len - 1 + EV_SEG - 1) / EV_SEG);
        return len * 64 * 24 + (size_t)cap * 64 * 8 + (size_t)wk * 64 * 4 + nseg * SEG_BYTES;
    }
    /* total with the read added; commit = keep it */
    size_t with(int64_t ns, int32_t cap, int32_t L, int32_t K, bool commit) {
        chunk_charge t = *this;
        t.up += (size_t)((ns + 7) / 8 * 8) * 2;
        t.seq += align_up((size_t)L + 1, 16);
        t.slots += (size_t)cap;
        const int32_t wk = std::max(K, 1);
        if (t.n % 64 == 0) {                          /* opens a wave: as long as this read (the longest of it), +1 for S[n] */
            t.wave_len = (size_t)ns + 1; t.wave_cap = std::max(cap, 1); t.wave_k = wk;
            t.det_waves += wave_bytes(t.wave_len, t.wave_cap, t.wave_k);
        } else if (cap > t.wave_cap || wk > t.wave_k) {
            t.det_waves -= wave_bytes(t.wave_len, t.wave_cap, t.wave_k);
            t.wave_cap = std::max(t.wave_cap, cap); t.wave_k = std::max(t.wave_k, wk);
            t.det_waves += wave_bytes(t.wave_len, t.wave_cap, t.wave_k);
        }
        if (align && K >= 1) {
            plan_read r = make_plan(0, L, cap, 1);
            r.K = K; r.n_bands = (int64_t)cap + K + 2;
            t.aln += scratch_bytes(r) + 768 + (size_t)K * 9 + 80 + 512;      /* + per-array alignment, map, count bytes, result scalars */
        }
        t.n += 1;
        const size_t N = t.n, n_waves = (N + 63) / 64;
        const size_t det = N * (4 + 8 + 4 + 12 + 8 + 4 + 8 + 4 + 4 + 4) + n_waves * 56 + 8192 + 4096 + 8192 + t.det_waves + ((size_t)1 << 20);
        const size_t head = align_up(align_up(t.up, 256) + t.seq, 256) + 2 * align_up(t.slots * sizeof(abea_event_t), 256) +
                            align_up(align_up(N * 4, 256) + N * sizeof(abea_scalings_t), 256) + align_up(N * 24, 256);
        const size_t aln_fixed = t.align ? align_up(N * sizeof(abea_read_desc), 256) + 16 * 256 + N * 80 + 4096 : 0;
        if (commit) *this = t;
        return head + std::max(det, t.aln + aln_fixed) + ((size_t)2 << 20);
    }
};

/* ------------------------------------------------------------------ stage D: signal up, detector, counts down */
int stage_detect(chain_state& S, abea_chain_slot& sl, const int32_t* ids, int32_t m, int chunk_no, uint8_t* arena, size_t arena_bytes) {
    abea_ctx* c = S.c;
    const abea_chain_job* J = S.J;
    const bool want_sc = J->read != nullptr;
    double t0 = abea_now_ms();
    S.log("flatten", chunk_no, m);
    sl.chunk_no = chunk_no; sl.m = m; sl.staged = false; sl.arena = arena; sl.arena_bytes = arena_bytes;
    sl.rd.assign(ids, ids + m);
    sl.ns32.resize((size_t)m); sl.cap.resize((size_t)m); sl.rl.resize((size_t)m); sl.nk.resize((size_t)m);
    sl.sig_ptr.resize((size_t)m); sl.ev_ptr.resize((size_t)m); sl.read_ptr.resize((size_t)m); sl.out_ptr.resize((size_t)m);
    sl.sc3.resize((size_t)m * 3); sl.ne.assign((size_t)m, 0); sl.run.assign((size_t)m, 0); sl.est.resize((size_t)m);
    size_t n_sig = 0, n_slot = 0, n_seq = 0;
    for (int32_t j = 0; j < m; ++j) {
        const int32_t i = ids[j];
        const int64_t ns = J->n_samples[i];
        sl.ns32[(size_t)j] = (int32_t)ns;
        sl.cap[(size_t)j] = (int32_t)std::min<size_t>((size_t)ns / S.opt.cap_div + 16, INT32_MAX / 2);
        sl.sig_ptr[(size_t)j] = (int64_t)n_sig; n_sig += (size_t)((ns + 7) / 8 * 8);
        sl.ev_ptr[(size_t)j] = (int64_t)n_slot; n_slot += (size_t)sl.cap[(size_t)j];
        sl.sc3[(size_t)j * 3] = J->offset[i]; sl.sc3[(size_t)j * 3 + 1] = J->range[i]; sl.sc3[(size_t)j * 3 + 2] = J->digitisation[i];
        sl.rl[(size_t)j] = want_sc ? J->read_len[i] : (int32_t)c->k;
        sl.nk[(size_t)j] = sl.rl[(size_t)j] - (int32_t)c->k + 1;
        sl.read_ptr[(size_t)j] = (int64_t)n_seq; n_seq += align_up((size_t)sl.rl[(size_t)j] + 1, 16);
    }
    sl.n_sig = n_sig; sl.n_slot = n_slot; sl.n_seq = n_seq;
    sl.o_seq = align_up(n_sig * 2, 256); sl.u_end = align_up(sl.o_seq + n_seq, 256);
    int rc = sl.up.need(sl.u_end);
    if (rc) return rc;
    int16_t* h_sig = (int16_t*)sl.up.u8();
    char* h_seq = (char*)sl.up.u8() + sl.o_seq;
    /* ---- flatten: float ADC counts -> int16 (2 bytes per sample over PCIe; f5c.h:276-286 keeps them widened), and in the
     *      same pass over the signal the pA conversion event_single leaves behind (f5c.c:693-696, the same two float
     *      operations): one read of the caller's 4 bytes per sample instead of two ---- */
    std::atomic<int32_t> bad(-1);
    const bool to_pa = J->signal_to_pa_in_place != 0;
    abea_parallel_for(c, m, 1, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; ++j) {
            const int32_t i = sl.rd[(size_t)j];
            const int64_t ns = sl.ns32[(size_t)j];
            int16_t* dst = h_sig + sl.sig_ptr[(size_t)j];
            float* src = J->rawptr[i];
            const float raw_unit = J->range[i] / J->digitisation[i], off = J->offset[i];
            bool ok = true;
            int64_t t = 0;
            for (; t + 8 <= ns; t += 8) {                      /* 8 samples: one 16-byte non-temporal store into the staging block */
                const __m128 a = _mm_loadu_ps(src + t), b = _mm_loadu_ps(src + t + 4);
                const __m128i qa = _mm_cvttps_epi32(a), qb = _mm_cvttps_epi32(b);
                const __m128 back_a = _mm_cvtepi32_ps(qa), back_b = _mm_cvtepi32_ps(qb);
                const __m128i pk = _mm_packs_epi32(qa, qb);   /* saturating; an out-of-range or fractional count fails the test below */
                const __m128 lo4 = _mm_set1_ps(-32768.0f), hi4 = _mm_set1_ps(32767.0f);
                const int same = _mm_movemask_ps(_mm_and_ps(_mm_and_ps(_mm_cmpeq_ps(back_a, a), _mm_cmpge_ps(a, lo4)), _mm_cmple_ps(a, hi4))) &
                                 _mm_movemask_ps(_mm_and_ps(_mm_and_ps(_mm_cmpeq_ps(back_b, b), _mm_cmpge_ps(b, lo4)), _mm_cmple_ps(b, hi4)));
                ok &= same == 0xF;
                _mm_stream_si128(reinterpret_cast<__m128i*>(dst + t), pk);
                if (to_pa) {
                    const __m128 o4 = _mm_set1_ps(off), u4 = _mm_set1_ps(raw_unit);
                    _mm_storeu_ps(src + t, _mm_mul_ps(_mm_add_ps(a, o4), u4));
                    _mm_storeu_ps(src + t + 4, _mm_mul_ps(_mm_add_ps(b, o4), u4));
                }
            }
            for (; t < ns; ++t) {
                const float v = src[t];
                const int32_t q = (int32_t)v;
                ok &= (v >= -32768.0f) & (v <= 32767.0f) & ((float)q == v);
                dst[t] = (int16_t)q;
                if (to_pa) src[t] = (v + off) * raw_unit;
            }
            if (!ok) bad.store(i);
            for (t = ns; t < (ns + 7) / 8 * 8; ++t) dst[t] = 0;
            char* sq = h_seq + sl.read_ptr[(size_t)j];
            if (want_sc) memcpy(sq, J->read[i], (size_t)sl.rl[(size_t)j]);
            else memset(sq, 'A', (size_t)sl.rl[(size_t)j]);
            sq[sl.rl[(size_t)j]] = '\0';
        }
        _mm_sfence();
    });
    if (bad.load() >= 0)
        return abea_fail(ABEA_EINVAL, "read %d: a raw sample is not an int16 ADC count (already converted to pA?)", bad.load());
    S.st.flatten_ms += abea_now_ms() - t0;
    /* ---- the chunk's share of the arena: [signal | sequences][event slots][compacted events][counts | scalings][idx][rest] ---- */
    uint8_t* p = arena;
    sl.d_up = p;                                     p += sl.u_end;
    sl.d_ev = (abea_event_t*)p;                      p += align_up(n_slot * sizeof(abea_event_t), 256);
    sl.d_evc = (abea_event_t*)p;                     p += align_up(n_slot * sizeof(abea_event_t), 256);
    sl.d_cnt = p;
    sl.o_sc_cnt = align_up((size_t)m * 4, 256);
    sl.cnt_bytes = align_up(sl.o_sc_cnt + (size_t)m * sizeof(abea_scalings_t), 256);
    p += sl.cnt_bytes;
    sl.d_idx = (int64_t*)p;                          p += align_up((size_t)m * 24, 256);
    sl.d_rest = p;
    if ((size_t)(p - arena) + ((size_t)1 << 20) > arena_bytes)
        return abea_fail(ABEA_ENOMEM, "internal: event chunk of %d reads leaves no scratch in its %zu-byte arena share", m, arena_bytes);
    sl.rest_bytes = arena_bytes - (size_t)(p - arena);
    rc = sl.cnt.need(sl.cnt_bytes);
    if (rc) return rc;
    S.log("enqueue D", chunk_no, m, n_sig);
    sl.busy = true;                                  /* from here on the slot's streams hold work of this call: every exit drains them */
    if (S.opt.up_kernel) {
        hipLaunchKernelGGL(abea_copy_out_kernel, dim3((unsigned)std::min<size_t>((size_t)S.opt.copy_blocks, (sl.u_end / 16 + 255) / 256)), dim3(256), 0, sl.stream,
                           (const uint4*)sl.up.p, (uint4*)sl.d_up, sl.u_end / 16);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(sl.d_up, sl.up.p, sl.u_end, hipMemcpyHostToDevice, sl.stream));
    }
    S.st.h2d_bytes += sl.u_end;
    abea_signal_batch sb;
    memset(&sb, 0, sizeof sb);
    sb.n_reads = m; sb.sig_ptr = sl.sig_ptr.data(); sb.n_samples = sl.ns32.data(); sb.scaling = sl.sc3.data();
    sb.event_ptr = sl.ev_ptr.data(); sb.event_cap = sl.cap.data(); sb.read_ptr = sl.read_ptr.data(); sb.read_len = sl.rl.data();
    sb.signal = (const int16_t*)sl.d_up; sb.reads = want_sc ? (const char*)(sl.d_up + sl.o_seq) : nullptr;
    sb.events = sl.d_ev; sb.n_events = (int32_t*)sl.d_cnt;
    sb.scalings = want_sc ? (abea_scalings_t*)(sl.d_cnt + sl.o_sc_cnt) : nullptr; sb.rna = J->rna;
    abea_ev_exec X;
    const size_t det_bytes = abea_detect_scratch_bytes(sl.ns32.data(), sl.cap.data(), want_sc ? sl.nk.data() : nullptr, m);
    if (det_bytes > sl.rest_bytes)
        return abea_fail(ABEA_ENOMEM, "internal: the detector needs %zu bytes for a chunk of %d reads, its arena share has %zu left", det_bytes, m, sl.rest_bytes);
    X.stream = sl.stream; X.scratch = sl.d_rest; X.scratch_bytes = det_bytes;
    X.h_pinned = &sl.idx_p; X.h_cap = &sl.idx_cap; X.async = true; X.e0 = sl.t0; X.e1 = sl.t1;
    rc = abea_detect_events_on(c, &sb, X);
    if (rc) return rc;
    /* the alignment's scratch (stage A, same stream, behind the detector) re-uses the detector's: d_rest stays where it is */
    /* counts (+ scalings) down by a kernel, not by an SDMA copy queued behind the detector (abea_copy_out_kernel) */
    hipLaunchKernelGGL(abea_copy_out_kernel, dim3((unsigned)std::min<size_t>(64, (sl.cnt_bytes / 16 + 255) / 256)), dim3(256), 0, sl.stream,
                       (const uint4*)sl.d_cnt, (uint4*)sl.cnt.p, sl.cnt_bytes / 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.e_cnt, sl.stream));
    S.st.d2h_bytes += sl.cnt_bytes;
    return ABEA_OK;
}

/* ------------------------------------------------------------------ stage A: tables down; descriptors up, alignment, results down */
int stage_align(chain_state& S, abea_chain_slot& sl) {
    if (!sl.busy || sl.staged) return ABEA_OK;
    abea_ctx* c = S.c;
    const abea_chain_job* J = S.J;
    const int32_t m = sl.m;
    const bool want_sc = J->read != nullptr;
    double t0 = abea_now_ms();
    HIP_TRY(hipEventSynchronize(sl.e_cnt));
    S.st.wait_ms += abea_now_ms() - t0;
    t0 = abea_now_ms();
    S.log("counts", sl.chunk_no, m);
    const int32_t* h_ne = (const int32_t*)sl.cnt.u8();
    const abea_scalings_t* h_sc = (const abea_scalings_t*)(sl.cnt.u8() + sl.o_sc_cnt);
    size_t n_ev = 0, n_rec = 0;
    sl.packed = S.opt.packed; sl.tab_on_dstream = S.opt.copy_engine;
    for (int32_t j = 0; j < m; ++j) {
        int32_t ne = std::max(h_ne[j], 0);
        if (ne > sl.cap[(size_t)j]) {                        /* overflowed its slots: redone after the pipeline, from the int16 staging */
            redo_read r;
            r.idx = sl.rd[(size_t)j]; r.n_events = ne;
            const int16_t* s16 = (const int16_t*)sl.up.u8() + sl.sig_ptr[(size_t)j];
            r.samples.assign(s16, s16 + sl.ns32[(size_t)j]);
            S.redo.push_back(std::move(r));
            ne = 0;
        }
        sl.ne[(size_t)j] = ne;
        /* a packed table starts on a multiple of 4 records (the pack kernel stores 4 records = 48 bytes per thread) */
        sl.out_ptr[(size_t)j] = (int64_t)n_rec; n_ev += (size_t)ne; n_rec += sl.packed ? ((size_t)ne + 3) / 4 * 4 : (size_t)ne;
        if (want_sc) sl.est[(size_t)j] = h_sc[j];
    }
    sl.n_ev = n_ev; sl.n_rec = n_rec;
    const size_t rec_bytes = sl.packed ? 12 : sizeof(abea_event_t);
    /* ---- compaction and the tables down: [src_ptr][dst_ptr][count] up, one kernel, one copy-out into pinned memory ---- */
    int rc = sl.tab.need(align_up(n_rec * rec_bytes, 16) + 256);
    if (rc) return rc;
    size_t need_desc = (size_t)m * 24 + 256;
    if (J->align) need_desc += align_up((size_t)m * sizeof(abea_read_desc), 256) + align_up((size_t)m * sizeof(abea_scalings_t), 256) + align_up((size_t)m * 4, 256);
    rc = sl.desc.need(need_desc);
    if (rc) return rc;
    {
        int64_t* h_idx = (int64_t*)sl.desc.u8();
        int32_t* h_cnt = (int32_t*)(h_idx + 2 * (size_t)m);
        for (int32_t j = 0; j < m; ++j) { h_idx[j] = sl.ev_ptr[(size_t)j]; h_idx[m + j] = sl.out_ptr[(size_t)j]; h_cnt[j] = sl.ne[(size_t)j]; }
        HIP_TRY(hipMemcpyAsync(sl.d_idx, h_idx, (size_t)m * 20, hipMemcpyHostToDevice, sl.stream));
        if (sl.packed)
            hipLaunchKernelGGL(abea_ev_pack_kernel, dim3((unsigned)m), dim3(256), 0, sl.stream, (int)m, (const abea_event_t*)sl.d_ev,
                               (const int64_t*)sl.d_idx, (const int64_t*)(sl.d_idx + m), (const int32_t*)(sl.d_idx + 2 * (size_t)m), (uint4*)sl.d_evc);
        else
            hipLaunchKernelGGL(abea_ev_compact_kernel, dim3((unsigned)m), dim3(256), 0, sl.stream, (int)m, (const abea_event_t*)sl.d_ev,
                               (const int64_t*)sl.d_idx, (const int64_t*)(sl.d_idx + m), (const int32_t*)(sl.d_idx + 2 * (size_t)m), sl.d_evc);
        const size_t n16 = (n_rec * rec_bytes + 15) / 16;
        if (n16 && sl.tab_on_dstream) {
            /* the copy engine on the slot's own device-to-host stream, behind the compaction only: the chunk's alignment kernels
             * (process mode) run on sl.stream meanwhile and no CU spends its time storing across PCIe */
            HIP_TRY(hipEventRecord(sl.e_cmp, sl.stream));
            HIP_TRY(hipStreamWaitEvent(sl.dstream, sl.e_cmp, 0));
            HIP_TRY(hipMemcpyAsync(sl.tab.p, sl.d_evc, n16 * 16, hipMemcpyDeviceToHost, sl.dstream));
        } else if (n16) {
            hipLaunchKernelGGL(abea_copy_out_kernel, dim3((unsigned)std::min<size_t>((size_t)S.opt.copy_blocks, (n16 + 255) / 256)), dim3(256), 0, sl.stream,
                               (const uint4*)sl.d_evc, (uint4*)sl.tab.p, n16);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sl.e_tab, sl.tab_on_dstream ? sl.dstream : sl.stream));
        S.st.d2h_bytes += n16 * 16; S.st.h2d_bytes += (size_t)m * 20;
    }
    HIP_TRY(hipEventRecord(sl.t2, sl.stream));
    if (!J->align) {
        HIP_TRY(hipEventRecord(sl.e_done, sl.stream));
        sl.staged = true;
        S.st.plan_ms += abea_now_ms() - t0;
        return ABEA_OK;
    }
    /* ---- alignment + scaling_single on the tables in HBM (align_db + scaling_db, f5c.c:833-845, 736-807) ---- */
    sl.descs.resize((size_t)m);
    sub_layout lay;
    size_t n_kmer = 0;
    for (int32_t j = 0; j < m; ++j) {
        plan_read r = make_plan(j, sl.rl[(size_t)j], sl.ne[(size_t)j], c->k);
        if (r.run && r.n_bands > ABEA_MAX_BANDS)
            return abea_fail(ABEA_EINVAL, "read %d has %lld bands; the limit is %lld", sl.rd[(size_t)j], (long long)r.n_bands, (long long)ABEA_MAX_BANDS);
        sl.run[(size_t)j] = r.run ? 1 : 0;
        /* a read whose table overflowed comes through here with E = 0 and is aligned later by redo_overflowed, which counts it */
        const bool redone = sl.ne[(size_t)j] == 0 && h_ne[j] > sl.cap[(size_t)j];
        if (r.run) ++S.st.n_reads_gpu; else if (!redone) ++S.st.n_reads_skipped;
        abea_read_desc& d = sl.descs[(size_t)j];
        plan_desc_layout(d, r, sl.est[(size_t)j], lay, S.st);
        d.read_off = sl.read_ptr[(size_t)j];
        d.event_off = sl.ev_ptr[(size_t)j];
        d.pair_off = 0;
        d.kmer_off = (int64_t)n_kmer;
        n_kmer += (size_t)std::max(r.K, 0);
    }
    abea_parallel_for(c, m, 64, [&](int64_t lo, int64_t hi) { for (int64_t j = lo; j < hi; ++j) plan_desc_consts(sl.descs[(size_t)j]); });
    /* result block, as abea_host.cpp lays it out: [n_pairs][diag][walk codes][scalings][events_per_base][flags][n_alignment][var][counts per k-mer] */
    size_t o = 0;
    sl.o_np = o;    o = align_up(o + (size_t)m * 4, 256);
    sl.o_diag = o;  o = align_up(o + (size_t)m * sizeof(abea_read_diag), 256);
    sl.o_codes = o; o = align_up(o + lay.n_code * 4, 256);
    sl.o_sc = o;    o = align_up(o + (size_t)m * sizeof(abea_scalings_t), 256);
    sl.o_epb = o;   o = align_up(o + (size_t)m * 8, 256);
    sl.o_flag = o;  o = align_up(o + (size_t)m * 4, 256);
    sl.o_nal = o;   o = align_up(o + (size_t)m * 4, 256);
    sl.o_var = o;   o = align_up(o + (size_t)m * 8, 256);
    sl.o_kcnt = o;  o = align_up(o + n_kmer, 256);
    sl.dn_copy = o;
    rc = sl.dn.need(o);
    if (rc) return rc;
    uint8_t* p = sl.d_rest;
    abea_read_desc* d_desc = (abea_read_desc*)p;       p += align_up((size_t)m * sizeof(abea_read_desc), 256);
    abea_kpar_t* d_kpar = (abea_kpar_t*)p;             p += align_up(lay.n_kpar * sizeof(abea_kpar_t), 256);
    uint32_t* d_krank = (uint32_t*)p;                  p += align_up(lay.n_kpar * 4, 256);
    float* d_evm = (float*)p;                          p += align_up(lay.n_evm * 4 + 512, 256);
    uint4* d_trace = (uint4*)p;                        p += align_up(lay.n_trace * sizeof(uint4), 256);
    uint8_t* d_dn = p;                                 p += sl.dn_copy;
    abea_index_pair_t* d_b2e = (abea_index_pair_t*)p;  p += align_up(n_kmer * sizeof(abea_index_pair_t), 256);
    if ((size_t)(p - sl.d_rest) > sl.rest_bytes)
        return abea_fail(ABEA_ENOMEM, "internal: the alignment of chunk %d needs %zu bytes, its arena share has %zu left", sl.chunk_no,
                         (size_t)(p - sl.d_rest), sl.rest_bytes);
    uint8_t* h = sl.desc.u8() + align_up((size_t)m * 24, 256);
    abea_read_desc* h_desc = (abea_read_desc*)h;          h += align_up((size_t)m * sizeof(abea_read_desc), 256);
    abea_scalings_t* h_sc_in = (abea_scalings_t*)h;       h += align_up((size_t)m * sizeof(abea_scalings_t), 256);
    int32_t* h_flag = (int32_t*)h;
    for (int32_t j = 0; j < m; ++j) {
        h_desc[j] = sl.descs[(size_t)j];
        h_sc_in[j] = sl.est[(size_t)j];
        h_flag[j] = J->read_stat_flag ? J->read_stat_flag[sl.rd[(size_t)j]] : 0;
    }
    S.st.plan_ms += abea_now_ms() - t0;
    S.log("enqueue A", sl.chunk_no, m, n_ev);
    HIP_TRY(hipMemcpyAsync(d_desc, h_desc, (size_t)m * sizeof(abea_read_desc), hipMemcpyHostToDevice, sl.stream));
    HIP_TRY(hipMemcpyAsync(d_dn + sl.o_sc, h_sc_in, (size_t)m * sizeof(abea_scalings_t), hipMemcpyHostToDevice, sl.stream));
    HIP_TRY(hipMemcpyAsync(d_dn + sl.o_flag, h_flag, (size_t)m * 4, hipMemcpyHostToDevice, sl.stream));
    S.st.h2d_bytes += (size_t)m * (sizeof(abea_read_desc) + sizeof(abea_scalings_t) + 4);
    const char* d_reads = (const char*)(sl.d_up + sl.o_seq);
    hipLaunchKernelGGL(abea_pre_kernel, dim3((unsigned)m), dim3(256), 0, sl.stream,
                       (const abea_read_desc*)d_desc, d_reads, (const abea_event_t*)sl.d_ev, c->d_model, (int)c->k, d_kpar, d_evm, d_krank);
    HIP_TRY(hipEventRecord(sl.t3, sl.stream));
    abea_fused_scaling fs;
    memset(&fs, 0, sizeof fs);
    fs.reads = d_reads; fs.model = c->d_model; fs.b2e = d_b2e;
    fs.sc_io = (abea_scalings_t*)(d_dn + sl.o_sc); fs.epb = (double*)(d_dn + sl.o_epb);
    fs.flag_io = (int32_t*)(d_dn + sl.o_flag); fs.nalign = (int32_t*)(d_dn + sl.o_nal);
    fs.kcnt = (uint8_t*)(d_dn + sl.o_kcnt); fs.var_f64 = (double*)(d_dn + sl.o_var);
    fs.kmer_size = (int32_t)c->k; fs.min_rescale = J->min_num_events_to_rescale > 0 ? J->min_num_events_to_rescale : 200;
    fs.krank = d_krank; fs.mterms = c->d_mterms;
    hipLaunchKernelGGL(abea_align_kernel, dim3((unsigned)m), dim3(64), 0, sl.stream,
                       (const abea_read_desc*)d_desc, (const float*)d_evm, (const abea_kpar_t*)d_kpar, d_trace,
                       (uint32_t*)(d_dn + sl.o_codes), (abea_pair_t*)nullptr, (int32_t*)(d_dn + sl.o_np),
                       (abea_read_diag*)(d_dn + sl.o_diag), (unsigned long long*)nullptr, (int64_t*)nullptr, fs);
    HIP_TRY(hipEventRecord(sl.t4, sl.stream));
    hipLaunchKernelGGL(abea_copy_out_kernel, dim3((unsigned)std::min<size_t>(512, (sl.dn_copy / 16 + 255) / 256)), dim3(256), 0, sl.stream,
                       (const uint4*)d_dn, (uint4*)sl.dn.p, sl.dn_copy / 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.e_done, sl.stream));
    S.st.d2h_bytes += sl.dn_copy;
    S.st.fill_launches += 1;
    sl.staged = true;
    return ABEA_OK;
}

/* the outputs of a read that has a signal but is not aligned (align_single's guards, f5c.c:811-830; scaling_single f5c.c:786-794) */
void not_aligned(const abea_chain_job* J, int32_t i, const abea_scalings_t& est) {
    J->n_pairs[i] = 0;
    if (J->diag) { abea_read_diag dg; memset(&dg, 0, sizeof dg); dg.max_score = -__builtin_inff(); dg.flags = ABEA_RF_SKIPPED; J->diag[i] = dg; }
    J->scalings[i] = est;
    J->events_per_base[i] = 0.0;
    J->read_stat_flag[i] |= ABEA_FAILED_ALIGNMENT;
    J->n_event_alignment[i] = 0;
}

/* event_t from the 12-byte records the tables crossed PCIe in: {uint32 start, float mean, float stdv}.  The events of a read tile
 * [0, n_samples) (events.c:466-513: event j runs from peak j-1 to peak j, the first from 0, the last to n_samples), so the end of an
 * event is the start of its successor IN DETECTION ORDER — the next record, or the previous one in an RNA table, which event_single
 * reversed (f5c.c:711-719) — and length = (float)(end - start), the very expression of events.c:498. */
void unpack_events(const uint32_t* rec, size_t ne, uint64_t n_samples, bool rna, abea_event_t* out) {
    for (size_t p = 0; p < ne; ++p) {
        const uint64_t start = rec[3 * p];
        const uint64_t end = !rna ? (p + 1 < ne ? (uint64_t)rec[3 * (p + 1)] : n_samples) : (p > 0 ? (uint64_t)rec[3 * (p - 1)] : n_samples);
        /* three 8-byte words: event_t is 20 bytes of fields + 4 of tail padding, written as zero (a caller that dumps or compares
         * whole structs must not see stack garbage there) */
        const float length = (float)(end - start);
        uint32_t lb; memcpy(&lb, &length, 4);
        const uint64_t w[3] = {start, (uint64_t)lb | ((uint64_t)rec[3 * p + 1] << 32), (uint64_t)rec[3 * p + 2]};
        memcpy(out + p, w, sizeof w);
    }
}

/* ------------------------------------------------------------------ retire: the caller's db */
int slot_finish(chain_state& S, abea_chain_slot& sl) {
    if (!sl.busy) return ABEA_OK;
    int rc = stage_align(S, sl);
    if (rc) return rc;
    const abea_chain_job* J = S.J;
    double t0 = abea_now_ms();
    S.log("wait", sl.chunk_no, sl.m);
    HIP_TRY(hipEventSynchronize(sl.e_done));
    HIP_TRY(hipEventSynchronize(sl.e_tab));
    S.st.wait_ms += abea_now_ms() - t0;
    t0 = abea_now_ms();
    S.log("scatter", sl.chunk_no, sl.m, sl.n_ev);
    const int32_t m = sl.m;
    const abea_event_t* h_ev = (const abea_event_t*)sl.tab.p;
    const uint32_t* h_rec = (const uint32_t*)sl.tab.p;                        /* packed form: 3 words per event */
    const bool packed = sl.packed, rna = J->rna != 0;
    /* the result block exists in process mode only (event_db alone reads none of it) */
    const uint8_t* const dnb = J->align ? sl.dn.u8() : nullptr;
    auto at = [&](size_t off) { return dnb ? dnb + off : nullptr; };
    const int32_t* npairs = (const int32_t*)at(sl.o_np);
    const abea_read_diag* diag = (const abea_read_diag*)at(sl.o_diag);
    const uint32_t* codes = (const uint32_t*)at(sl.o_codes);
    const abea_scalings_t* sc = (const abea_scalings_t*)at(sl.o_sc);
    const double* epb = (const double*)at(sl.o_epb);
    const int32_t* flag = (const int32_t*)at(sl.o_flag);
    const int32_t* nal = (const int32_t*)at(sl.o_nal);
    const double* var64 = (const double*)at(sl.o_var);
    const uint8_t* kcnt = at(sl.o_kcnt);
    const bool want_sc = J->read != nullptr;
    abea_parallel_for(S.c, m, 1, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; ++j) {
            const int32_t i = sl.rd[(size_t)j];
            const size_t ne = (size_t)sl.ne[(size_t)j];
            if (ne == 0 && ((const int32_t*)sl.cnt.u8())[j] > sl.cap[(size_t)j]) continue;        /* redone later: outputs untouched */
            /* getevents() returns a malloc()ed table (events.c:562-582); released by the caller like free_db_tmp does */
            abea_event_t* t = (abea_event_t*)malloc(std::max<size_t>(ne, 1) * sizeof(abea_event_t));
            if (!t) { S.oom.store(true); continue; }
            if (!packed) memcpy(t, h_ev + sl.out_ptr[(size_t)j], ne * sizeof(abea_event_t));
            else unpack_events(h_rec + (size_t)sl.out_ptr[(size_t)j] * 3, ne, (uint64_t)sl.ns32[(size_t)j], rna, t);
            J->events[i] = t; J->n_events[i] = ne;
            if (J->scalings_estimated && want_sc) J->scalings_estimated[i] = sl.est[(size_t)j];
            if (!J->align) { if (J->scalings) J->scalings[i] = sl.est[(size_t)j]; continue; }
            const int32_t L = sl.rl[(size_t)j], K = sl.nk[(size_t)j];
            /* event_single malloc()s the pair buffer of every read with a signal (f5c.c:722-725) */
            if (J->pairs && ne > 0) {
                J->pairs[i] = (abea_pair_t*)malloc(sizeof(abea_pair_t) * (ne + (size_t)L));
                if (!J->pairs[i]) { S.oom.store(true); continue; }
            }
            if (!sl.run[(size_t)j]) { not_aligned(J, i, sl.est[(size_t)j]); continue; }
            const abea_read_desc& d = sl.descs[(size_t)j];
            const int32_t np = npairs[j];
            J->n_pairs[i] = np;
            if (J->diag) J->diag[i] = diag[j];
            if (J->pairs && np > 0) abea_expand_walk_codes(codes + d.code_off, np, K - 1, diag[j].best_event, J->pairs[i]);
            if (np > 0) {                                      /* scaling_single -> postalign malloc()s the map of an aligned read (f5c.c:746) */
                abea_index_pair_t* map = (abea_index_pair_t*)malloc(sizeof(abea_index_pair_t) * (size_t)K);
                if (!map) { S.oom.store(true); continue; }
                if (abea_expand_kmer_counts_to_map(kcnt + d.kmer_off, K, diag[j].best_event, map) != ABEA_OK)
                    abea_expand_walk_codes_to_map(codes + d.code_off, np, K - 1, diag[j].best_event, map);    /* a k-mer of 255+ events */
                J->base_to_event_map[i] = map;
            }
            abea_scalings_t o = sc[j];
            abea_apply_log_var(o, var64[j]);                                          /* align.c:758-760 */
            J->scalings[i] = o;
            J->events_per_base[i] = epb[j];
            J->read_stat_flag[i] = flag[j];
            J->n_event_alignment[i] = nal[j];
        }
    });
    S.st.unflatten_ms += abea_now_ms() - t0;
    if (S.oom.load()) return abea_fail(ABEA_ENOMEM, "malloc of a per-read output buffer failed");
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, sl.t0, sl.t1)); S.st.event_ms += ms;
    if (J->align) {
        HIP_TRY(hipEventElapsedTime(&ms, sl.t2, sl.t3)); S.st.pre_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, sl.t3, sl.t4)); S.st.fill_ms += ms;
        for (int32_t j = 0; j < m; ++j) if (sl.run[(size_t)j]) S.st.sum_pairs += npairs[j];
    }
    if (S.origin) {
        float a = 0, e = 0;
        if (hipEventElapsedTime(&a, S.origin, sl.t0) == hipSuccess && hipEventElapsedTime(&e, S.origin, J->align ? sl.t4 : sl.t2) == hipSuccess) {
            S.spans.emplace_back(a, e);
            if (S.trace) fprintf(stderr, "[abea chain dev %d] gpu chunk %2d  kernels %9.3f .. %9.3f ms  reads %d\n", S.c->device, sl.chunk_no, a, e, sl.m);
        }
    }
    S.st.n_sub_batches += 1;
    sl.busy = false;
    S.log("retired", sl.chunk_no, sl.m);
    return ABEA_OK;
}

/* on every exit nothing of this call may stay in flight */
struct chain_guard {
    abea_ctx* c; int n_slots;
    ~chain_guard() {
        for (int q = 0; q < n_slots && q < (int)c->chain_slots.size(); ++q) {
            abea_chain_slot* s = c->chain_slots[(size_t)q];
            if (s && s->busy) { hipStreamSynchronize(s->stream); hipStreamSynchronize(s->dstream); s->busy = false; }
        }
    }
};

/* ------------------------------------------------------------------ overflowed tables: synchronously, from the int16 staging */
int redo_overflowed(chain_state& S) {
    abea_ctx* c = S.c;
    const abea_chain_job* J = S.J;
    const bool want_sc = J->read != nullptr;
    for (redo_read& r : S.redo) {
        const int32_t i = r.idx;
        const int64_t ns = (int64_t)r.samples.size();
        const int32_t L = want_sc ? J->read_len[i] : (int32_t)c->k;
        const size_t n_sig = (size_t)((ns + 7) / 8 * 8);
        const size_t o_seq = align_up(n_sig * 2, 256), u_end = align_up(o_seq + (size_t)L + 1, 256);
        const int32_t cap = r.n_events;                      /* the true count */
        pinned_buf up, dn;
        int rc = up.need(u_end);
        if (!rc) rc = dn.need(align_up((size_t)cap * sizeof(abea_event_t), 256) + 512);
        if (rc) { up.release(); dn.release(); return rc; }
        memset(up.p, 0, u_end);
        memcpy(up.p, r.samples.data(), (size_t)ns * 2);
        if (want_sc) memcpy(up.u8() + o_seq, J->read[i], (size_t)L); else memset(up.u8() + o_seq, 'A', (size_t)L);
        uint8_t* p = c->arena;
        uint8_t* d_up = p;                               p += u_end;
        abea_event_t* d_ev = (abea_event_t*)p;           p += align_up((size_t)cap * sizeof(abea_event_t), 256);
        int32_t* d_ne = (int32_t*)p;                     p += 256;
        abea_scalings_t* d_sc = (abea_scalings_t*)p;     p += 256;
        const size_t used = align_up((size_t)(p - c->arena), 4096);
        if (used + ((size_t)64 << 20) > c->arena_bytes) { up.release(); dn.release(); return abea_fail(ABEA_ENOMEM, "read %d does not fit the arena", i); }
        const int64_t zero = 0; const int32_t ns32 = (int32_t)ns;
        const float sc3[3] = {J->offset[i], J->range[i], J->digitisation[i]};
        abea_signal_batch sb;
        memset(&sb, 0, sizeof sb);
        sb.n_reads = 1; sb.sig_ptr = &zero; sb.n_samples = &ns32; sb.scaling = sc3; sb.event_ptr = &zero; sb.event_cap = &cap;
        sb.read_ptr = &zero; sb.read_len = &L; sb.signal = (const int16_t*)d_up; sb.reads = want_sc ? (const char*)(d_up + o_seq) : nullptr;
        sb.events = d_ev; sb.n_events = d_ne; sb.scalings = want_sc ? d_sc : nullptr; sb.rna = J->rna;
        abea_ev_exec X;
        X.stream = c->stream; X.scratch = c->arena + used; X.scratch_bytes = c->arena_bytes - used;
        X.h_pinned = (void**)&c->h_desc; X.h_cap = &c->h_desc_cap; X.async = false; X.e0 = c->ev[0]; X.e1 = c->ev[1];
        hipError_t e = hipMemcpyAsync(d_up, up.p, u_end, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) rc = abea_detect_events_on(c, &sb, X);
        int32_t ne = 0; abea_scalings_t est; memset(&est, 0, sizeof est);
        if (e == hipSuccess && !rc) {
            e = hipMemcpyAsync(dn.p, d_ev, (size_t)cap * sizeof(abea_event_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dn.u8() + align_up((size_t)cap * sizeof(abea_event_t), 256), d_ne, 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && want_sc) e = hipMemcpyAsync(dn.u8() + align_up((size_t)cap * sizeof(abea_event_t), 256) + 64, d_sc, sizeof est, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            ne = *(const int32_t*)(dn.u8() + align_up((size_t)cap * sizeof(abea_event_t), 256));
            if (want_sc) est = *(const abea_scalings_t*)(dn.u8() + align_up((size_t)cap * sizeof(abea_event_t), 256) + 64);
        }
        if (e != hipSuccess) rc = abea_fail(ABEA_EHIP, "redo of read %d: %s", i, hipGetErrorString(e));
        if (!rc && ne != cap) rc = abea_fail(ABEA_EHIP, "internal: read %d gave %d events, then %d", i, cap, ne);
        if (!rc) {
            abea_event_t* t = (abea_event_t*)malloc(std::max<size_t>((size_t)ne, 1) * sizeof(abea_event_t));
            if (!t) rc = abea_fail(ABEA_ENOMEM, "malloc of an event table failed");
            else {
                memcpy(t, dn.p, (size_t)ne * sizeof(abea_event_t));
                J->events[i] = t; J->n_events[i] = (uint64_t)ne;
                if (J->scalings_estimated && want_sc) J->scalings_estimated[i] = est;
                if (J->scalings) J->scalings[i] = est;
            }
        }
        up.release(); dn.release();
        if (rc) return rc;
        S.st.event_ms += c->stats.event_ms;
    }
    if (!J->align || S.redo.empty()) return ABEA_OK;
    /* align_db + scaling_db for them through the host entry (their tables are host tables now) */
    const int32_t n = (int32_t)S.redo.size();
    std::vector<const char*> rd((size_t)n); std::vector<int32_t> rl((size_t)n), npairs((size_t)n), flag((size_t)n), nal((size_t)n);
    std::vector<const abea_event_t*> ev((size_t)n); std::vector<uint64_t> nev((size_t)n);
    std::vector<abea_scalings_t> sc((size_t)n); std::vector<double> epb((size_t)n);
    std::vector<abea_pair_t*> pr((size_t)n, nullptr); std::vector<abea_index_pair_t*> mp((size_t)n, nullptr);
    std::vector<abea_read_diag> dg((size_t)n);
    for (int32_t q = 0; q < n; ++q) {
        const int32_t i = S.redo[(size_t)q].idx;
        rd[(size_t)q] = J->read[i]; rl[(size_t)q] = J->read_len[i]; ev[(size_t)q] = J->events[i]; nev[(size_t)q] = J->n_events[i];
        sc[(size_t)q] = J->scalings[i]; flag[(size_t)q] = J->read_stat_flag[i];
        const int32_t K = J->read_len[i] - (int32_t)c->k + 1;
        if (J->pairs) { pr[(size_t)q] = (abea_pair_t*)malloc(sizeof(abea_pair_t) * ((size_t)nev[(size_t)q] + (size_t)rl[(size_t)q])); J->pairs[i] = pr[(size_t)q]; }
        mp[(size_t)q] = K > 0 ? (abea_index_pair_t*)malloc(sizeof(abea_index_pair_t) * (size_t)K) : nullptr;
        J->base_to_event_map[i] = mp[(size_t)q];
        if ((J->pairs && !pr[(size_t)q]) || (K > 0 && !mp[(size_t)q])) return abea_fail(ABEA_ENOMEM, "malloc failed for read %d", i);
    }
    abea_host_batch H;
    memset(&H, 0, sizeof H);
    H.n_reads = n; H.read = rd.data(); H.read_len = rl.data(); H.events = ev.data(); H.n_events = nev.data(); H.scalings = sc.data();
    H.pairs = J->pairs ? pr.data() : nullptr; H.n_pairs = npairs.data(); H.diag = dg.data();
    H.base_to_event_map = mp.data(); H.scalings_out = sc.data(); H.events_per_base = epb.data(); H.read_stat_flag = flag.data();
    H.n_event_alignment = nal.data(); H.min_num_events_to_rescale = J->min_num_events_to_rescale;
    const abea_stats keep = S.c->stats;
    int rc = abea_host_batch_locked(c, &H);
    const abea_stats hst = c->stats;
    c->stats = keep;
    if (rc) return rc;
    S.st.pre_ms += hst.pre_ms; S.st.fill_ms += hst.fill_ms; S.st.n_reads_gpu += hst.n_reads_gpu; S.st.n_reads_skipped += hst.n_reads_skipped;
    for (int32_t q = 0; q < n; ++q) {
        const int32_t i = S.redo[(size_t)q].idx;
        J->n_pairs[i] = npairs[(size_t)q]; if (J->diag) J->diag[i] = dg[(size_t)q];
        J->scalings[i] = sc[(size_t)q]; J->events_per_base[i] = epb[(size_t)q]; J->read_stat_flag[i] = flag[(size_t)q];
        J->n_event_alignment[i] = nal[(size_t)q];
        if (npairs[(size_t)q] <= 0 && mp[(size_t)q]) { free(mp[(size_t)q]); J->base_to_event_map[i] = nullptr; }
    }
    return ABEA_OK;
}

}  // namespace

/* ------------------------------------------------------------------ the pipeline on one device context */
int abea_chain_run(abea_ctx* c, const abea_chain_job* J, const int32_t* mine, int32_t n_mine, abea_stats* st_out) {
    const double t_start = abea_now_ms();
    HIP_TRY(hipSetDevice(c->device));
    chain_state S;
    S.c = c; S.J = J; S.opt = read_chain_opts();
    S.trace = getenv("ABEA_HOST_TRACE") != nullptr; S.t_origin = t_start;
    memset(&S.st, 0, sizeof S.st);
    S.st.arena_bytes = c->arena_bytes; S.st.n_devices = 1;
    const bool want_sc = J->read != nullptr;
    /* ---- reads with a signal, longest first (f5c.c:684: reads without one get et.n = 0, et.event = NULL, f5c.c:727-731) ---- */
    std::vector<int32_t> todo; todo.reserve((size_t)n_mine);
    for (int32_t q = 0; q < n_mine; ++q) {
        const int32_t i = mine ? mine[q] : q;
        if (J->n_samples[i] <= 0) {
            if (J->align) {                                   /* align_single's first guard (f5c.c:812, 826-828) + scaling_single (f5c.c:786-794) */
                abea_scalings_t z; memset(&z, 0, sizeof z);
                not_aligned(J, i, z);
                if (J->scalings_estimated) J->scalings_estimated[i] = z;
                ++S.st.n_reads_skipped;
            }
            continue;
        }
        if (!J->rawptr[i]) return abea_fail(ABEA_EINVAL, "read %d: null signal", i);
        if (J->n_samples[i] > (int64_t)INT32_MAX - 64) return abea_fail(ABEA_EINVAL, "read %d: %" PRId64 " samples", i, J->n_samples[i]);
        if (want_sc && (!J->read[i] || J->read_len[i] < (int32_t)c->k)) return abea_fail(ABEA_EINVAL, "read %d: sequence shorter than k", i);
        todo.push_back(i);
    }
    std::stable_sort(todo.begin(), todo.end(), [&](int32_t a, int32_t b) { return J->n_samples[a] > J->n_samples[b]; });
    const int n_slots = S.opt.n_slots;
    {
        std::lock_guard<std::mutex> lk(c->slots_mu);
        if (c->chain_slots.size() < (size_t)ABEA_MAX_SLOTS) c->chain_slots.resize((size_t)ABEA_MAX_SLOTS, nullptr);
    }
    for (int q = 0; q < n_slots; ++q) {
        if (!c->chain_slots[(size_t)q]) { abea_chain_slot* s = nullptr; const int rc = slot_make(&s); if (rc) return rc; c->chain_slots[(size_t)q] = s; }
        c->chain_slots[(size_t)q]->busy = false;
    }
    chain_guard guard{c, n_slots};
    struct origin_event { hipEvent_t e = nullptr; ~origin_event() { if (e) hipEventDestroy(e); } } origin;
    if (hipEventCreate(&origin.e) == hipSuccess && hipEventRecord(origin.e, c->chain_slots[0]->stream) == hipSuccess) S.origin = origin.e;
    const size_t slot_arena = c->arena_bytes / (size_t)n_slots / 4096 * 4096;
    S.st.host_threads = abea_pool_threads(c);
    S.st.setup_ms = abea_now_ms() - t_start;
    /* ---- chunks: closed at chunk_samples samples (the first two a quarter / half of that) once they hold reads_min reads, at
     *      reads_max reads, or when the next read's upper bound would not fit the slot's share of the arena ---- */
    std::vector<std::pair<size_t, size_t>> chunks;            /* [pos, end) into todo */
    for (size_t pos = 0; pos < todo.size();) {
        const int ramp = chunks.empty() ? 4 : chunks.size() == 1 ? 2 : 1;
        const size_t want_s = S.opt.chunk_samples / (size_t)ramp;
        const int32_t want_r = std::max(1, S.opt.reads_min / ramp);
        size_t samples = 0, end = pos;
        chunk_charge charge; charge.align = J->align;
        while (end < todo.size()) {
            const int32_t i = todo[end];
            const int64_t ns = J->n_samples[i];
            const int32_t L = want_sc ? J->read_len[i] : (int32_t)c->k;
            const int32_t cap = (int32_t)std::min<size_t>((size_t)ns / S.opt.cap_div + 16, INT32_MAX / 2);
            if (charge.with(ns, cap, L, L - (int32_t)c->k + 1, false) > slot_arena) {
                if (end > pos) break;
                return abea_fail(ABEA_ENOMEM, "read %d (%" PRId64 " samples) does not fit a %zu-byte share of the arena: the detector's scratch is "
                                 "laid out in 64-lane waves as long as their longest read (%zu bytes for this one)", i, ns, slot_arena,
                                 chunk_charge::wave_bytes((size_t)ns + 1, cap, std::max(L - (int32_t)c->k + 1, 1)));
            }
            charge.with(ns, cap, L, L - (int32_t)c->k + 1, true); samples += (size_t)ns; ++end;
            const int32_t cnt = (int32_t)(end - pos);
            if ((cnt >= want_r && samples >= want_s) || cnt >= S.opt.reads_max) break;
        }
        chunks.emplace_back(pos, end);
        pos = end;
    }
    /* ---- the pipeline, driven by what has completed (round 6).  Round 5 issued stage D of chunk c + 1 and then BLOCKED on the counts
     *      of chunk c: two chunks in flight, and when the pair finished the GPU idled until the next chunk was flattened and its
     *      signal had crossed PCIe (GPU-clock timeline profiles/r06/chain_timeline_*: kernels 40-71 ms, 81-110 ms, 127-157 ms ...).
     *      Now the caller's thread never waits while it could work: (1) every chunk whose counts are back gets its stage A at once,
     *      (2) while a slot is free the next chunk is flattened and sent up — up to `depth` chunks ahead of the detector, so a
     *      signal is in HBM before the kernels that read it can start —, (3) finished chunks are retired, oldest first, and only
     *      when none of the three is possible does the thread block, on the oldest chunk in flight. ---- */
    /* pinned staging sized once for the largest chunk of the call (the slot a chunk lands on depends on what has completed: growing
     * a slot's blocks when a larger chunk arrives re-pinned hundreds of MB in the middle of the pipeline, 10-15 ms a time) */
    {
        size_t up_max = 0, tab_max = 0, cnt_max = 0, desc_max = 0, dn_max = 0;
        for (const std::pair<size_t, size_t>& ck : chunks) {
            size_t n_sig = 0, n_seq = 0, n_slot = 0, n_code = 0, n_kmer = 0;
            for (size_t q = ck.first; q < ck.second; ++q) {
                const int32_t i = todo[q];
                const int64_t ns = J->n_samples[i];
                n_sig += (size_t)((ns + 7) / 8 * 8);
                n_seq += align_up((size_t)(want_sc ? J->read_len[i] : (int32_t)c->k) + 1, 16);
                const size_t cap = std::min<size_t>((size_t)ns / S.opt.cap_div + 16, INT32_MAX / 2);
                n_slot += cap;
                if (J->align) {                                   /* walk codes (2 bit per step, <= E + K steps) and the count byte per k-mer */
                    const size_t K = (size_t)std::max(J->read_len[i] - (int32_t)c->k + 1, 0);
                    n_code += (cap + K) / 16 + 8; n_kmer += K;
                }
            }
            const size_t m = ck.second - ck.first;
            up_max = std::max(up_max, align_up(align_up(n_sig * 2, 256) + n_seq, 256));
            tab_max = std::max(tab_max, align_up((n_slot + 3 * m) * (S.opt.packed ? 12 : sizeof(abea_event_t)), 16) + 256);
            cnt_max = std::max(cnt_max, align_up(align_up(m * 4, 256) + m * sizeof(abea_scalings_t), 256));
            desc_max = std::max(desc_max, m * 24 + 256 + (J->align ? align_up(m * sizeof(abea_read_desc), 256) + align_up(m * sizeof(abea_scalings_t), 256) + align_up(m * 4, 256) : 0));
            if (J->align) dn_max = std::max(dn_max, m * (4 + sizeof(abea_read_diag) + sizeof(abea_scalings_t) + 8 + 4 + 4 + 8) + n_code * 4 + n_kmer + 10 * 256);
        }
        for (int q = 0; q < std::min<int>(n_slots, (int)chunks.size()); ++q) {
            abea_chain_slot& sl = *c->chain_slots[(size_t)q];
            int rc = sl.up.need(up_max);
            if (!rc) rc = sl.tab.need(tab_max);
            if (!rc) rc = sl.cnt.need(cnt_max);
            if (!rc) rc = sl.desc.need(desc_max);
            if (!rc && dn_max) rc = sl.dn.need(dn_max);
            if (rc) return rc;
        }
    }
    S.st.setup_ms = abea_now_ms() - t_start;
    std::vector<int> inflight;                                /* slot numbers, in issue order */
    std::vector<char> slot_busy((size_t)n_slots, 0);
    size_t next_chunk = 0;
    const int depth = std::max(1, std::min(n_slots, S.opt.depth));
    auto ready = [](hipEvent_t e) { return hipEventQuery(e) == hipSuccess; };
    while (next_chunk < chunks.size() || !inflight.empty()) {
        bool progressed = false;
        for (int q : inflight) {                              /* (1) counts are back: tables down, alignment */
            abea_chain_slot& sl = *c->chain_slots[(size_t)q];
            if (!sl.staged && ready(sl.e_cnt)) { const int rc = stage_align(S, sl); if (rc) return rc; progressed = true; }
        }
        if (next_chunk < chunks.size() && (int)inflight.size() < depth) {      /* (2) the next chunk up */
            int q = 0;
            while (slot_busy[(size_t)q]) ++q;
            const std::pair<size_t, size_t> ck = chunks[next_chunk];
            const int rc = stage_detect(S, *c->chain_slots[(size_t)q], todo.data() + ck.first, (int32_t)(ck.second - ck.first), (int)next_chunk,
                                        c->arena + (size_t)q * slot_arena, slot_arena);
            if (rc) return rc;
            slot_busy[(size_t)q] = 1; inflight.push_back(q); ++next_chunk;
            continue;
        }
        if (!inflight.empty()) {                              /* (3) retire the oldest chunk if it is complete */
            abea_chain_slot& sl = *c->chain_slots[(size_t)inflight.front()];
            if (sl.staged && ready(sl.e_done) && ready(sl.e_tab)) {
                const int rc = slot_finish(S, sl);
                if (rc) return rc;
                slot_busy[(size_t)inflight.front()] = 0; inflight.erase(inflight.begin());
                continue;
            }
        }
        if (progressed) continue;
        /* nothing to do but wait: for the counts of the oldest chunk that has none yet, else for the oldest chunk's results */
        double t0 = abea_now_ms();
        abea_chain_slot* waitfor = nullptr;
        for (int q : inflight) if (!c->chain_slots[(size_t)q]->staged) { waitfor = c->chain_slots[(size_t)q]; break; }
        if (waitfor) HIP_TRY(hipEventSynchronize(waitfor->e_cnt));
        else {
            abea_chain_slot& sl = *c->chain_slots[(size_t)inflight.front()];
            HIP_TRY(hipEventSynchronize(sl.e_done));
            HIP_TRY(hipEventSynchronize(sl.e_tab));
        }
        S.st.wait_ms += abea_now_ms() - t0;
    }
    S.st.gpu_busy_ms = interval_union_ms(S.spans);
    int rc = redo_overflowed(S);
    if (rc) return rc;
    S.st.host_ms = S.st.flatten_ms + S.st.unflatten_ms;
    S.st.total_ms = abea_now_ms() - t_start;
    *st_out = S.st;
    return ABEA_OK;
}
