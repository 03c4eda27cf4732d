/* abea_process.cpp — host-buffer entries for the callers either side of align_db (rows N2 / N3 of SURVEY §8f):
 *
 *   abea_events_batch_host   event_db  = pthread_db(event_single)   src/f5c.c:682-734
 *   abea_process_batch_host  process_db_rsq = event_db -> align_db -> scaling_db   src/resquiggle.c:283-315
 *                            (process_db runs the same three steps first, src/f5c.c:907-936)
 *   abea_rsq_format_batch    output_db_rsq's loop over the batch   src/resquiggle.c:319-449
 *
 * f5c holds a read's raw signal as FLOAT ADC counts (signal_t.rawptr, src/f5c.h:276-286: the slow5 / fast5 readers widen
 * int16) and event_single converts it to pA in place (f5c.c:693-696) before getevents().  The device detector takes the
 * int16 counts and does the same float arithmetic itself, so the flatten loop narrows the floats back (2 bytes per sample
 * over PCIe; a sample that is not an integer in int16 range is refused: it cannot have come from an ADC), optionally writes
 * the pA values into the caller's buffer as the reference does, and the event tables come back as malloc()ed event_t arrays
 * exactly where getevents() would have put them.
 *
 * Both entries are views over ONE chunk pipeline (abea_chain.cpp, round 5): signal up, detector, tables down and — for the
 * chain — the alignment with scaling_single fused reading the event means from the tables where the detector left them in
 * HBM.  On a multi-device context the reads are split over the devices (longest-processing-time-first on the sample count)
 * and every device runs the pipeline on its share from its own host thread.  No CPU fallback: every number in the outputs
 * was computed on the GPU.
 */
#include <atomic>
#include <cinttypes>
#include <string>
#include <thread>
#include "abea_internal.h"

namespace {

/* the pipeline over the devices of a context: one device, or an LPT split on the sample count with one host thread each */
int chain_over_devices(abea_ctx* c, const abea_chain_job* J, abea_stats* st_out) {
    const double t0 = abea_now_ms();
    abea_host_prepare_pools(c);
    if (c->children.empty()) return abea_chain_run(c, J, nullptr, J->n_reads, st_out);
    const int32_t n = J->n_reads, nd = (int32_t)c->children.size();
    std::vector<int64_t> weight((size_t)n);
    for (int32_t i = 0; i < n; ++i) weight[(size_t)i] = std::max<int64_t>(J->n_samples[i], 0);
    std::vector<int32_t> bin_of((size_t)n);
    int rc = abea_lpt_split(weight.data(), n, nd, bin_of.data());
    if (rc) return rc;
    std::vector<std::vector<int32_t>> share((size_t)nd);
    for (int32_t i = 0; i < n; ++i) share[(size_t)bin_of[(size_t)i]].push_back(i);
    std::vector<int> rcs((size_t)nd, ABEA_OK);
    std::vector<abea_stats> sts((size_t)nd);
    std::vector<std::string> errs((size_t)nd);
    std::vector<std::thread> th;
    for (int32_t d = 0; d < nd; ++d)
        th.emplace_back([&, d]() {
            memset(&sts[(size_t)d], 0, sizeof(abea_stats));
            rcs[(size_t)d] = abea_chain_run(c->children[(size_t)d], J, share[(size_t)d].data(), (int32_t)share[(size_t)d].size(), &sts[(size_t)d]);
            if (rcs[(size_t)d]) errs[(size_t)d] = abea_last_error();
        });
    for (auto& t : th) t.join();
    abea_stats st; memset(&st, 0, sizeof st);
    for (int32_t d = 0; d < nd; ++d) {
        if (rcs[(size_t)d]) return abea_fail(rcs[(size_t)d], "device %d: %s", c->children[(size_t)d]->device, errs[(size_t)d].c_str());
        const abea_stats& b = sts[(size_t)d];
        st.pre_ms = std::max(st.pre_ms, b.pre_ms); st.fill_ms = std::max(st.fill_ms, b.fill_ms); st.event_ms = std::max(st.event_ms, b.event_ms);
        st.flatten_ms = std::max(st.flatten_ms, b.flatten_ms); st.unflatten_ms = std::max(st.unflatten_ms, b.unflatten_ms);
        st.wait_ms = std::max(st.wait_ms, b.wait_ms); st.plan_ms = std::max(st.plan_ms, b.plan_ms); st.setup_ms = std::max(st.setup_ms, b.setup_ms);
        st.host_ms = std::max(st.host_ms, b.host_ms); st.gpu_busy_ms = std::max(st.gpu_busy_ms, b.gpu_busy_ms);
        st.n_reads_gpu += b.n_reads_gpu; st.n_reads_skipped += b.n_reads_skipped; st.n_sub_batches += b.n_sub_batches;
        st.sum_events += b.sum_events; st.sum_bands += b.sum_bands; st.sum_pairs += b.sum_pairs; st.fill_launches += b.fill_launches;
        st.arena_bytes += b.arena_bytes; st.bytes_ref += b.bytes_ref; st.bytes_min += b.bytes_min; st.bytes_moved += b.bytes_moved;
        st.h2d_bytes += b.h2d_bytes; st.d2h_bytes += b.d2h_bytes; st.host_threads += b.host_threads;
    }
    st.n_devices = nd;
    st.total_ms = abea_now_ms() - t0;
    *st_out = st;
    return ABEA_OK;
}

}  // namespace

extern "C" int abea_events_batch_host(abea_ctx* c, const abea_events_host_batch* B) {
    if (!c || !B) return abea_fail(ABEA_EINVAL, "null argument");
    if (B->n_reads < 0) return abea_fail(ABEA_EINVAL, "n_reads < 0");
    if (B->n_reads == 0) return ABEA_OK;
    if (!B->rawptr || !B->n_samples || !B->offset || !B->range || !B->digitisation || !B->events || !B->n_events)
        return abea_fail(ABEA_EINVAL, "abea_events_batch_host: null array");
    if (B->scalings && (!B->read || !B->read_len))
        return abea_fail(ABEA_EINVAL, "abea_events_batch_host: scalings need the read sequences");
    ABEA_API_ENTER(c, "abea_events_batch_host");
    for (int32_t i = 0; i < B->n_reads; ++i) { B->events[i] = nullptr; B->n_events[i] = 0; }   /* before any exit: the error path free()s these */
    abea_chain_job J;
    memset(&J, 0, sizeof J);
    J.n_reads = B->n_reads; J.rawptr = B->rawptr; J.n_samples = B->n_samples; J.offset = B->offset; J.range = B->range;
    J.digitisation = B->digitisation; J.read = B->scalings ? B->read : nullptr; J.read_len = B->read_len; J.rna = B->rna;
    J.signal_to_pa_in_place = B->signal_to_pa_in_place; J.events = B->events; J.n_events = B->n_events; J.scalings = B->scalings;
    J.align = false;
    abea_stats st;
    const int rc = chain_over_devices(c, &J, &st);
    if (rc) {                                            /* nothing half-built is handed back */
        for (int32_t i = 0; i < B->n_reads; ++i) { free(B->events[i]); B->events[i] = nullptr; B->n_events[i] = 0; }
        return rc;
    }
    c->stats = st;
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
    for (int32_t i = 0; i < n; ++i) {                    /* before any exit: the error path free()s these */
        P->events[i] = nullptr; P->n_events[i] = 0;
        if (P->pairs) P->pairs[i] = nullptr;
        P->base_to_event_map[i] = nullptr; P->n_pairs[i] = 0;
    }
    abea_chain_job J;
    memset(&J, 0, sizeof J);
    J.n_reads = n; J.rawptr = P->rawptr; J.n_samples = P->n_samples; J.offset = P->offset; J.range = P->range;
    J.digitisation = P->digitisation; J.read = P->read; J.read_len = P->read_len; J.rna = P->rna;
    J.signal_to_pa_in_place = P->signal_to_pa_in_place; J.events = P->events; J.n_events = P->n_events;
    J.scalings = P->scalings; J.scalings_estimated = P->scalings_estimated; J.align = true;
    J.pairs = P->pairs; J.n_pairs = P->n_pairs; J.diag = P->diag; J.base_to_event_map = P->base_to_event_map;
    J.events_per_base = P->events_per_base; J.read_stat_flag = P->read_stat_flag; J.n_event_alignment = P->n_event_alignment;
    J.min_num_events_to_rescale = P->min_num_events_to_rescale;
    abea_stats st;
    const int rc = chain_over_devices(c, &J, &st);
    if (rc) {
        for (int32_t i = 0; i < n; ++i) {
            free(P->events[i]); P->events[i] = nullptr; P->n_events[i] = 0;
            if (P->pairs) { free(P->pairs[i]); P->pairs[i] = nullptr; }
            free(P->base_to_event_map[i]); P->base_to_event_map[i] = nullptr;
        }
        return rc;
    }
    c->stats = st;
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
