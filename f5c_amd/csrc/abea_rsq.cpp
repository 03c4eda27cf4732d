/* abea_rsq.cpp — the text a resquiggle caller prints for one read (row N3 of SURVEY §8f).
 *
 * `f5c resquiggle` is event_single -> align_db -> scaling_single (process_db_rsq, src/resquiggle.c:283-315: the three
 * device entries of this library chained) followed by output_db_rsq (src/resquiggle.c:319-449), which turns
 * base_to_event_map + the event table into one TSV line per k-mer or one PAF line per read.  This file is that last
 * step for one read: host-only string work, no device code; file and BLOW5/FASTQ I/O stay with the caller.
 */
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include "../../include/abea.h"

namespace {
void appendf(std::string& s, const char* fmt, ...) {
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    const int len = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (len < 0) return;
    if ((size_t)len < sizeof buf) { s.append(buf, (size_t)len); return; }
    std::string big((size_t)len + 1, '\0');
    va_start(ap, fmt);
    vsnprintf(&big[0], big.size(), fmt, ap);
    va_end(ap);
    s.append(big.data(), (size_t)len);
}
}  // namespace

extern "C" int64_t abea_rsq_format(char* out, size_t cap, int fmt, const char* read_id, int32_t read_len,
                                   uint32_t kmer_size, abea_index_pair_t* base_to_event_map,
                                   const abea_event_t* events, int64_t n_samples, float scale, float shift, int rna) {
    if (!read_id || !base_to_event_map || !events || (fmt != 0 && fmt != 1)) return ABEA_EINVAL;
    const int32_t n_kmers = read_len - (int32_t)kmer_size + 1;
    if (n_kmers <= 0) return ABEA_EINVAL;
    abea_index_pair_t* map = base_to_event_map;
    if (rna) {                                           /* resquiggle.c:346-357: the signal runs 3'->5' */
        for (int32_t j = 0; j < n_kmers / 2; ++j) std::swap(map[j], map[n_kmers - 1 - j]);
        for (int32_t j = 0; j < n_kmers; ++j) std::swap(map[j].start, map[j].stop);
    }
    std::string text, ss;
    text.reserve(fmt ? 256 : (size_t)n_kmers * 48);
    int64_t first_start = -1, last_end = -1, read_start = -1, read_end = -1;
    int64_t cursor = 0, run = 0, deletions = 0, count_samples = 0;
    bool before_first = true;
    int matches = 0;
    for (int32_t j = 0; j < n_kmers; ++j) {
        int64_t sig_start = -1, sig_end = -1;
        const int32_t e0 = map[j].start, e1 = map[j].stop;
        if (e0 == -1) {                                  /* k-mer without events: a deletion from the read's point of view */
            if (e1 != -1) return ABEA_EINVAL;
            if (!before_first) ++deletions;
        } else {
            if (e1 == -1) return ABEA_EINVAL;
            sig_start = (int64_t)events[e0].start;                                   /* inclusive */
            if (before_first) { first_start = sig_start; read_start = j; cursor = sig_start; before_first = false; }
            last_end = sig_end = (int64_t)events[e1].start + (int)events[e1].length;   /* exclusive */
            read_end = j;
            if (fmt == 1) {                              /* resquiggle.c:383-401: deletions, skipped samples, matched samples */
                if (deletions > 0) { appendf(ss, "%dD", (int)deletions); deletions = 0; }
                if (j == 0) cursor = sig_start;
                cursor += (run = sig_start - cursor);
                if (run) { appendf(ss, "%dI", (int)run); count_samples += run; }
                cursor += (run = sig_end - sig_start);
                if (run) { ++matches; appendf(ss, "%d,", (int)run); count_samples += run; }
            }
        }
        if (fmt == 0) {                                  /* resquiggle.c:406-427 */
            appendf(text, "%s\t%d\t", read_id, rna ? n_kmers - j - 1 : j);
            if (sig_start < 0) text += ".\t"; else appendf(text, "%ld\t", (long)sig_start);
            if (sig_end < 0) text += "."; else appendf(text, "%ld", (long)sig_end);
            text += "\n";
            if (sig_start >= 0 && sig_end >= 0 && sig_end <= sig_start) return ABEA_EINVAL;   /* the reference exits here */
        }
    }
    if (fmt == 1) {                                      /* resquiggle.c:431-447 */
        if (first_start == -1 || last_end == -1 || count_samples != last_end - first_start) return ABEA_EINVAL;
        appendf(text, "%s\t%ld\t%ld\t%ld\t+\t", read_id, (long)n_samples, (long)first_start, (long)last_end);
        appendf(text, "%s\t%d\t%ld\t%ld\t", read_id, n_kmers, (long)(rna ? n_kmers - read_start : read_start),
                (long)(rna ? n_kmers - 1 - read_end : read_end + 1));
        appendf(text, "%d\t%d\t%d\t", matches, n_kmers, 255);
        appendf(text, "sc:f:%f\t", scale);
        appendf(text, "sh:f:%f\t", shift);
        text += "ss:Z:"; text += ss; text += "\n";
    }
    if (out && cap) {
        const size_t n = text.size() < cap - 1 ? text.size() : cap - 1;
        memcpy(out, text.data(), n);
        out[n] = '\0';
    }
    return (int64_t)text.size();
}
