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
#include <algorithm>
#include <utility>
#include <vector>
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

/* samples [begin, end) a k-mer's events cover; begin == -1: the k-mer has no events */
struct kmer_span { int64_t begin, end; };

/* base_to_event_map + event table -> one span per k-mer, in the order the text lists them (resquiggle.c:359-381: first event's
 * start, inclusive, to last event's start + length, exclusive).  false: an entry with only one of its two ends set. */
bool spans_of(const abea_index_pair_t* map, int32_t n_kmers, const abea_event_t* events, std::vector<kmer_span>& spans) {
    spans.resize((size_t)n_kmers);
    for (int32_t j = 0; j < n_kmers; ++j) {
        const abea_index_pair_t m = map[j];
        if ((m.start == -1) != (m.stop == -1)) return false;
        if (m.start == -1) { spans[(size_t)j] = kmer_span{-1, -1}; continue; }
        const abea_event_t& last = events[m.stop];
        spans[(size_t)j] = kmer_span{(int64_t)events[m.start].start, (int64_t)last.start + (int)last.length};
    }
    return true;
}

/* TSV (resquiggle.c:406-427): a line per k-mer — read id, k-mer index (counted from the 5' end: reversed for RNA), first
 * sample, one-past-last sample, "." for a k-mer without events.  false where the reference exits: an empty or inverted span. */
bool emit_tsv(std::string& text, const char* read_id, const std::vector<kmer_span>& spans, bool rna) {
    const int32_t n = (int32_t)spans.size();
    for (int32_t j = 0; j < n; ++j) {
        const kmer_span& sp = spans[(size_t)j];
        appendf(text, "%s\t%d\t", read_id, rna ? n - j - 1 : j);
        if (sp.begin < 0) { text += ".\t.\n"; continue; }
        appendf(text, "%ld\t%ld\n", (long)sp.begin, (long)sp.end);
        if (sp.end <= sp.begin) return false;
    }
    return true;
}

/* PAF (resquiggle.c:383-401, 431-447): one line; ss:Z: walks the k-mers from the first to the last that has events: "<n>D" for
 * n k-mers without events in between, "<n>I" for samples between two spans that belong to no k-mer, "<n>," for the samples of
 * a span.  k-mers without events after the last span are not reported (the reference never flushes its deletion count).
 * false when the pieces do not add up to the signal stretch they describe (the reference asserts). */
bool emit_paf(std::string& text, const char* read_id, const std::vector<kmer_span>& spans, int64_t n_samples, float scale,
              float shift, bool rna) {
    const int32_t n = (int32_t)spans.size();
    int32_t first = 0, last = n - 1;
    while (first < n && spans[(size_t)first].begin < 0) ++first;
    while (last >= 0 && spans[(size_t)last].begin < 0) --last;
    if (first > last) return false;
    std::string ss;
    int64_t at = spans[(size_t)first].begin, covered = 0;
    int pending_gaps = 0, n_match = 0;
    for (int32_t j = first; j <= last; ++j) {
        const kmer_span& sp = spans[(size_t)j];
        if (sp.begin < 0) { ++pending_gaps; continue; }
        if (pending_gaps) { appendf(ss, "%dD", pending_gaps); pending_gaps = 0; }
        const int64_t skipped = sp.begin - at, width = sp.end - sp.begin;
        if (skipped) { appendf(ss, "%dI", (int)skipped); covered += skipped; }
        if (width) { appendf(ss, "%d,", (int)width); covered += width; ++n_match; }
        at = sp.end;
    }
    const int64_t sig_from = spans[(size_t)first].begin, sig_to = spans[(size_t)last].end;
    if (covered != sig_to - sig_from) return false;
    appendf(text, "%s\t%ld\t%ld\t%ld\t+\t", read_id, (long)n_samples, (long)sig_from, (long)sig_to);
    appendf(text, "%s\t%d\t%ld\t%ld\t", read_id, n, (long)(rna ? n - first : first), (long)(rna ? n - 1 - last : last + 1));
    appendf(text, "%d\t%d\t%d\tsc:f:%f\tsh:f:%f\tss:Z:", n_match, n, 255, scale, shift);
    text += ss; text += "\n";
    return true;
}
}  // namespace

extern "C" int64_t abea_rsq_format(char* out, size_t cap, int fmt, const char* read_id, int32_t read_len,
                                   uint32_t kmer_size, abea_index_pair_t* base_to_event_map,
                                   const abea_event_t* events, int64_t n_samples, float scale, float shift, int rna) {
    if (!read_id || !base_to_event_map || !events || (fmt != 0 && fmt != 1)) return ABEA_EINVAL;
    const int32_t n_kmers = read_len - (int32_t)kmer_size + 1;
    if (n_kmers <= 0) return ABEA_EINVAL;
    if (rna) {                                           /* resquiggle.c:346-357: the signal runs 3'->5'; in place, as documented */
        std::reverse(base_to_event_map, base_to_event_map + n_kmers);
        for (int32_t j = 0; j < n_kmers; ++j) std::swap(base_to_event_map[j].start, base_to_event_map[j].stop);
    }
    std::vector<kmer_span> spans;
    if (!spans_of(base_to_event_map, n_kmers, events, spans)) return ABEA_EINVAL;
    std::string text;
    text.reserve(fmt ? 256 + (size_t)n_kmers * 4 : (size_t)n_kmers * 48);
    const bool ok = fmt == 0 ? emit_tsv(text, read_id, spans, rna != 0) : emit_paf(text, read_id, spans, n_samples, scale, shift, rna != 0);
    if (!ok) return ABEA_EINVAL;
    if (out && cap) {
        const size_t n = text.size() < cap - 1 ? text.size() : cap - 1;
        memcpy(out, text.data(), n);
        out[n] = '\0';
    }
    return (int64_t)text.size();
}
