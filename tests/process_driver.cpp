/* tests/process_driver.cpp — a plain g++ caller shaped like f5c's process_db_rsq (src/resquiggle.c:283-315) followed by
 * output_db_rsq (resquiggle.c:319-449) and process_db's --print-scaling block (src/f5c.c:1008-1020): no Python anywhere in
 * the chain.  Reads a batch dumped by the test, fills the shim's db view as INTEGRATION.md's glue does, and calls
 *   mode 0: abea_f5c_process      raw float ADC signals -> events -> alignment -> scaling_single
 *   mode 1: abea_f5c_align_scale  given event tables (means) + estimated scalings -> alignment -> scaling_single
 *   mode 2: abea_f5c_event_db, then abea_f5c_align_scale — the two calls process_db makes, one after the other
 * and writes what f5c would hold / print afterwards:
 *   <out>.events   binary: per read u64 n, then n event_t
 *   <out>.state    text : i flag n_event_alignment events_per_base shift scale var n_pairs (floats as %a)
 *   <out>.pairs    text : the --print-banded-aln lines (f5c.c:989-1006)
 *   <out>.b2e      text : per read "start,stop ..." or NULL
 *   <out>.scaling  text : the --print-scaling block
 *   <out>.tsv / <out>.paf : abea_rsq_format_batch, fmt 0 / 1
 *   <out>.pa       binary: the signals as the call left them (pA, f5c.c:693-696), modes 0 and 2 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/abea_f5c_shim.h"

static std::vector<char> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> b((size_t)n);
    if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { perror("fread"); exit(2); }
    fclose(f);
    return b;
}
static FILE* open_out(const std::string& base, const char* ext, const char* mode) {
    FILE* f = fopen((base + ext).c_str(), mode);
    if (!f) { perror(ext); exit(2); }
    return f;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s batch.bin out_prefix\n", argv[0]); return 2; }
    std::vector<char> buf = slurp(argv[1]);
    const std::string out = argv[2];
    const char* p = buf.data();
    int32_t hdr[6]; memcpy(hdr, p, 24); p += 24;            /* n_reads, kmer_size, n_model, mode, rna, pad */
    const int32_t n = hdr[0]; const uint32_t k = (uint32_t)hdr[1]; const int32_t n_model = hdr[2], mode = hdr[3], rna = hdr[4];
    const abea_model_t* model = (const abea_model_t*)p; p += (size_t)n_model * sizeof(abea_model_t);
    std::vector<int32_t> read_len(n); memcpy(read_len.data(), p, 4 * (size_t)n); p += 4 * (size_t)n;
    std::vector<int64_t> nsample(n); memcpy(nsample.data(), p, 8 * (size_t)n); p += 8 * (size_t)n;
    std::vector<float> offset(n), range(n), digit(n);
    for (int32_t i = 0; i < n; ++i) { float t[3]; memcpy(t, p, 12); p += 12; offset[i] = t[0]; range[i] = t[1]; digit[i] = t[2]; }
    std::vector<std::string> ids(n);
    for (int32_t i = 0; i < n; ++i) { ids[i] = std::string(p, strnlen(p, 40)); p += 40; }
    std::vector<char*> read(n);
    for (int32_t i = 0; i < n; ++i) {
        read[i] = (char*)malloc((size_t)read_len[i] + 1);
        memcpy(read[i], p, (size_t)read_len[i]); read[i][read_len[i]] = 0; p += read_len[i];
    }
    std::vector<float*> raw(n, nullptr);
    std::vector<abea_f5c_event_table> et(n);
    std::vector<abea_scalings_t> sc(n);
    memset(et.data(), 0, sizeof(abea_f5c_event_table) * (size_t)n);
    memset(sc.data(), 0, sizeof(abea_scalings_t) * (size_t)n);
    std::vector<abea_pair_t*> pairs(n, nullptr);
    if (mode == 0 || mode == 2) {
        for (int32_t i = 0; i < n; ++i) {
            if (nsample[i] <= 0) continue;
            raw[i] = (float*)malloc(sizeof(float) * (size_t)nsample[i]);
            memcpy(raw[i], p, 4 * (size_t)nsample[i]); p += 4 * (size_t)nsample[i];
        }
    } else {
        std::vector<int32_t> ne(n); memcpy(ne.data(), p, 4 * (size_t)n); p += 4 * (size_t)n;
        memcpy(sc.data(), p, sizeof(abea_scalings_t) * (size_t)n); p += sizeof(abea_scalings_t) * (size_t)n;
        for (int32_t i = 0; i < n; ++i) {
            et[i].n = (size_t)ne[i]; et[i].start = 0; et[i].end = et[i].n;
            et[i].event = (abea_event_t*)calloc(et[i].n + 1, sizeof(abea_event_t));
            for (size_t j = 0; j < et[i].n; ++j) { float m; memcpy(&m, p, 4); p += 4; et[i].event[j].mean = m; }
            pairs[i] = (abea_pair_t*)malloc(sizeof(abea_pair_t) * (et[i].n + (size_t)read_len[i]));
        }
    }
    std::vector<int32_t> n_pairs(n, -1), flag(n, 0), nal(n, 0);
    std::vector<abea_index_pair_t*> b2e(n, nullptr);
    std::vector<double> epb(n, 0.0);

    abea_f5c_db db; memset(&db, 0, sizeof db);               /* the glue of INTEGRATION.md, on stand-in arrays */
    db.n_bam_rec = n; db.read = read.data(); db.read_len = read_len.data(); db.nsample = nsample.data(); db.et = et.data();
    db.scalings = sc.data(); db.event_align_pairs = pairs.data(); db.n_event_align_pairs = n_pairs.data();
    db.base_to_event_map = b2e.data(); db.events_per_base = epb.data(); db.read_stat_flag = flag.data();
    db.n_event_alignment = nal.data();
    db.rawptr = raw.data(); db.offset = offset.data(); db.range = range.data(); db.digitisation = digit.data();
    for (int32_t i = 0; i < n; ++i) db.sum_bases += read_len[i];

    abea_f5c_core core; memset(&core, 0, sizeof core);
    core.model = model; core.kmer_size = k; core.cuda_dev_id = 0; core.cuda_mem_frac = 0.2f; core.verbosity = 1; core.rna = rna;
    abea_f5c_init(&core);
    if (mode == 0) abea_f5c_process(&core, &db);
    else if (mode == 1) abea_f5c_align_scale(&core, &db);
    else { abea_f5c_event_db(&core, &db); abea_f5c_align_scale(&core, &db); }
    fprintf(stderr, "event stage %.2f ms, alignment kernels %.2f ms\n", core.event_time * 1e3, core.align_kernel_time * 1e3);

    FILE* f = open_out(out, ".events", "wb");
    for (int32_t i = 0; i < n; ++i) {
        const uint64_t ne = et[i].n;
        fwrite(&ne, 8, 1, f);
        if (ne) fwrite(et[i].event, sizeof(abea_event_t), (size_t)ne, f);
    }
    fclose(f);
    f = open_out(out, ".state", "w");
    for (int32_t i = 0; i < n; ++i)
        fprintf(f, "%d\t%d\t%d\t%a\t%a\t%a\t%a\t%d\t%a\n", i, flag[i], nal[i], epb[i], (double)sc[i].shift, (double)sc[i].scale,
                (double)sc[i].var, n_pairs[i], (double)sc[i].log_var);
    fclose(f);
    f = open_out(out, ".pairs", "w");                        /* f5c.c:989-1006 */
    for (int32_t i = 0; i < n; ++i) {
        if (flag[i] & ABEA_FAILED_ALIGNMENT) continue;
        fprintf(f, ">%s\tN_ALGN_PAIR:%d\t{ref_pos,read_pos}\n", ids[i].c_str(), (int)n_pairs[i]);
        for (int32_t j = 0; j < n_pairs[i]; ++j) fprintf(f, "{%d,%d}\t", pairs[i][j].ref_pos, pairs[i][j].read_pos);
        fprintf(f, "\n");
    }
    fclose(f);
    f = open_out(out, ".b2e", "w");
    for (int32_t i = 0; i < n; ++i) {
        if (b2e[i]) for (int32_t j = 0; j < read_len[i] - (int32_t)k + 1; ++j) fprintf(f, "%d,%d ", b2e[i][j].start, b2e[i][j].stop);
        else fprintf(f, "NULL");
        fprintf(f, "\n");
    }
    fclose(f);
    f = open_out(out, ".scaling", "w");                      /* f5c.c:1008-1020 */
    fprintf(f, "read\tshift\tscale\tvar\n");
    for (int32_t i = 0; i < n; ++i) {
        if (flag[i] & (ABEA_FAILED_ALIGNMENT | ABEA_FAILED_CALIBRATION)) continue;
        fprintf(f, "%s\t%.2lf\t%.2lf\t%.2lf\n", ids[i].c_str(), sc[i].shift, sc[i].scale, sc[i].var);
    }
    fclose(f);
    if (mode == 0 || mode == 2) {
        f = open_out(out, ".pa", "wb");
        for (int32_t i = 0; i < n; ++i) if (nsample[i] > 0) fwrite(raw[i], 4, (size_t)nsample[i], f);
        fclose(f);
    }
    /* output_db_rsq, both formats (size query first, then the text) */
    std::vector<const char*> idp(n); std::vector<const abea_event_t*> evp(n);
    for (int32_t i = 0; i < n; ++i) { idp[i] = ids[i].c_str(); evp[i] = et[i].event; }
    for (int fmt = 0; fmt < 2 && mode != 1; ++fmt) {        /* mode 1's tables hold means only: no sample coordinates to print */
        int32_t printed = 0;
        const int64_t len = abea_rsq_format_batch(nullptr, 0, fmt, n, idp.data(), read_len.data(), k, b2e.data(), evp.data(),
                                                  nsample.data(), sc.data(), flag.data(), rna, &printed);
        if (len < 0) { fprintf(stderr, "abea_rsq_format_batch failed: %lld\n", (long long)len); return 3; }
        std::vector<char> text((size_t)len + 1);
        abea_rsq_format_batch(text.data(), text.size(), fmt, n, idp.data(), read_len.data(), k, b2e.data(), evp.data(), nsample.data(),
                              sc.data(), flag.data(), rna, &printed);
        f = open_out(out, fmt ? ".paf" : ".tsv", "w");
        fwrite(text.data(), 1, (size_t)len, f);
        fclose(f);
        fprintf(stderr, "rsq fmt %d: %d reads printed, %lld bytes\n", fmt, printed, (long long)len);
    }
    abea_f5c_free(&core);
    for (int32_t i = 0; i < n; ++i) { free(read[i]); free(raw[i]); free(et[i].event); free(pairs[i]); free(b2e[i]); }
    return 0;
}
