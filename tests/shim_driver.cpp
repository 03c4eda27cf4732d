/* tests/shim_driver.cpp — a plain C++ caller shaped like f5c's process_db slice around align_db():
 * reads a flattened batch dumped by the test, builds db_t-like per-read arrays (malloc'ed output
 * buffers of n_events+read_len pairs, f5c.c:724), calls the shim, prints the pair lists in the
 * reference's --print-banded-aln format (f5c.c:989-1006: "<read index>\t{ref,read}\t..." lines). */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
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

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s batch.bin out.txt\n", argv[0]); return 2; }
    std::vector<char> buf = slurp(argv[1]);
    const char* p = buf.data();
    int32_t hdr[4]; memcpy(hdr, p, 16); p += 16;            /* n_reads, kmer_size, n_model, pad */
    const int32_t n = hdr[0]; const uint32_t k = (uint32_t)hdr[1]; const int32_t n_model = hdr[2];
    const abea_model_t* model = (const abea_model_t*)p; p += (size_t)n_model * sizeof(abea_model_t);
    std::vector<int32_t> read_len(n), n_events(n);
    memcpy(read_len.data(), p, 4 * (size_t)n); p += 4 * (size_t)n;
    memcpy(n_events.data(), p, 4 * (size_t)n); p += 4 * (size_t)n;
    std::vector<abea_scalings_t> sc(n);
    memcpy(sc.data(), p, sizeof(abea_scalings_t) * (size_t)n); p += sizeof(abea_scalings_t) * (size_t)n;

    abea_f5c_db db; memset(&db, 0, sizeof db);
    db.n_bam_rec = n;
    std::vector<char*> read(n); std::vector<abea_f5c_event_table> et(n);
    std::vector<abea_pair_t*> pairs(n); std::vector<int32_t> n_pairs(n, -1);
    for (int32_t i = 0; i < n; ++i) {
        read[i] = (char*)malloc((size_t)read_len[i] + 1);
        memcpy(read[i], p, (size_t)read_len[i]); read[i][read_len[i]] = 0; p += read_len[i];
        db.sum_bases += read_len[i];
    }
    for (int32_t i = 0; i < n; ++i) {
        et[i].n = (size_t)n_events[i]; et[i].start = 0; et[i].end = et[i].n;
        et[i].event = (abea_event_t*)malloc(sizeof(abea_event_t) * (et[i].n + 1));
        memcpy(et[i].event, p, sizeof(abea_event_t) * et[i].n); p += sizeof(abea_event_t) * et[i].n;
        pairs[i] = (abea_pair_t*)malloc(sizeof(abea_pair_t) * (et[i].n + (size_t)read_len[i]));   /* f5c.c:724 */
    }
    db.read = read.data(); db.read_len = read_len.data(); db.nsample = nullptr; db.et = et.data();
    db.scalings = sc.data(); db.event_align_pairs = pairs.data(); db.n_event_align_pairs = n_pairs.data();
    /* test knobs: SHIM_NSAMPLE0 = reads whose db->sig[i]->nsample is 0 (bad reads, f5c.c:826-828); SHIM_DEVS = device
     * list for one multi-GPU context; SHIM_FUSED = align_db + scaling_db in one call */
    std::vector<int64_t> nsample(n, 4000);
    if (const char* e = getenv("SHIM_NSAMPLE0")) {
        for (const char* q = e; *q;) { nsample[(size_t)strtol(q, (char**)&q, 10)] = 0; if (*q == ',') ++q; }
        db.nsample = nsample.data();
    }
    std::vector<int32_t> devs;
    if (const char* e = getenv("SHIM_DEVS"))
        for (const char* q = e; *q;) { devs.push_back((int32_t)strtol(q, (char**)&q, 10)); if (*q == ',') ++q; }
    const bool fused = getenv("SHIM_FUSED") != nullptr;
    std::vector<abea_index_pair_t*> b2e(n, nullptr); std::vector<double> epb(n, 0.0);
    std::vector<int32_t> flag(n, 0), nal(n, 0);
    std::vector<abea_scalings_t> sc0 = sc;
    db.base_to_event_map = b2e.data(); db.events_per_base = epb.data(); db.read_stat_flag = flag.data();
    db.n_event_alignment = nal.data();

    abea_f5c_core core; memset(&core, 0, sizeof core);
    core.model = model; core.kmer_size = k; core.cuda_dev_id = 0; core.cuda_mem_frac = 0.2f; core.verbosity = 2;
    core.cuda_dev_ids = devs.data(); core.n_cuda_devs = (int32_t)devs.size();
    abea_f5c_init(&core);
    if (fused) {
        abea_f5c_align_scale(&core, &db);
        for (int32_t i = 0; i < n; ++i) { free(b2e[i]); b2e[i] = nullptr; flag[i] = 0; }
        sc = sc0; db.scalings = sc.data();
        abea_f5c_align_scale(&core, &db);                    /* a second batch through the same context */
    } else if (getenv("SHIM_ASYNC")) {
        /* two process_db batches in flight on one context: the first and the second half of the reads as two db views */
        const int32_t h = n / 2;
        abea_f5c_db a = db, b = db;
        a.n_bam_rec = h; b.n_bam_rec = n - h;
        b.read += h; b.read_len += h; b.et += h; b.scalings += h; b.event_align_pairs += h; b.n_event_align_pairs += h;
        if (b.nsample) b.nsample += h;
        for (int rep = 0; rep < 2; ++rep) {                  /* twice: lanes are reused */
            void* ha = abea_f5c_align_submit(&core, &a);
            void* hb = abea_f5c_align_submit(&core, &b);
            abea_f5c_align_wait(&core, hb);
            abea_f5c_align_wait(&core, ha);
        }
    } else {
        abea_f5c_align(&core, &db);
        const int reps = getenv("SHIM_REPS") ? atoi(getenv("SHIM_REPS")) : 1;   /* more batches through the same context */
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            abea_f5c_align(&core, &db);
            if (getenv("SHIM_REPS"))
                fprintf(stderr, "abea_f5c_align wall %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    }
    FILE* out = fopen(argv[2], "w");
    for (int32_t i = 0; i < n && !getenv("SHIM_NOPRINT"); ++i) {      /* SHIM_NOPRINT: timing runs on large batches */
        fprintf(out, "%d\t", i);
        for (int32_t j = 0; j < n_pairs[i]; ++j) fprintf(out, "{%d,%d}\t", pairs[i][j].ref_pos, pairs[i][j].read_pos);
        fprintf(out, "\n");
    }
    fclose(out);
    if (fused) {                                             /* "<i> flag n_alignment events_per_base shift scale var | start,stop ..." */
        char path[4096]; snprintf(path, sizeof path, "%s.scale", argv[2]);
        FILE* f = fopen(path, "w");
        for (int32_t i = 0; i < n; ++i) {
            fprintf(f, "%d\t%d\t%d\t%a\t%a\t%a\t%a\t", i, flag[i], nal[i], epb[i], (double)sc[i].shift, (double)sc[i].scale, (double)sc[i].var);
            if (b2e[i]) for (int32_t j = 0; j < read_len[i] - (int32_t)k + 1; ++j) fprintf(f, "%d,%d ", b2e[i][j].start, b2e[i][j].stop);
            else fprintf(f, "NULL");
            fprintf(f, "\n");
        }
        fclose(f);
    }
    fprintf(stderr, "kernel time %.3f ms, memcpy %.3f ms\n", core.align_kernel_time * 1e3 / 2, core.align_cuda_memcpy * 1e3 / 2);
    abea_f5c_free(&core);
    for (int32_t i = 0; i < n; ++i) { free(read[i]); free(et[i].event); free(pairs[i]); free(b2e[i]); }
    return 0;
}
