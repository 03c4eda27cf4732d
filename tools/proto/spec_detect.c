/* Prototype (CPU) of the segment-parallel peak detector planned for abea_ev_detect: validates, on real and synthetic
 * signals, that runs of the two-detector automaton (events.c:380-452) started from the reset state at arbitrary
 * segment starts re-synchronise with the true trajectory, and measures how fast.  Not part of the product or tests'
 * import graph; build: gcc -O2 -o /tmp/spec_detect tools/proto/spec_detect.c -lm ; input: a file of
 * [int64 n][float pA * n] records. */
#include "../../oracle/abea_oracle.c"

typedef struct { float pv[2]; int pp[2]; int valid[2]; long long masked_to; } st_t;

static void st_reset(st_t* s) { s->pv[0] = s->pv[1] = FLT_MAX; s->pp[0] = s->pp[1] = -1; s->valid[0] = s->valid[1] = 0; s->masked_to = -1; }

/* one position; returns number of peaks fired (0..2), positions in out[] */
static int st_step(st_t* s, long long p, float c0, float c1, int* out) {
    static const float thr[2] = {1.4f, 9.0f};
    static const int win[2] = {3, 6};
    const float h = 0.2f;
    int nf = 0;
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && s->masked_to >= p) continue;
        const float cur = k ? c1 : c0;
        if (s->pp[k] == -1) {
            if (cur < s->pv[k]) s->pv[k] = cur;
            else if (cur - s->pv[k] > h) { s->pv[k] = cur; s->pp[k] = (int)p; }
        } else {
            if (cur > s->pv[k]) { s->pv[k] = cur; s->pp[k] = (int)p; }
            if (k == 0 && s->pv[0] > thr[0]) { s->masked_to = s->pp[0] + win[0]; s->pp[1] = -1; s->pv[1] = FLT_MAX; s->valid[1] = 0; }
            if (s->pv[k] - cur > h && s->pv[k] > thr[k]) s->valid[k] = 1;
            if (s->valid[k] && (p - s->pp[k]) > win[k] / 2) { out[nf++] = s->pp[k]; s->pp[k] = -1; s->pv[k] = cur; s->valid[k] = 0; }
        }
    }
    return nf;
}
static int st_equal(const st_t* a, const st_t* b, long long p) {   /* equal as far as any position > p can tell */
    for (int k = 0; k < 2; ++k)
        if (a->pp[k] != b->pp[k] || a->valid[k] != b->valid[k] || memcmp(&a->pv[k], &b->pv[k], 4)) return 0;
    const long long ma = a->masked_to > p ? a->masked_to : p, mb = b->masked_to > p ? b->masked_to : p;
    return ma == mb;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    const int G = argc > 2 ? atoi(argv[2]) : 512, F = argc > 3 ? atoi(argv[3]) : 128;
    long long n, reads = 0, fails = 0, segs = 0, maxsync = 0, sumsync = 0, mism = 0;
    long long hist[8] = {0};
    while (fread(&n, 8, 1, f) == 1) {
        float* pa = malloc(4 * n);
        if (fread(pa, 4, n, f) != (size_t)n) return 2;
        double* S = calloc(n + 1, 8); double* Q = calloc(n + 1, 8);
        for (long long i = 0; i < n; ++i) { S[i + 1] = S[i] + pa[i]; Q[i + 1] = Q[i] + pa[i] * pa[i]; }
        float* t1 = ev_tstat(S, Q, n, 3); float* t2 = ev_tstat(S, Q, n, 6);
        /* truth: sequential; the reference starts with masked_to = 0, i.e. position 0 is skipped by both detectors */
        int* truth = malloc(8 * n + 16); long long nt = 0;
        { st_t s; st_reset(&s); int o[2];
          for (long long p = 1; p < n; ++p) { int nf = st_step(&s, p, t1[p], t2[p], o); for (int q = 0; q < nf; ++q) truth[nt++] = o[q]; } }
        /* speculative: every segment from the reset state */
        const long long nseg = (n + G - 1) / G;
        st_t* endst = malloc(sizeof(st_t) * nseg);
        int** sp = malloc(sizeof(int*) * nseg); int* nsp = calloc(nseg, sizeof(int));
        for (long long j = 0; j < nseg; ++j) {
            st_t s; st_reset(&s); int o[2];
            sp[j] = malloc(8 * G + 16);
            const long long lo = j * G < 1 ? 1 : j * G, hi = (j + 1) * G < n ? (j + 1) * G : n;
            for (long long p = lo; p < hi; ++p) { int nf = st_step(&s, p, t1[p], t2[p], o); for (int q = 0; q < nf; ++q) sp[j][nsp[j]++] = o[q]; }
            endst[j] = s;
        }
        /* fix-up: true run from the previous segment's speculative end state, in lockstep with a replay of the speculative run */
        int* got = malloc(8 * n + 16); long long ng = 0; int failed = 0;
        for (long long j = 0; j < nseg; ++j) {
            int skip = 0;
            if (j > 0) {
                st_t tr = endst[j - 1], spc; st_reset(&spc); int o[2], o2[2];
                const long long lo = j * G, hi = (j + 1) * G < n ? (j + 1) * G : n;
                long long p = lo; int synced = st_equal(&tr, &spc, lo - 1);
                for (; p < hi && !synced && p < lo + F; ++p) {
                    int nf = st_step(&tr, p, t1[p], t2[p], o); for (int q = 0; q < nf; ++q) got[ng++] = o[q];
                    skip += st_step(&spc, p, t1[p], t2[p], o2);
                    synced = st_equal(&tr, &spc, p);
                }
                if (!synced) failed = 1;      /* would go to the sequential fallback */
                const long long d = p - lo; sumsync += d; if (d > maxsync) maxsync = d;
                int b = 0; while ((1 << (b + 2)) < d && b < 7) ++b; hist[b]++;
            }
            for (int q = skip; q < nsp[j]; ++q) got[ng++] = sp[j][q];
            ++segs;
        }
        if (failed) ++fails;
        else if (ng != nt || memcmp(got, truth, 4 * nt)) { ++mism; fprintf(stderr, "MISMATCH read %lld n=%lld peaks %lld vs %lld\n", reads, n, ng, nt); }
        ++reads;
        for (long long j = 0; j < nseg; ++j) free(sp[j]);
        free(sp); free(nsp); free(endst); free(got); free(truth); free(t1); free(t2); free(S); free(Q); free(pa);
    }
    printf("G=%d F=%d reads=%lld segments=%lld fallback_reads=%lld mismatches=%lld mean_sync=%.2f max_sync=%lld\n", G, F, reads, segs, fails, mism,
           segs ? (double)sumsync / segs : 0.0, maxsync);
    printf("sync distance histogram (<=4,8,16,32,64,128,256,more):"); for (int b = 0; b < 8; ++b) printf(" %lld", hist[b]); printf("\n");
    return mism != 0;
}
