// Host-loop variants for abea_host.cpp: flatten = gather event_t.mean (stride 24 B) into a float array,
// expand = 2-bit walk codes -> 8-byte pairs.  Plain stores vs non-temporal stores vs software prefetch.
// g++ -O3 -march=x86-64-v3 -pthread hostbw2.cpp -o hostbw2 && ./hostbw2 [GiB of events]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>
struct ev_t { uint64_t start; float length, mean, stdv; };
struct pair_t { int32_t ref_pos, read_pos; };
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double par(int T, F f) {
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(f, t);
    for (auto& x : th) x.join();
    return now() - t0;
}
static void gather_plain(const ev_t* ev, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = ev[i].mean; }
static void gather_nt(const ev_t* ev, float* out, size_t n, int pf) {
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        if (pf) _mm_prefetch((const char*)(ev + i + pf), _MM_HINT_NTA);
        _mm_stream_ps(out + i, _mm_set_ps(ev[i + 3].mean, ev[i + 2].mean, ev[i + 1].mean, ev[i].mean));
    }
    for (; i < n; ++i) out[i] = ev[i].mean;
    _mm_sfence();
}
static void gather_pf(const ev_t* ev, float* out, size_t n, int pf) {
    for (size_t i = 0; i < n; ++i) { if ((i & 7) == 0) _mm_prefetch((const char*)(ev + i + pf), _MM_HINT_NTA); out[i] = ev[i].mean; }
}
static void expand_plain(const uint32_t* codes, int32_t n, int32_t k, int32_t e, pair_t* out) {
    pair_t* o = out + n;
    for (int32_t j = 0; j < n; j += 16) {
        uint32_t w = codes[j >> 4];
        const int32_t lim = n - j < 16 ? n - j : 16;
        for (int32_t t = 0; t < lim; ++t) { --o; o->ref_pos = k; o->read_pos = e; const uint32_t cd = w & 3u; w >>= 2; k -= (cd != 1u); e -= (cd != 2u); }
    }
}
static void expand_nt(const uint32_t* codes, int32_t n, int32_t k, int32_t e, pair_t* out) {
    // totals first (2 bits per step: 1 = event only, 2 = k-mer only), then walk the codes backwards writing ascending
    int32_t n1 = 0, n2 = 0;
    for (int32_t j = 0; j < n; j += 16) {
        uint32_t w = codes[j >> 4];
        if (n - j < 16) w &= (1u << (2 * (n - j))) - 1u;
        n1 += __builtin_popcount(w & ~(w >> 1) & 0x55555555u);
        n2 += __builtin_popcount((w >> 1) & ~w & 0x55555555u);
    }
    k -= n - n1; e -= n - n2;                  // position after the last step
    long long* o = (long long*)out;
    for (int32_t j = n - 1; j >= 0; --j) {
        const uint32_t cd = (codes[j >> 4] >> (2 * (j & 15))) & 3u;
        k += (cd != 1u); e += (cd != 2u);
        _mm_stream_si64(o++, (long long)(((uint64_t)(uint32_t)e << 32) | (uint32_t)k));
    }
    _mm_sfence();
}
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n_ev = (size_t)(gib * (1ull << 30) / sizeof(ev_t));
    ev_t* ev = (ev_t*)aligned_alloc(4096, n_ev * sizeof(ev_t));
    float* out = (float*)aligned_alloc(4096, n_ev * 4);
    const size_t n_pairs = n_ev;                                    // ~1 pair per event
    pair_t* pairs = (pair_t*)aligned_alloc(4096, n_pairs * 8);
    uint32_t* codes = (uint32_t*)aligned_alloc(4096, n_pairs / 4 + 64);
    par(16, [&](int t) { size_t lo = n_ev * t / 16, hi = n_ev * (t + 1) / 16; for (size_t i = lo; i < hi; ++i) { ev[i].mean = (float)i; out[i] = 0; pairs[i].ref_pos = 0; } });
    for (size_t i = 0; i < n_pairs / 16 + 16; ++i) codes[i] = (uint32_t)(i * 2654435761u) & 0xAAAAAAAAu ? ((uint32_t)(i * 2654435761u) & 0x55555555u) : 0x24924924u;
    const int32_t READ = 16384;                                       // pairs per read
    for (int T : {8, 12, 14, 16, 24}) {
        double a = par(T, [&](int t) { size_t lo = n_ev * t / T / 4 * 4, hi = n_ev * (t + 1) / T / 4 * 4; gather_plain(ev + lo, out + lo, hi - lo); });
        double b = par(T, [&](int t) { size_t lo = n_ev * t / T / 4 * 4, hi = n_ev * (t + 1) / T / 4 * 4; gather_nt(ev + lo, out + lo, hi - lo, 0); });
        double c = par(T, [&](int t) { size_t lo = n_ev * t / T / 4 * 4, hi = n_ev * (t + 1) / T / 4 * 4; gather_nt(ev + lo, out + lo, hi - lo, 64); });
        double d = par(T, [&](int t) { size_t lo = n_ev * t / T / 4 * 4, hi = n_ev * (t + 1) / T / 4 * 4; gather_pf(ev + lo, out + lo, hi - lo, 128); });
        const size_t n_reads = n_pairs / READ;
        double x = par(T, [&](int t) { for (size_t r = n_reads * t / T; r < n_reads * (t + 1) / T; ++r) expand_plain(codes + r * (READ / 16), READ, 1 << 30, 1 << 30, pairs + r * READ); });
        double y = par(T, [&](int t) { for (size_t r = n_reads * t / T; r < n_reads * (t + 1) / T; ++r) expand_nt(codes + r * (READ / 16), READ, 1 << 30, 1 << 30, pairs + r * READ); });
        printf("threads %2d: flatten plain %.0f | nt %.0f | nt+prefetch %.0f | prefetch %.0f Mevents/s ;  expand plain %.0f | nt ascending %.0f Mpairs/s\n",
               T, n_ev / a / 1e6, n_ev / b / 1e6, n_ev / c / 1e6, n_ev / d / 1e6, n_reads * READ / x / 1e6, n_reads * READ / y / 1e6);
    }
    return 0;
}
