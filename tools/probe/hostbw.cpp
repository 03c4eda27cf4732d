// Host memory-bandwidth probe for the host-buffer entry's flatten / un-flatten loops:
//   gather : read float at stride 24 B (event_t.mean) -> contiguous float   (what flatten does)
//   copy   : memcpy-like copy of 8-B pairs, temporal vs non-temporal stores  (what un-flatten does)
// g++ -O3 -march=native -pthread hostbw.cpp -o hostbw && ./hostbw [GiB]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>
struct ev_t { uint64_t start; float length, mean, stdv; };
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double par(int T, F f) {
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(f, t);
    for (auto& x : th) x.join();
    return now() - t0;
}
static void nt_copy(void* dst, const void* src, size_t n) {   // dst 32-B aligned, n multiple of 32
    const __m256i* s = (const __m256i*)src; __m256i* d = (__m256i*)dst;
    for (size_t i = 0; i < n / 32; ++i) _mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
    _mm_sfence();
}
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n_ev = (size_t)(gib * (1ull << 30) / sizeof(ev_t));
    ev_t* ev = (ev_t*)aligned_alloc(4096, n_ev * sizeof(ev_t));
    float* out = (float*)aligned_alloc(4096, n_ev * 4);
    const size_t n_b = n_ev * 8;
    char* a = (char*)aligned_alloc(4096, n_b); char* b = (char*)aligned_alloc(4096, n_b);
    par(16, [&](int t) { size_t lo = n_ev * t / 16, hi = n_ev * (t + 1) / 16; for (size_t i = lo; i < hi; ++i) { ev[i].mean = (float)i; out[i] = 0; } memset(a + n_b * t / 16, 1, n_b / 16); memset(b + n_b * t / 16, 2, n_b / 16); });
    for (int T : {1, 4, 8, 16, 32, 64}) {
        double tg = par(T, [&](int t) { size_t lo = n_ev * t / T, hi = n_ev * (t + 1) / T; for (size_t i = lo; i < hi; ++i) out[i] = ev[i].mean; });
        double tc = par(T, [&](int t) { size_t lo = n_b / T * t / 4096 * 4096, hi = (t == T - 1) ? n_b : n_b / T * (t + 1) / 4096 * 4096; memcpy(b + lo, a + lo, hi - lo); });
        double tn = par(T, [&](int t) { size_t lo = n_b / T * t / 4096 * 4096, hi = (t == T - 1) ? n_b / 4096 * 4096 : n_b / T * (t + 1) / 4096 * 4096; nt_copy(b + lo, a + lo, hi - lo); });
        // per-read sized memcpy (128 KiB pieces), like un-flatten
        double tp = par(T, [&](int t) { size_t lo = n_b / T * t / 4096 * 4096, hi = (t == T - 1) ? n_b / 4096 * 4096 : n_b / T * (t + 1) / 4096 * 4096; for (size_t o = lo; o < hi; o += 131072) memcpy(b + o, a + o, std::min<size_t>(131072, hi - o)); });
        printf("threads %2d: gather 24B-stride %.1f Mevents/s (%.1f GB/s read) | memcpy %.1f GB/s | nt-copy %.1f GB/s | 128KiB memcpy %.1f GB/s (payload, one direction)\n",
               T, n_ev / tg / 1e6, n_ev * 24 / tg / 1e9, n_b / tc / 1e9, n_b / tn / 1e9, n_b / tp / 1e9);
    }
    return 0;
}
