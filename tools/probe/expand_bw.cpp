// Variants of the host entry's pair expansion (2-bit walk codes -> 8-byte pairs, written back to front):
// plain stores vs non-temporal 8-byte / 32-byte stores.  g++ -O3 -march=x86-64-v3 -pthread expand_bw.cpp
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>
struct pair_t { int32_t ref_pos, read_pos; };
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double par(int T, F f) {
    double t0 = now(); std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(f, t);
    for (auto& x : th) x.join();
    return now() - t0;
}
static void expand_plain(const uint32_t* codes, int32_t n, int32_t k, int32_t e, pair_t* out) {
    pair_t* o = out + n;
    for (int32_t j = 0; j < n; j += 16) {
        uint32_t w = codes[j >> 4];
        const int32_t lim = n - j < 16 ? n - j : 16;
        for (int32_t t = 0; t < lim; ++t) { --o; o->ref_pos = k; o->read_pos = e; const uint32_t cd = w & 3u; w >>= 2; k -= (cd != 1u); e -= (cd != 2u); }
    }
}
static void expand_nt64(const uint32_t* codes, int32_t n, int32_t k, int32_t e, pair_t* out) {
    long long* o = (long long*)(out + n);
    for (int32_t j = 0; j < n; j += 16) {
        uint32_t w = codes[j >> 4];
        const int32_t lim = n - j < 16 ? n - j : 16;
        for (int32_t t = 0; t < lim; ++t) { --o; _mm_stream_si64(o, (long long)(((uint64_t)(uint32_t)e << 32) | (uint32_t)k)); const uint32_t cd = w & 3u; w >>= 2; k -= (cd != 1u); e -= (cd != 2u); }
    }
    _mm_sfence();
}
static void expand_nt256(const uint32_t* codes, int32_t n, int32_t k, int32_t e, pair_t* out) {
    // back to front; the 32-byte blocks of `out` are written whole with one streaming store each
    uint64_t* base = (uint64_t*)out;
    int64_t i = n - 1;                       // index of the pair being produced
    int32_t j = 0; uint32_t w = 0;
    auto next = [&]() -> uint64_t {
        if ((j & 15) == 0) w = codes[j >> 4];
        const uint64_t v = ((uint64_t)(uint32_t)e << 32) | (uint32_t)k;
        const uint32_t cd = w & 3u; w >>= 2; ++j; k -= (cd != 1u); e -= (cd != 2u);
        return v;
    };
    while (i >= 0 && (((uintptr_t)(base + i + 1)) & 31)) { base[i] = next(); --i; }      // unaligned tail
    while (i >= 3) {
        const uint64_t a = next(), b = next(), c = next(), d = next();                    // pairs i, i-1, i-2, i-3
        _mm256_stream_si256((__m256i*)(base + i - 3), _mm256_set_epi64x((long long)a, (long long)b, (long long)c, (long long)d));
        i -= 4;
    }
    while (i >= 0) { base[i] = next(); --i; }
    _mm_sfence();
}
int main(int argc, char** argv) {
    const size_t n_pairs = (size_t)((argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30) / 8);
    pair_t* pairs = (pair_t*)aligned_alloc(4096, n_pairs * 8 + 4096);
    uint32_t* codes = (uint32_t*)aligned_alloc(4096, n_pairs / 4 + 4096);
    const int32_t READ = 25001;                                        // pairs per read (odd: unaligned ends)
    par(16, [&](int t) { for (size_t i = n_pairs * t / 16; i < n_pairs * (t + 1) / 16; ++i) pairs[i].ref_pos = 0; });
    for (size_t i = 0; i < n_pairs / 16 + 64; ++i) { uint32_t x = (uint32_t)(i * 2654435761u); uint32_t wv = 0; for (int q = 0; q < 16; ++q) { wv |= ((x >> (2 * q)) % 3u) << (2 * q); } codes[i] = wv; }
    const size_t n_reads = n_pairs / READ;
    // correctness of the variants against plain
    { std::vector<pair_t> a(READ), b(READ), c(READ + 4);
      expand_plain(codes, READ, 1 << 30, 1 << 30, a.data()); expand_nt64(codes, READ, 1 << 30, 1 << 30, b.data()); expand_nt256(codes, READ, 1 << 30, 1 << 30, c.data() + 1);
      printf("nt64 %s, nt256 %s\n", memcmp(a.data(), b.data(), READ * 8) ? "DIFFERS" : "ok", memcmp(a.data(), c.data() + 1, READ * 8) ? "DIFFERS" : "ok"); }
    for (int T : {8, 12, 14, 16}) {
        double x = par(T, [&](int t) { for (size_t r = n_reads * t / T; r < n_reads * (t + 1) / T; ++r) expand_plain(codes + r * (READ / 16 + 1), READ, 1 << 30, 1 << 30, pairs + r * READ); });
        double y = par(T, [&](int t) { for (size_t r = n_reads * t / T; r < n_reads * (t + 1) / T; ++r) expand_nt64(codes + r * (READ / 16 + 1), READ, 1 << 30, 1 << 30, pairs + r * READ); });
        double z = par(T, [&](int t) { for (size_t r = n_reads * t / T; r < n_reads * (t + 1) / T; ++r) expand_nt256(codes + r * (READ / 16 + 1), READ, 1 << 30, 1 << 30, pairs + r * READ); });
        printf("threads %2d: expand plain %.0f | nt 8-byte %.0f | nt 32-byte %.0f Mpairs/s\n", T, n_reads * READ / x / 1e6, n_reads * READ / y / 1e6, n_reads * READ / z / 1e6);
    }
    return 0;
}
