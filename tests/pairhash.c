/* tests/pairhash.c — the hash of tests/pairhash.py in C (test infrastructure): the full-size tests hash 2.5 G pairs on the
 * GPU box, where the numpy formulation (a gather + eight array passes per 16 M pairs) costs minutes of box time.  Compiled on
 * demand by tests/pairhash.py (gcc -O2 -shared); the numpy version stays the definition and the two are compared in
 * tests/test_config_goldens.py. */
#include <stdint.h>

void pairhash_range(const int32_t* pairs, const int64_t* pair_ptr, const int32_t* n_pairs, int64_t lo, int64_t hi, uint64_t* out) {
    const uint64_t C1 = 0x9E3779B97F4A7C15ull, C2 = 0xBF58476D1CE4E5B9ull;
    for (int64_t i = lo; i < hi; ++i) {
        const int32_t* p = pairs + 2 * pair_ptr[i];
        const int64_t n = n_pairs[i] > 0 ? n_pairs[i] : 0;
        uint64_t h = 0;
        for (int64_t j = 0; j < n; ++j) {
            const uint64_t x = ((uint64_t)(uint32_t)p[2 * j] << 32) | (uint32_t)p[2 * j + 1];
            uint64_t y = x * C1;
            y = (y ^ (y >> 29)) * C2;
            h += y * (2 * (uint64_t)j + 1);
        }
        out[i] = h;
    }
}
