/* Exhaustive check (all 2^32 floats) that   q0 = x * rc;  r = fma(-q0, c, x);  q = fma(r, rc, q0)   with rc = RN(1/c) is the
 * correctly rounded x / c for the window lengths of the event detector, c in {3, 6, 7, 14} (events.c:52-65), denormals on —
 * the float divisions of events.c:343-366 by the constant window length.  Prints the mismatching inputs by class.
 *   gcc -O2 -ffp-contract=off -mfma -fopenmp tools/proto/div_const_check.c -o /tmp/dcc -lm && /tmp/dcc */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void) {
    const float cs[4] = {3.f, 6.f, 7.f, 14.f};
    for (int k = 0; k < 4; ++k) {
        const float c = cs[k];
        volatile float one = 1.0f;
        const float rc = one / c;
        uint64_t bad = 0, bad_inf = 0, bad_sub = 0, bad_norm = 0; uint32_t first = 0;
        #pragma omp parallel for reduction(+:bad,bad_inf,bad_sub,bad_norm) schedule(static)
        for (int64_t i = 0; i < (1ll << 32); ++i) {
            const uint32_t u = (uint32_t)i;
            const float x = u2f(u);
            if (x != x) continue;                                   /* NaN in -> NaN out either way */
            const float ref = x / c;
            const float q0 = x * rc;
            const float r = fmaf(-q0, c, x);
            const float q = fmaf(r, rc, q0);
            if (f2u(q) != f2u(ref)) {
                ++bad;
                if (isinf(x)) ++bad_inf;
                else if (fabsf(ref) < 1.17549435e-38f * 4) ++bad_sub;
                else { ++bad_norm; if (!first) first = u; }
            }
        }
        printf("c = %2.0f rc = %a: mismatches %llu (x = +-inf: %llu, |x/c| < 4 FLT_MIN: %llu, elsewhere: %llu first 0x%08x)\n", c, rc,
               (unsigned long long)bad, (unsigned long long)bad_inf, (unsigned long long)bad_sub, (unsigned long long)bad_norm, first);
    }
    /* the two double divisions (sum1 / w, sumsq1 / w): not enumerable — proved in abea_kernels.hip (the quotient of a double by 3, 6,
     * 7 or 14 is representable or at least 1/14 ulp away from every rounding boundary, the sequence's value is within 2^-51 ulp of
     * it) — and sampled here: 2^32 random mantissas x 5 exponents per divisor */
    for (int k = 0; k < 4; ++k) {
        const double c = cs[k];
        volatile double one = 1.0;
        const double rc = one / c;
        uint64_t bad = 0;
        #pragma omp parallel for reduction(+:bad) schedule(static)
        for (int64_t i = 0; i < (1ll << 32); ++i) {
            uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)k;          /* splitmix64 */
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            const uint64_t mant = z & 0xFFFFFFFFFFFFFull;
            static const int ex[5] = {1023 - 149, 1023 - 20, 1023, 1023 + 20, 1023 + 132};
            for (int e = 0; e < 5; ++e) {
                uint64_t u = ((uint64_t)ex[e] << 52) | mant | ((z >> 63) << 63);
                double x; memcpy(&x, &u, 8);
                const double ref = x / c, q0 = x * rc, r = fma(-q0, c, x), q = fma(r, rc, q0);
                if (memcmp(&q, &ref, 8)) ++bad;
            }
        }
        printf("double, c = %2.0f: %llu mismatches in 5 x 2^32 samples\n", c, (unsigned long long)bad);
    }
    return 0;
}
