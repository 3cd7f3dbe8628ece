/* Groundwork for DESIGN.md section 9 item 3: is  q' = fma(fma(-q, d, a), r, q)  with  r = 1.0 / d,  q = a * r  always the correctly
 * rounded a / d?  (One IEEE division per LDLT pivot plus three instructions per element instead of one division per element.)
 * Exhaustive over doubles is impossible; this hammers random operands, operands with extreme mantissas (all ones, one, 1 + ulp) and
 * quotients that sit next to a rounding boundary.   gcc -O2 -march=x86-64-v3 -fopenmp -ffp-contract=off markstein_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>

static inline uint64_t rng(uint64_t* s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static inline double mk(uint64_t mant, int e) { uint64_t b = ((uint64_t)(1023 + e) << 52) | (mant & 0xFFFFFFFFFFFFFull); double d; memcpy(&d, &b, 8); return d; }

int main(void)
{
    long long bad = 0, total = 0;
#pragma omp parallel reduction(+ : bad, total)
    {
        uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(omp_get_thread_num() + 1);
        const uint64_t special[8] = {0, 1, 2, 0xFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFEull, 0x8000000000000ull, 0x7FFFFFFFFFFFFull, 0x8000000000001ull};
        for (long long it = 0; it < 40000000; ++it) {
            uint64_t ma = rng(&s), md = rng(&s);
            if ((it & 7) == 1) md = special[rng(&s) & 7];
            if ((it & 7) == 2) ma = special[rng(&s) & 7];
            const int ea = (int)(rng(&s) % 120) - 60, ed = (int)(rng(&s) % 120) - 60;
            double d = mk(md, ed), a = mk(ma, ea);
            if ((it & 7) == 3) { /* a = RN(q0 * d) for a random q0: the true quotient is then within an ulp of a representable number */
                const double q0 = mk(rng(&s), ea - ed);
                a = q0 * d;
            }
            if (rng(&s) & 1) a = -a;
            if (rng(&s) & 1) d = -d;
            const double want = a / d;
            const double r = 1.0 / d;
            const double q = a * r;
            const double e = fma(-q, d, a);
            const double got = fma(e, r, q);
            ++total;
            if (memcmp(&want, &got, 8) != 0) {
                ++bad;
                if (bad < 4) printf("mismatch a=%a d=%a want=%a got=%a\n", a, d, want, got);
            }
        }
    }
    printf("cases %lld mismatches %lld\n", total, bad);
    return bad ? 1 : 0;
}
