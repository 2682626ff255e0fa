/* TEST INFRASTRUCTURE.  The fused normal-map kernels (csrc/ds_normalmap.hip, nm_store) compute the three components of
 *     normal = (zx, -zy, 1) / n,   n = sqrt(zx^2 + zy^2 + 1)                    normalmap_generation.py:34-39
 * with ONE correctly rounded division  y = 1.0 / n  (which is the z component) and, for the other two,
 *     q0 = a * y;  r = fma(-q0, n, a);  q = fma(r, y, q0)
 * instead of two more divisions.  With y = RN(1/n) this is Markstein's sequence: q = RN(a / n) unless the significand of n
 * is all ones -- impossible here (n^2 is a multiple of 2^-16 below 2^38, so n is either a power of two or at least 2^-41 away
 * from one).  This program checks q == a / n (IEEE binary64, round to nearest) on the operand set of the 3 x 3 Sobel path --
 * zx, zy multiples of 2^-8 with numerators up to 4 * 65535 -- exhaustively for numerators 0..6000 x 0..6000 and on 2 * 10^9
 * random pairs of the full range (signs are symmetric), and on the operand set of the np.gradient path (multiples of 2^-9,
 * numerators up to 65535: a 4001 x 4001 block + 10^9 random pairs).  Prints "bad=0 of N" and exits 0 when the identity holds.
 *     gcc -O2 -fopenmp -ffp-contract=off */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <omp.h>

static inline int check(double zx, double zy)
{
    const double n = sqrt(zx * zx + zy * zy + 1.0);
    volatile double yv = 1.0 / n;
    const double y = yv;
    int bad = 0;
    const double a[2] = { zx, zy };
    for (int k = 0; k < 2; k++) {
        const double q0 = a[k] * y;
        const double r = fma(-q0, n, a[k]);
        const double q = fma(r, y, q0);
        if (q != a[k] / n) bad++;
    }
    return bad;
}

int main(void)
{
    long long bad = 0, tot = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad, tot)
    for (int i = 0; i <= 6000; i++)
        for (int j = 0; j <= 6000; j++) { bad += check(i / 256.0, j / 256.0); tot += 2; }
#pragma omp parallel reduction(+ : bad, tot)
    {
        uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(omp_get_thread_num() + 1);
        const long long per = 2000000000ll / omp_get_num_threads();
        for (long long t = 0; t < per; t++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const int i = (int)(s % 262141u), j = (int)((s >> 32) % 262141u);
            bad += check(i / 256.0, j / 256.0); tot += 2;
        }
    }
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad, tot)
    for (int i = 0; i <= 4000; i++)
        for (int j = 0; j <= 4000; j++) { bad += check(i / 512.0, j / 512.0); tot += 2; }
#pragma omp parallel reduction(+ : bad, tot)
    {
        uint64_t s = 0xD1B54A32D192ED03ull * (uint64_t)(omp_get_thread_num() + 1);
        const long long per = 1000000000ll / omp_get_num_threads();
        for (long long t = 0; t < per; t++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const int i = (int)(s % 65536u), j = (int)((s >> 32) % 65536u);
            bad += check(i / 512.0, j / 512.0); tot += 2;
        }
    }
    printf("bad=%lld of %lld\n", bad, tot);
    return bad != 0;
}
