/* Exhaustive check of the division shortcut a candidate k_nfa_eval rewrite used (round 2: exact, but slower than the division it replaces -- DESIGN.md, open items) for the binomial tail of LSD's NFA (imgproc/lsd.cpp nfa():
 * bin_term = (double)(n - i + 1) / (double)i with small positive integers): with r = RN(1 / b) (one IEEE division, tabulated),
 *     q0 = RN(a * r);  e = fma(-b, q0, a);  q1 = fma(e, r, q0)
 * is claimed to equal RN(a / b) (Markstein's correction step; exact when b's significand is not all ones, true for integers < 2^53).
 * This program tests EVERY pair 1 <= a, b <= N (N = argv[1], default 2^18 >= the 196,608 scaled pixels of a VGA frame) against the hardware division
 * and prints the number of mismatches (N = 2^20: 0 mismatches, ~7 minutes on 8 cores).   gcc -O2 -fopenmp -ffp-contract=off tools/markstein_check.c -o /tmp/markstein_check -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : (1L << 18);
    long bad = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad)
    for (long b = 1; b <= N; b++) {
        const double db = (double)b, r = 1.0 / db;
        for (long a = 1; a <= N; a++) {
            const double da = (double)a;
            const double q0 = da * r;
            const double e = fma(-db, q0, da);
            const double q1 = fma(e, r, q0);
            if (q1 != da / db) bad++;
        }
    }
    printf("N = %ld: %ld mismatches among %ld x %ld pairs\n", N, bad, N, N);
    return bad != 0;
}
