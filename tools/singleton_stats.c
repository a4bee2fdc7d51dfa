/* tools/singleton_stats.c -- how many LSD seeds grow a ONE-pixel region, and how many of those a static test could have
 * foreseen (no 8-neighbour whose level-line angle is within the tolerance of the seed's own angle: "static"), or a test of
 * the neighbours' flags at the seed's turn ("dynamic": every compatible neighbour already USED).  Built on the oracle's own
 * region loop (the file is included, ORC_LSD_STATS switches the counters on).  Test infrastructure, like the oracle.
 *   gcc -O2 -shared -fPIC -fopenmp -ffp-contract=off -w -DORC_LSD_STATS -I oracle tools/singleton_stats.c oracle/orb_oracle.c oracle/lbd_oracle.c oracle/timing.c -o tools/scratch/libsingle.so -lm  */
#include <stdint.h>
long orc_stat[16];   /* 0 regions, 1 one-pixel regions, 2 of them foreseen by the static neighbour test, 3 pixels accepted, 4 regions of <= 3 pixels, 5 their pixels,
                        6 seeds the static test calls single that grew more than themselves (must be 0) */
#include "../oracle/lsd_oracle.c"
long *orc_stats(void) { return orc_stat; }
