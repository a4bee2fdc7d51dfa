/* TEST INFRASTRUCTURE -- per-stage timers of the CPU restatement (oracle.h: ORC_T0 / ORC_T1), used by bench_oracle.c's cpu_baseline figures.
 * A file of its own because orb_oracle.c / frame_oracle.c are also linked without bench_oracle.c (oracle/refprobe/build.py). */
#include <time.h>
#include "oracle.h"

int orc_stage_timing = 0;
__thread double orc_stage_s[ORC_NSTAGES];
double orc_now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
