/*
 * bench_oracle.c -- CPU baseline driver: runs the oracle's whole per-frame front-end (ORB extract + LSD/LBD extract
 * + SearchByProjection against a local map + line projection search) over a stream of frames, one frame per
 * OpenMP thread, and returns the elapsed seconds.  TEST / BENCH INFRASTRUCTURE ONLY (bench.py `cpu_baseline`).
 * The reference runs each frame on one thread (ORBextractor has no internal threading; Tracking is
 * single-threaded), so "one frame per core on all cores" is its throughput configuration.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <omp.h>
#include "oracle.h"

/* (the stage timers orc_stage_timing / orc_stage_s / orc_now_s live in timing.c: orb_oracle.c is also linked alone into the reference probe) */

/* one frame through the whole front-end on the calling thread (two_threads: ORB and LSD/LBD extraction on two OpenMP threads, as the
 * PL-SLAM family's Frame constructor does); returns the checksum contribution */
static long one_frame(const uint8_t *img, int w, int h, int nfeatures, int nlines, const float *scale, const orc_mappoints *MP, const orc_maplines *ML,
                      float th, float nnratio, int two_threads, double *stage_out)
{
    const int cap = nfeatures + 64;
    orc_keypoint *kps = (orc_keypoint *)malloc(sizeof(orc_keypoint) * cap);
    uint8_t *desc = (uint8_t *)malloc((size_t)cap * 32);
    orc_keyline *kl = (orc_keyline *)malloc(sizeof(orc_keyline) * nlines);
    uint8_t *ldesc = (uint8_t *)malloc((size_t)nlines * 32);
    double *eq = (double *)malloc(sizeof(double) * 3 * nlines);
    int n = 0, nl = 0, nd = 0;
    double st_a[ORC_NSTAGES], st_b[ORC_NSTAGES];
    memset(st_a, 0, sizeof(st_a)); memset(st_b, 0, sizeof(st_b));
#pragma omp parallel sections num_threads(2) if (two_threads)
    {
#pragma omp section
        {
            memset(orc_stage_s, 0, sizeof(orc_stage_s));
            n = orc_orb_extract(img, w, h, w, nfeatures, 1.2f, 8, 20, 7, kps, desc, cap, NULL);
            memcpy(st_a, orc_stage_s, sizeof(st_a));
        }
#pragma omp section
        {
            memset(orc_stage_s, 0, sizeof(orc_stage_s));
            nl = orc_line_extract(img, w, h, w, nlines, ORC_LSD_SEED_RASTER, kl, ldesc, eq, nlines, &nd);
            memcpy(st_b, orc_stage_s, sizeof(st_b));
        }
    }
    if (n > cap) n = cap;
    memset(orc_stage_s, 0, sizeof(orc_stage_s));
    float *ux = (float *)malloc(sizeof(float) * (n + 1)), *uy = (float *)malloc(sizeof(float) * (n + 1)), *ur = (float *)malloc(sizeof(float) * (n + 1)),
          *ang = (float *)malloc(sizeof(float) * (n + 1));
    int *oc = (int *)malloc(sizeof(int) * (n + 1));
    int32_t *match = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    for (int i = 0; i < n; i++) { ux[i] = kps[i].x; uy[i] = kps[i].y; ur[i] = -1.f; ang[i] = kps[i].angle; oc[i] = kps[i].octave; match[i] = -1; }
    orc_frame F;
    F.n = n; F.ux = ux; F.uy = uy; F.octave = oc; F.uright = ur; F.desc = desc; F.angle = ang;
    F.minx = 0; F.miny = 0; F.maxx = (float)w; F.maxy = (float)h;
    F.grid_inv_w = 64.0f / (float)w; F.grid_inv_h = 48.0f / (float)h;
    F.scale_factors = scale; F.nlevels = 8;
    ORC_T0(t_mp);
    const int nm = MP ? orc_search_by_projection_map(&F, MP, th, nnratio, match) : 0;
    ORC_T1(t_mp, ORC_ST_MATCH_POINTS);
    float *px = (float *)malloc(sizeof(float) * (nl + 1)), *py = (float *)malloc(sizeof(float) * (nl + 1)), *la = (float *)malloc(sizeof(float) * (nl + 1));
    int *lo = (int *)malloc(sizeof(int) * (nl + 1));
    int32_t *lmatch = (int32_t *)malloc(sizeof(int32_t) * (nl + 1));
    for (int i = 0; i < nl; i++) { px[i] = kl[i].pt_x; py[i] = kl[i].pt_y; la[i] = kl[i].angle; lo[i] = kl[i].octave; lmatch[i] = -1; }
    orc_lineframe LF;
    LF.n = nl; LF.pt_x = px; LF.pt_y = py; LF.angle = la; LF.octave = lo; LF.desc = ldesc; LF.scale_factors = scale;
    ORC_T0(t_ml);
    const int nml = ML ? orc_search_by_projection_lines(&LF, ML, th, nnratio, lmatch) : 0;
    ORC_T1(t_ml, ORC_ST_MATCH_LINES);
    if (stage_out)
        for (int k = 0; k < ORC_NSTAGES; k++) stage_out[k] += st_a[k] + st_b[k] + orc_stage_s[k];
    free(kps); free(desc); free(kl); free(ldesc); free(eq); free(ux); free(uy); free(ur); free(ang); free(oc); free(match);
    free(px); free(py); free(la); free(lo); free(lmatch);
    return (long)n + nl + nm + nml;
}

/* BASELINE.md section 4 figure (a): ONE frame at a time, as the reference's caller loop does (Examples/RGB-D/rgbd_tum.cc:98-116 brackets
 * SLAM.TrackRGBD with steady_clock; vTimesTrack median / mean at :134-142).  `warmup` untimed frames, then n_frames timed ones: per_frame_s[i] =
 * wall seconds of frame i, stage_s[ORC_NSTAGES] = seconds per stage summed over the timed frames (thread-seconds when two_threads). */
long orc_frontend_latency(const uint8_t *imgs, int n_distinct, int w, int h, int n_frames, int warmup, int nfeatures, int nlines,
                          const orc_mappoints *MP, const orc_maplines *ML, float th, float nnratio, int two_threads, double *per_frame_s, double *stage_s)
{
    float scale[16], inv[16], s2[16], is2[16];
    int per[16], umax[16];
    orc_orb_tables(nfeatures, 1.2f, 8, scale, inv, s2, is2, per, umax);
    long sum = 0;
    for (int k = 0; k < ORC_NSTAGES; k++) stage_s[k] = 0;
    for (int f = -warmup; f < n_frames; f++) {
        const uint8_t *img = imgs + (size_t)(((f % n_distinct) + n_distinct) % n_distinct) * w * h;
        orc_stage_timing = f >= 0;
        const double t0 = orc_now_s();
        const long c = one_frame(img, w, h, nfeatures, nlines, scale, MP, ML, th, nnratio, two_threads, f >= 0 ? stage_s : NULL);
        const double t1 = orc_now_s();
        if (f >= 0) { per_frame_s[f] = t1 - t0; sum += c; }
    }
    orc_stage_timing = 0;
    return sum;
}

double orc_frontend_throughput(const uint8_t *imgs, int n_distinct, int w, int h, int n_frames, int threads, int nfeatures, int nlines,
                               const orc_mappoints *MP, const orc_maplines *ML, float th, float nnratio, long *checksum)
{
    float scale[16], inv[16], s2[16], is2[16];
    int per[16], umax[16];
    orc_orb_tables(nfeatures, 1.2f, 8, scale, inv, s2, is2, per, umax);
    long sum = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : sum)
    for (int f = 0; f < n_frames; f++) {
        const uint8_t *img = imgs + (size_t)(f % n_distinct) * w * h;
        const int cap = nfeatures + 64;
        orc_keypoint *kps = (orc_keypoint *)malloc(sizeof(orc_keypoint) * cap);
        uint8_t *desc = (uint8_t *)malloc((size_t)cap * 32);
        int n = orc_orb_extract(img, w, h, w, nfeatures, 1.2f, 8, 20, 7, kps, desc, cap, NULL);
        if (n > cap) n = cap;
        orc_keyline *kl = (orc_keyline *)malloc(sizeof(orc_keyline) * nlines);
        uint8_t *ldesc = (uint8_t *)malloc((size_t)nlines * 32);
        double *eq = (double *)malloc(sizeof(double) * 3 * nlines);
        int nd = 0;
        int nl = orc_line_extract(img, w, h, w, nlines, ORC_LSD_SEED_RASTER, kl, ldesc, eq, nlines, &nd);
        /* flat Frame view */
        float *ux = (float *)malloc(sizeof(float) * (n + 1)), *uy = (float *)malloc(sizeof(float) * (n + 1)),
              *ur = (float *)malloc(sizeof(float) * (n + 1)), *ang = (float *)malloc(sizeof(float) * (n + 1));
        int *oc = (int *)malloc(sizeof(int) * (n + 1));
        int32_t *match = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
        for (int i = 0; i < n; i++) { ux[i] = kps[i].x; uy[i] = kps[i].y; ur[i] = -1.f; ang[i] = kps[i].angle; oc[i] = kps[i].octave; match[i] = -1; }
        orc_frame F;
        F.n = n; F.ux = ux; F.uy = uy; F.octave = oc; F.uright = ur; F.desc = desc; F.angle = ang;
        F.minx = 0; F.miny = 0; F.maxx = (float)w; F.maxy = (float)h;
        F.grid_inv_w = 64.0f / (float)w; F.grid_inv_h = 48.0f / (float)h;
        F.scale_factors = scale; F.nlevels = 8;
        int nm = MP ? orc_search_by_projection_map(&F, MP, th, nnratio, match) : 0;
        /* lines */
        float *px = (float *)malloc(sizeof(float) * (nl + 1)), *py = (float *)malloc(sizeof(float) * (nl + 1)), *la = (float *)malloc(sizeof(float) * (nl + 1));
        int *lo = (int *)malloc(sizeof(int) * (nl + 1));
        int32_t *lmatch = (int32_t *)malloc(sizeof(int32_t) * (nl + 1));
        for (int i = 0; i < nl; i++) { px[i] = kl[i].pt_x; py[i] = kl[i].pt_y; la[i] = kl[i].angle; lo[i] = kl[i].octave; lmatch[i] = -1; }
        orc_lineframe LF;
        LF.n = nl; LF.pt_x = px; LF.pt_y = py; LF.angle = la; LF.octave = lo; LF.desc = ldesc; LF.scale_factors = scale;
        int nml = ML ? orc_search_by_projection_lines(&LF, ML, th, nnratio, lmatch) : 0;
        sum += n + nl + nm + nml;
        free(kps); free(desc); free(kl); free(ldesc); free(eq); free(ux); free(uy); free(ur); free(ang); free(oc); free(match);
        free(px); free(py); free(la); free(lo); free(lmatch);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (checksum) *checksum = sum;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
