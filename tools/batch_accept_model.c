/* batch_accept_model.c -- CPU model of region_grow's accept loop as the GPU runs it (groups of up to 7 centres = 63 lanes, cone pre-test on the float
 * sums) with and without BATCH-ACCEPT, to count loop trips before any kernel is changed.  Not part of any parity path; TEST / DESIGN TOOLING ONLY.
 *
 * Build + run:  tools/batch_accept_model.sh [frames]     (the script renames the oracle's region_grow and compiles this file on top of it)
 *
 * The model's region_grow replaces the oracle's (also inside refine), reproduces the reference's accept decisions (asserted against the reference function on
 * a copy of the flags for every region) and counts, per region:
 *   old trips  = 1 + accepts + exact-test rejections per group       (classification passes of the shipped loop)
 *   new trips  = passes of the batch loop described in DESIGN.md (rank-dependent margins)
 */
#include <assert.h>

#ifndef RC_CAP
#define RC_CAP 8
#endif
#ifndef RANK_ITERS
#define RANK_ITERS 1
#endif

static long g_old_trips, g_new_trips, g_accepts, g_groups, g_batches, g_batch_hist[65], g_singles, g_border, g_regions, g_mismatch, g_retrim;
static long g_lane_cands;
static float g_pc_sx, g_pc_sy;
static long g_pc_trips, g_pc_batches, g_pc_hist[65], g_pc_singles, g_pc_mismatch;
#ifndef PC_ITERS
#define PC_ITERS 1
#endif
static long g_sz_regions[8], g_sz_pixels[8], g_sz_groups[8];

typedef struct { float t1, t2; } growth_t;
static growth_t grow_thresholds(double prec)
{
    const double delta = 8.7266462599716e-4;
    growth_t t; t.t1 = NAN; t.t2 = NAN;
    if (prec - delta > 0.0 && prec + delta < 1.55) {
        t.t1 = tanf((float)(prec - delta)) * (1.0f - 1.0e-4f);
        t.t2 = tanf((float)(prec + delta)) * (1.0f + 1.0e-4f);
    }
    return t;
}

static int exact_aligned(const lsd_t *L, size_t c, double theta, double prec)
{
    const double a = L->angles[c];
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) { n_theta -= M_2__PI; if (n_theta < 0) n_theta = -n_theta; }
    return n_theta <= prec;
}

static void region_grow(lsd_t *L, int sx, int sy, regpt *reg, int *reg_size, double *reg_angle, double prec)
{
    const int W = L->w, H = L->h;
    const growth_t th = grow_thresholds(prec);
    const float k1 = (1.0f + th.t1) * 1.0001f, k2 = (1.0f + th.t2) * 1.0001f;
    int n = 1;
    size_t addr = (size_t)sy * W + sx;
    reg[0].x = sx; reg[0].y = sy; reg[0].used = L->used + addr;
    *reg_angle = L->angles[addr];
    reg[0].angle = *reg_angle; reg[0].modgrad = L->modgrad[addr];
    float sumdx = (float)cos(*reg_angle), sumdy = (float)sin(*reg_angle);
    *reg[0].used = USED;
    g_regions++;
    const long groups0 = g_groups;
    int i = 0, nx_n = 1, cur_n;
    while (1) {
        cur_n = nx_n;
        nx_n = n - (i + cur_n); if (nx_n > 7) nx_n = 7; if (nx_n < 0) nx_n = 0;
        g_groups++;
        /* lanes of the group */
        int la[63], lv[63], nl = 9 * cur_n;
        for (int s = 0; s < cur_n; s++)
            for (int k = 0; k < 9; k++) {
                const int xx = reg[i + s].x + k % 3 - 1, yy = reg[i + s].y + k / 3 - 1;
                const int ok = xx >= 0 && yy >= 0 && xx < W && yy < H;
                la[s * 9 + k] = ok ? yy * W + xx : -1;
                lv[s * 9 + k] = ok && L->used[yy * W + xx] != USED && L->angles[yy * W + xx] != NOTDEF;
            }
        for (int j = 0; j < nl; j++) g_lane_cands += lv[j];
        /* ---------------- new loop on a copy of the state; the old loop (= reference order) is what commits */
        {
            int cand[63]; memcpy(cand, lv, sizeof(cand));
            float sx_ = sumdx, sy_ = sumdy;
            long trips = 0;
            for (;;) {
                trips++;
                float dot[63], acr[63]; int r[63];
                int any = 0;
                for (int j = 0; j < nl; j++) if (cand[j]) {
                    any = 1;
                    const float ux = (float)cos((double)(float)L->angles[la[j]]), uy = (float)sin((double)(float)L->angles[la[j]]);
                    dot[j] = sx_ * ux + sy_ * uy; acr[j] = fabsf(sx_ * uy - sy_ * ux);
                }
                if (!any) break;
                /* rank bounds */
                int pos[63];
                for (int j = 0; j < nl; j++) pos[j] = cand[j] && !(acr[j] >= th.t2 * dot[j] + (float)RC_CAP * k2);
                for (int it = 0; it < RANK_ITERS; it++) {
                    int c = 0;
                    for (int j = 0; j < nl; j++) { r[j] = c; c += pos[j]; }
                    if (it + 1 < RANK_ITERS)
                        for (int j = 0; j < nl; j++) if (pos[j] && r[j] < RC_CAP && acr[j] >= th.t2 * dot[j] + (float)r[j] * k2) pos[j] = 0;
                }
                int u = nl;
                for (int j = 0; j < nl; j++) if (cand[j]) {
                    const int sa = r[j] < RC_CAP && acr[j] + (float)r[j] * k1 <= th.t1 * dot[j];
                    const int sr = r[j] < RC_CAP && acr[j] >= th.t2 * dot[j] + (float)r[j] * k2;
                    if (!sa && !sr) { u = j; break; }
                }
                /* batch = surely accepted lanes before u, first lane per pixel */
                int nb = 0;
                for (int j = 0; j < u; j++) if (cand[j] && acr[j] + (float)r[j] * k1 <= th.t1 * dot[j]) {
                    int dup = 0;
                    for (int l = 0; l < j; l++) if (cand[l] == 2 && la[l] == la[j]) dup = 1;
                    if (!dup) { cand[j] = 2; nb++; sx_ = (float)((double)sx_ + cos((double)(float)L->angles[la[j]])); sy_ = (float)((double)sy_ + sin((double)(float)L->angles[la[j]])); }
                }
                if (nb) {
                    g_batches++; g_batch_hist[nb]++;
                    for (int j = 0; j < nl; j++) {
                        if (cand[j] == 2) continue;
                        if (j < u) cand[j] = 0;
                        else for (int l = 0; l < u; l++) if (cand[l] == 2 && la[l] == la[j]) cand[j] = 0;
                    }
                    for (int j = 0; j < u; j++) if (cand[j] == 2) cand[j] = 0;
                    continue;
                }
                for (int j = 0; j < u; j++) cand[j] = 0;
                if (u == nl) break;
                /* lane u alone, zero margin */
                if (acr[u] >= th.t2 * dot[u]) { cand[u] = 0; g_retrim++; continue; }   /* (a real kernel would go on to the next lane in the same trip) */
                int acc = acr[u] <= th.t1 * dot[u];
                if (!acc) { g_border++; acc = exact_aligned(L, la[u], (double)orc_fast_atan2(sy_, sx_) * DEG_TO_RADS, prec); }
                g_singles++;
                if (acc) {
                    sx_ = (float)((double)sx_ + cos((double)(float)L->angles[la[u]])); sy_ = (float)((double)sy_ + sin((double)(float)L->angles[la[u]]));
                    for (int j = u + 1; j < nl; j++) if (la[j] == la[u]) cand[j] = 0;
                }
                cand[u] = 0;
            }
            g_new_trips += trips;
            /* ---------------- scheme 2: prefix consistency.  Guess A = lanes surely aligned with the sums at the start of the trip (first lane per pixel); every
             * candidate lane is then classified against ITS OWN sums S_j = S + (float prefix sum of the guessed lanes before it); all lanes in front of the first lane
             * whose class differs from the guess (or that is a border lane) are decided for good: the guessed lanes among them are accepted in one step. */
            {
                int cand2[63]; memcpy(cand2, lv, sizeof(cand2));
                float px_ = sumdx, py_ = sumdy;
                long trips2 = 0;
                for (;;) {
                    int any = 0;
                    for (int j = 0; j < nl; j++) any |= cand2[j];
                    if (!any) break;
                    trips2++;
                    float ux[63], uy[63];
                    int A[63], A2[63], bord[63];
                    for (int j = 0; j < nl; j++) if (cand2[j]) { ux[j] = (float)cos((double)(float)L->angles[la[j]]); uy[j] = (float)sin((double)(float)L->angles[la[j]]); }
                    for (int j = 0; j < nl; j++) {
                        A[j] = 0;
                        if (!cand2[j]) continue;
                        const float d = px_ * ux[j] + py_ * uy[j], a = fabsf(px_ * uy[j] - py_ * ux[j]);
                        A[j] = a <= th.t1 * d;
                        if (A[j]) for (int l = 0; l < j; l++) if (A[l] && la[l] == la[j]) A[j] = 0;
                    }
                    for (int it = 0; it < PC_ITERS; it++) {
                        float sx2 = px_, sy2 = py_;
                        for (int j = 0; j < nl; j++) {
                            A2[j] = 0; bord[j] = 0;
                            if (cand2[j]) {
                                const float d = sx2 * ux[j] + sy2 * uy[j], a = fabsf(sx2 * uy[j] - sy2 * ux[j]);
                                const int sa = a <= th.t1 * d, sr = a >= th.t2 * d;
                                int dup = 0;
                                for (int l = 0; l < j; l++) if (A[l] && la[l] == la[j]) dup = 1;
                                A2[j] = sa && !dup; bord[j] = !sa && !sr && !dup;
                            }
                            if (A[j]) { sx2 += ux[j]; sy2 += uy[j]; }
                        }
                        if (it + 1 < PC_ITERS) memcpy(A, A2, sizeof(A));
                    }
                    int d = nl;
                    for (int j = 0; j < nl; j++) if (cand2[j] && (A[j] != A2[j] || bord[j])) { d = j; break; }
                    int nb = 0;
                    for (int j = 0; j < d; j++) if (A2[j]) {
                        nb++;
                        px_ = (float)((double)px_ + cos((double)(float)L->angles[la[j]])); py_ = (float)((double)py_ + sin((double)(float)L->angles[la[j]]));
                    }
                    if (nb) { g_pc_batches++; g_pc_hist[nb]++; }
                    for (int j = d; j < nl; j++) if (cand2[j]) for (int l = 0; l < d; l++) if (A2[l] && la[l] == la[j]) cand2[j] = 0;
                    for (int j = 0; j < d; j++) cand2[j] = 0;
                    if (d < nl && cand2[d] && !nb) {   /* nothing accepted in front of lane d: its sums are the trip's, so it is a border lane -> the reference's own test */
                        const int acc = exact_aligned(L, la[d], (double)orc_fast_atan2(py_, px_) * DEG_TO_RADS, prec);
                        g_pc_singles++;
                        if (acc) {
                            px_ = (float)((double)px_ + cos((double)(float)L->angles[la[d]])); py_ = (float)((double)py_ + sin((double)(float)L->angles[la[d]]));
                            for (int j = d + 1; j < nl; j++) if (la[j] == la[d]) cand2[j] = 0;
                        }
                        cand2[d] = 0;
                    }
                }
                g_pc_trips += trips2;
                g_pc_sx = px_; g_pc_sy = py_;
            }

            /* the batch loop must leave the same sums as the reference order below: checked after the group */
            const int n_before = n;
            long old = 1;
            for (int j = 0; j < nl; j++) {
                if (la[j] < 0) continue;
                const size_t c = (size_t)la[j];
                if (L->used[c] != USED && L->angles[c] != NOTDEF) {
                    /* shipped loop: surely-not lanes cost nothing; accepts and exact-test rejections cost a trip */
                    const float ux = (float)cos((double)(float)L->angles[c]), uy = (float)sin((double)(float)L->angles[c]);
                    const float d = sumdx * ux + sumdy * uy, a = fabsf(sumdx * uy - sumdy * ux);
                    const int surely_not = a >= th.t2 * d, surely = a <= th.t1 * d;
                    const int al = exact_aligned(L, c, *reg_angle, prec);
                    if ((surely_not && al) || (surely && !al)) { g_mismatch++; fprintf(stderr, "mismatch: prec %.6f t1 %g t2 %g S (%g, %g) u (%g, %g) d %g a %g surely %d not %d al %d reg_angle %.6f pix angle %.6f n %d\n", prec, th.t1, th.t2, sumdx, sumdy, ux, uy, d, a, surely, surely_not, al, *reg_angle, L->angles[c], n); }
                    if (!surely_not) old++;
                    if (al) {
                        L->used[c] = USED;
                        reg[n].x = la[j] % W; reg[n].y = la[j] / W; reg[n].used = L->used + c; reg[n].modgrad = L->modgrad[c];
                        reg[n].angle = L->angles[c];
                        ++n;
                        sumdx = (float)((double)sumdx + cos((double)(float)L->angles[c]));
                        sumdy = (float)((double)sumdy + sin((double)(float)L->angles[c]));
                        *reg_angle = (double)orc_fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
            }
            g_old_trips += old;
            g_accepts += n - n_before;
            if (sx_ != sumdx || sy_ != sumdy) g_mismatch++;
            if (g_pc_sx != sumdx || g_pc_sy != sumdy) g_pc_mismatch++;
        }
        i += cur_n;
        if (nx_n == 0) {
            if (i >= n) break;
            nx_n = n - i < 7 ? n - i : 7;
        }
    }
    { int b = n <= 1 ? 0 : n <= 3 ? 1 : n <= 7 ? 2 : n <= 15 ? 3 : n <= 31 ? 4 : n <= 127 ? 5 : 6; g_sz_regions[b]++; g_sz_pixels[b] += n; g_sz_groups[b] += g_groups - groups0; }
    *reg_size = n;
}

int main(int argc, char **argv)
{
    long tot_lines = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 1; }
        int w = 640, h = 480;
        uint8_t *img = (uint8_t *)malloc((size_t)w * h);
        if (fread(img, 1, (size_t)w * h, f) != (size_t)w * h) { fprintf(stderr, "short file %s\n", argv[a]); return 1; }
        fclose(f);
        float *lines = (float *)malloc(sizeof(float) * 4 * 20000);
        const long o0 = g_old_trips, n0 = g_new_trips, a0 = g_accepts, g0 = g_groups;
        const int nl = orc_lsd_detect(img, w, h, w, 0, lines, 20000, NULL);
        tot_lines += nl;
        printf("%s: %d lines, accepts %ld, groups %ld, old trips %ld, new trips %ld (%.2f)\n", argv[a], nl, g_accepts - a0, g_groups - g0, g_old_trips - o0,
               g_new_trips - n0, (double)(g_new_trips - n0) / (double)(g_old_trips - o0));
        free(img); free(lines);
    }
    printf("RC_CAP %d RANK_ITERS %d: regions %ld groups %ld accepts %ld cand lanes %ld | old trips %ld (%.2f per group) new trips %ld (%.2f per group) ratio %.3f\n", RC_CAP,
           RANK_ITERS, g_regions, g_groups, g_accepts, g_lane_cands, g_old_trips, (double)g_old_trips / g_groups, g_new_trips, (double)g_new_trips / g_groups,
           (double)g_new_trips / g_old_trips);
    printf("  batches %ld singles %ld (border %ld) retrims %ld mismatches %ld; batch sizes:", g_batches, g_singles, g_border, g_retrim, g_mismatch);
    for (int k = 1; k < 20; k++) printf(" %d:%ld", k, g_batch_hist[k]);
    printf("\n");
    printf("  PREFIX-CONSISTENCY (iters %d): trips %ld (%.2f per group, ratio to old %.3f) batches %ld singles %ld mismatches %ld; batch sizes:", PC_ITERS, g_pc_trips, (double)g_pc_trips / g_groups, (double)g_pc_trips / g_old_trips, g_pc_batches, g_pc_singles, g_pc_mismatch);
    for (int k = 1; k < 20; k++) printf(" %d:%ld", k, g_pc_hist[k]);
    printf("\n");
    printf("  region sizes 1 | 2-3 | 4-7 | 8-15 | 16-31 | 32-127 | 128+: regions");
    for (int b = 0; b < 7; b++) printf(" %ld", g_sz_regions[b]);
    printf("  pixels");
    for (int b = 0; b < 7; b++) printf(" %ld", g_sz_pixels[b]);
    printf("  groups");
    for (int b = 0; b < 7; b++) printf(" %ld", g_sz_groups[b]);
    printf("\n");
    return g_mismatch != 0;
}
