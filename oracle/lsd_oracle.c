/*
 * lsd_oracle.c -- CPU restatement of the LSD line detector as the reference reaches it.
 * TEST INFRASTRUCTURE ONLY (same rules as orb_oracle.c).
 *
 * Reference entry: LineSegment::ExtractLineSegment, include/ExtractLineSegment.h:38 (body absent from
 * /root/reference).  It calls cv::line_descriptor::LSDDetector::detect(img, lines,
 * scale=int(1.2)=1, numOctaves=1), which runs cv::createLineSegmentDetector(LSD_REFINE_ADV)->detect
 * on octave 0.  Neither OpenCV 3.3.x imgproc/lsd.cpp nor opencv_contrib line_descriptor is
 * vendored, so this file restates their PUBLISHED algorithm (von Gioi et al., "LSD: a Line Segment
 * Detector", IPOL 2012, as implemented in OpenCV 3.3) -- PARITY UNPINNED: no reference test, fixture
 * or executable pins any number produced here.  Choices that differ between OpenCV versions are
 * parameters with the 3.3 behaviour as default:
 *   - the image is converted to double before the sigma=0.75 blur and the 0.8x INTER_LINEAR resize
 *     (3.0-3.3: `img.convertTo(image, CV_64FC1)`), gradients in double;
 *   - seeds are visited in RASTER order: 3.0-3.3 build the 1024-bin pseudo-ordering as a linked list
 *     but iterate `list[i]` by index (seed_order=ORC_LSD_SEED_RASTER).  seed_order=ORC_LSD_SEED_BINNED
 *     gives the published order (bins descending, raster inside a bin);
 *   - `sumdx += cos(float(angle))` resolves to ::cos(double) under GCC 5.4 (no std::cos in scope).
 * All arithmetic is double unless the upstream code says float; no FMA contraction (-ffp-contract=off).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define NOTDEF (-1024.0)
#define NOTUSED 0
#define USED 1
#define M_3_2_PI (3 * 3.14159265358979323846 / 2)
#define M_2__PI (2 * 3.14159265358979323846)
#define CV_PI_ 3.1415926535897932384626433832795
#define DEG_TO_RADS (CV_PI_ / 180)
#define RELATIVE_ERROR_FACTOR 100.0

typedef struct { int x, y; uint8_t *used; double angle, modgrad; } regpt;
typedef struct { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; } rect_t;

typedef struct {
    int w, h;            /* scaled image size */
    double *img;         /* scaled image */
    double *angles, *modgrad;
    uint8_t *used;
    double LOG_NT;
} lsd_t;

static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

/* cv::getGaussianKernel(n, sigma, CV_64F) */
void orc_gauss_kernel_f64(int n, double sigma, double *k)
{
    double scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        k[i] = exp(scale2X * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] *= sum;
}

/* GaussianBlur on CV_64F, separable, REFLECT_101: RowFilter then SymmColumnFilter (see header) */
static void gaussian_blur_f64(const double *src, double *dst, int w, int h, const double *k, int ksize)
{
    const int r = ksize / 2;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double s = k[0] * src[(size_t)y * w + reflect101(x - r, w)];
            for (int t = 1; t < ksize; t++) s += k[t] * src[(size_t)y * w + reflect101(x - r + t, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double s = k[r] * tmp[(size_t)y * w + x] + 0.0;
            for (int t = 1; t <= r; t++)
                s += k[r + t] * (tmp[(size_t)reflect101(y + t, h) * w + x] + tmp[(size_t)reflect101(y - t, h) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
    free(tmp);
}

/* cv::resize(src64f, dst, Size(), fx, fy, INTER_LINEAR): float coefficients, double data */
static void resize_linear_f64(const double *src, int sw, int sh, double *dst, int dw, int dh, double inv_scale_x,
                              double inv_scale_y)
{
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    float *alpha = (float *)malloc(sizeof(float) * 2 * dw);
    double *r0 = (double *)malloc(sizeof(double) * dw), *r1 = (double *)malloc(sizeof(double) * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (dx < xmax) xmax = dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        alpha[2 * dx] = 1.f - fx;
        alpha[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        const double b0 = (double)(1.f - fy), b1 = (double)fy;
        int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const double *S0 = src + (size_t)y0 * sw, *S1 = src + (size_t)y1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (dx < xmax) {
                r0[dx] = S0[sx] * alpha[2 * dx] + S0[sx + 1] * alpha[2 * dx + 1];
                r1[dx] = S1[sx] * alpha[2 * dx] + S1[sx + 1] * alpha[2 * dx + 1];
            } else {
                r0[dx] = S0[sx] * 1.0;
                r1[dx] = S1[sx] * 1.0;
            }
        }
        for (int dx = 0; dx < dw; dx++) dst[(size_t)dy * dw + dx] = r0[dx] * b0 + r1[dx] * b1;
    }
    free(xofs); free(alpha); free(r0); free(r1);
}

static inline double dist_(double x1, double y1, double x2, double y2) { return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }
static inline double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }

static inline double angle_diff_signed(double a, double b)
{
    double diff = a - b;
    while (diff <= -CV_PI_) diff += M_2__PI;
    while (diff > CV_PI_) diff -= M_2__PI;
    return diff;
}
static inline double angle_diff(double a, double b)
{
    double diff = angle_diff_signed(a, b);
    if (diff < 0) diff = -diff;
    return diff;
}

static inline int double_equal(double a, double b)
{
    if (a == b) return 1;
    double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    if (abs_max < DBL_MIN) abs_max = DBL_MIN;
    return (abs_diff / abs_max) <= (RELATIVE_ERROR_FACTOR * DBL_EPSILON);
}

static inline int is_aligned(const lsd_t *L, int x, int y, double theta, double prec)
{
    if (x < 0 || y < 0 || x >= L->w || y >= L->h) return 0;
    const double a = L->angles[(size_t)y * L->w + x];
    if (a == NOTDEF) return 0;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) {
        n_theta -= M_2__PI;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
}

static void region_grow(lsd_t *L, int sx, int sy, regpt *reg, int *reg_size, double *reg_angle, double prec)
{
    const int W = L->w, H = L->h;
    int n = 1;
    size_t addr = (size_t)sy * W + sx;
    reg[0].x = sx; reg[0].y = sy; reg[0].used = L->used + addr;
    *reg_angle = L->angles[addr];
    reg[0].angle = *reg_angle;
    reg[0].modgrad = L->modgrad[addr];
    float sumdx = (float)cos(*reg_angle);
    float sumdy = (float)sin(*reg_angle);
    *reg[0].used = USED;
    for (int i = 0; i < n; ++i) {
        const int px = reg[i].x, py = reg[i].y;
        int xx_min = px - 1 > 0 ? px - 1 : 0, xx_max = px + 1 < W - 1 ? px + 1 : W - 1;
        int yy_min = py - 1 > 0 ? py - 1 : 0, yy_max = py + 1 < H - 1 ? py + 1 : H - 1;
        for (int yy = yy_min; yy <= yy_max; ++yy) {
            for (int xx = xx_min; xx <= xx_max; ++xx) {
                size_t c = (size_t)yy * W + xx;
                if (L->used[c] != USED && is_aligned(L, xx, yy, *reg_angle, prec)) {
                    L->used[c] = USED;
                    reg[n].x = xx; reg[n].y = yy; reg[n].used = L->used + c;
                    reg[n].modgrad = L->modgrad[c];
                    const double angle = L->angles[c];
                    reg[n].angle = angle;
                    ++n;
                    /* `sumdx += cos(float(angle))`: ::cos(double) of the float-rounded angle, float accumulate */
                    sumdx = (float)((double)sumdx + cos((double)(float)angle));
                    sumdy = (float)((double)sumdy + sin((double)(float)angle));
                    *reg_angle = (double)orc_fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                }
            }
        }
    }
    *reg_size = n;
#ifdef ORC_LSD_STATS   /* tools/singleton_stats.c, tests/test_oracle_props.py: the argument behind the GPU's "static singles" (lsd_kernels.hip, singles_run) */
    {
        extern long orc_stat[16];
        orc_stat[0]++; orc_stat[3] += n;
        if (n <= 3) { orc_stat[4]++; orc_stat[5] += n; }
        /* neighbours whose angle passes the FIRST test of a region seeded here (theta = the seed's own angle), whatever their flags */
        int compat = 0;
        for (int yy = sy - 1; yy <= sy + 1; ++yy) for (int xx = sx - 1; xx <= sx + 1; ++xx)
            if ((xx != sx || yy != sy) && is_aligned(L, xx, yy, L->angles[addr], prec)) compat++;
        if (n == 1) { orc_stat[1]++; if (!compat) orc_stat[2]++; }
        if (!compat && n != 1) orc_stat[6]++;   /* must never happen: a seed without such a neighbour grows itself alone */
    }
#endif
}

static double get_theta(const regpt *reg, int reg_size, double x, double y, double reg_angle, double prec)
{
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int i = 0; i < reg_size; ++i) {
        const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
        double dx = regx - x, dy = regy - y;
        Ixx += dy * dy * weight;
        Iyy += dx * dx * weight;
        Ixy -= dx * dy * weight;
    }
    double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)orc_fast_atan2((float)(lambda - Ixx), (float)Ixy)
                                           : (double)orc_fast_atan2((float)Ixy, (float)(lambda - Iyy));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += CV_PI_;
    return theta;
}

static void region2rect(const regpt *reg, int reg_size, double reg_angle, double prec, double p, rect_t *rec)
{
    double x = 0, y = 0, sum = 0;
    for (int i = 0; i < reg_size; ++i) {
        const double weight = reg[i].modgrad;
        x += (double)reg[i].x * weight;
        y += (double)reg[i].y * weight;
        sum += weight;
    }
    x /= sum;
    y /= sum;
    double theta = get_theta(reg, reg_size, x, y, reg_angle, prec);
    double dx = cos(theta), dy = sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = 0; i < reg_size; ++i) {
        double regdx = (double)reg[i].x - x, regdy = (double)reg[i].y - y;
        double l = regdx * dx + regdy * dy;
        double w = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
        if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
    }
    rec->x1 = x + l_min * dx; rec->y1 = y + l_min * dy;
    rec->x2 = x + l_max * dx; rec->y2 = y + l_max * dy;
    rec->width = w_max - w_min;
    rec->x = x; rec->y = y; rec->theta = theta; rec->dx = dx; rec->dy = dy; rec->prec = prec; rec->p = p;
    if (rec->width < 1.0) rec->width = 1.0;
}

static int reduce_region_radius(lsd_t *L, regpt *reg, int *reg_size, double reg_angle, double prec, double p,
                                rect_t *rec, double density, double density_th)
{
    double xc = (double)reg[0].x, yc = (double)reg[0].y;
    double radSq1 = distSq(xc, yc, rec->x1, rec->y1), radSq2 = distSq(xc, yc, rec->x2, rec->y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    (void)L;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        for (int i = 0; i < *reg_size; ++i) {
            if (distSq(xc, yc, (double)reg[i].x, (double)reg[i].y) > radSq) {
                *(reg[i].used) = NOTUSED;
                regpt t = reg[i]; reg[i] = reg[*reg_size - 1]; reg[*reg_size - 1] = t;
                --(*reg_size);
                --i;
            }
        }
        if (*reg_size < 2) return 0;
        region2rect(reg, *reg_size, reg_angle, prec, p, rec);
        density = (double)*reg_size / (dist_(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    }
    return 1;
}

static int g_last_regrow_n = 0;
static int refine(lsd_t *L, regpt *reg, int *reg_size, double reg_angle, double prec, double p, rect_t *rec,
                  double density_th)
{
    double density = (double)*reg_size / (dist_(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density >= density_th) return 1;
    double xc = (double)reg[0].x, yc = (double)reg[0].y;
    const double ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (int i = 0; i < *reg_size; ++i) {
        *(reg[i].used) = NOTUSED;
        if (dist_(xc, yc, reg[i].x, reg[i].y) < rec->width) {
            const double angle = reg[i].angle;
            double ang_d = angle_diff_signed(angle, ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    double mean_angle = sum / (double)n;
    double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    region_grow(L, reg[0].x, reg[0].y, reg, reg_size, &reg_angle, tau);
    g_last_regrow_n = *reg_size;   /* (read by the band-speculation model below) */
    if (*reg_size < 2) return 0;
    region2rect(reg, *reg_size, reg_angle, prec, p, rec);
    density = (double)*reg_size / (dist_(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density < density_th) return reduce_region_radius(L, reg, reg_size, reg_angle, prec, p, rec, density, density_th);
    return 1;
}

static inline double log_gamma_windschitl(double x)
{
    return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
}
static inline double log_gamma_lanczos(double x)
{
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
        a -= log(x + (double)n);
        b += q[n] * pow(x, (double)n);
    }
    return a + log(b);
}
#define log_gamma(x) ((x) > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x))

static double nfa(const lsd_t *L, int n, int k, double p)
{
    const double LOG_NT = L->LOG_NT;
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - (double)n * log10(p);
    double p_term = p / (1 - p);
    double log1term = log_gamma((double)n + 1) - log_gamma((double)k + 1) - log_gamma((double)(n - k) + 1) +
                      (double)k * log(p) + (double)(n - k) * log(1.0 - p);
    double term = exp(log1term);
    if (double_equal(term, 0)) {
        if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;
        else return -LOG_NT;
    }
    double bin_tail = term;
    double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
        double bin_term = (double)(n - i + 1) / (double)i;
        double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            double err = term * ((1 - pow(mult_term, (double)(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * fabs(-log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - LOG_NT;
}

typedef struct { int x, y, taken; } edge_t;

static double rect_nfa(const lsd_t *L, const rect_t *rec)
{
    int total_pts = 0, alg_pts = 0;
    double half_width = rec->width / 2.0;
    double dyhw = rec->dy * half_width, dxhw = rec->dx * half_width;
    edge_t o[4];
    o[0].x = (int)(rec->x1 - dyhw); o[0].y = (int)(rec->y1 + dxhw); o[0].taken = 0;
    o[1].x = (int)(rec->x2 - dyhw); o[1].y = (int)(rec->y2 + dxhw); o[1].taken = 0;
    o[2].x = (int)(rec->x2 + dyhw); o[2].y = (int)(rec->y2 - dxhw); o[2].taken = 0;
    o[3].x = (int)(rec->x1 + dyhw); o[3].y = (int)(rec->y1 - dxhw); o[3].taken = 0;
    /* std::sort(AsmallerB_XoverY) on 4 elements == insertion sort; equal elements are identical */
    for (int i = 1; i < 4; i++) {
        edge_t t = o[i];
        int j = i - 1;
        while (j >= 0 && (o[j].x > t.x || (o[j].x == t.x && o[j].y > t.y))) { o[j + 1] = o[j]; j--; }
        o[j + 1] = t;
    }
    edge_t *min_y = &o[0], *max_y = &o[0];
    for (int i = 1; i < 4; ++i) {
        if (min_y->y > o[i].y) min_y = &o[i];
        if (max_y->y < o[i].y) max_y = &o[i];
    }
    min_y->taken = 1;
    edge_t *leftmost = 0;
    for (int i = 0; i < 4; ++i)
        if (!o[i].taken) { if (!leftmost) leftmost = &o[i]; else if (leftmost->x > o[i].x) leftmost = &o[i]; }
    leftmost->taken = 1;
    edge_t *rightmost = 0;
    for (int i = 0; i < 4; ++i)
        if (!o[i].taken) { if (!rightmost) rightmost = &o[i]; else if (rightmost->x < o[i].x) rightmost = &o[i]; }
    rightmost->taken = 1;
    edge_t *tailp = 0;
    for (int i = 0; i < 4; ++i)
        if (!o[i].taken) { if (!tailp) tailp = &o[i]; else if (tailp->x > o[i].x) tailp = &o[i]; }
    tailp->taken = 1;
    /* integer divisions and the `tailp->p.x` typo are upstream behaviour */
    double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
    double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
    double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
    double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = min_y->x, right_x = min_y->x;
    int min_iter = min_y->y, max_iter = max_y->y;
    for (int y = min_iter; y <= max_iter; ++y) {
        if (y >= 0 && y < L->h) {
            for (int x = (int)left_x; x <= (int)right_x; ++x) {
                if (x < 0 || x >= L->w) continue;
                ++total_pts;
                if (is_aligned(L, x, y, rec->theta, rec->prec)) ++alg_pts;
            }
        } else {
            /* upstream `continue`s before the step update for out-of-image rows */
            continue;
        }
        if (y >= leftmost->y) lstep = slstep;
        if (y >= rightmost->y) rstep = srstep;
        left_x += lstep;
        right_x += rstep;
    }
    return nfa(L, total_pts, alg_pts, rec->p);
}

static double rect_improve(const lsd_t *L, rect_t *rec, double LOG_EPS)
{
    double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(L, rec);
    if (log_nfa > LOG_EPS) return log_nfa;
    rect_t r = *rec;
    for (int n = 0; n < 5; ++n) {
        r.p /= 2;
        r.prec = r.p * CV_PI_;
        double v = rect_nfa(L, &r);
        if (v > log_nfa) { log_nfa = v; *rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.width -= delta;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
            r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
            r.width -= delta;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
            r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
            r.width -= delta;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.p /= 2;
            r.prec = r.p * CV_PI_;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    return log_nfa;
}

/* LineSegmentDetectorImpl::detect with LSD_REFINE_ADV and default parameters.
 * lines: x1,y1,x2,y2 (float) per segment in detection order.  Returns the segment count. */
int orc_lsd_detect(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int seed_order, float *lines, int cap,
                   orc_lsd_debug *dbg)
{
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, LOG_EPS = 0, DENSITY_TH = 0.7;
    const int N_BINS = 1024;
    if (!gray || w <= 0 || h <= 0) return 0;
    lsd_t L;
    const double prec = CV_PI_ * ANG_TH / 180, p = ANG_TH / 180;
    const double rho = QUANT / sin(prec);
    /* image -> double; blur; resize */
    double *img = (double *)malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) img[(size_t)y * w + x] = (double)gray[(size_t)y * pitch + x];
    const double sigma = SIGMA_SCALE / SCALE;
    const double sprec = 3;
    const unsigned hk = (unsigned)ceil(sigma * sqrt(2 * sprec * log(10.0)));
    const int ksize = 1 + 2 * (int)hk;
    double kern[64];
    orc_gauss_kernel_f64(ksize, sigma, kern);
    double *blur = (double *)malloc(sizeof(double) * (size_t)w * h);
    gaussian_blur_f64(img, blur, w, h, kern, ksize);
    L.w = (int)lrint(w * SCALE);
    L.h = (int)lrint(h * SCALE);
    L.img = (double *)malloc(sizeof(double) * (size_t)L.w * L.h);
    resize_linear_f64(blur, w, h, L.img, L.w, L.h, SCALE, SCALE);
    free(img); free(blur);
    const int W = L.w, H = L.h;
    const size_t NP = (size_t)W * H;
    L.angles = (double *)malloc(sizeof(double) * NP);
    L.modgrad = (double *)calloc(NP, sizeof(double)); /* cv::Mat_<double>(size) is uninitialised for the last row/col; never read */
    L.used = (uint8_t *)calloc(NP, 1);
    /* ll_angle */
    for (int x = 0; x < W; x++) L.angles[(size_t)(H - 1) * W + x] = NOTDEF;
    for (int y = 0; y < H; y++) L.angles[(size_t)y * W + (W - 1)] = NOTDEF;
    double max_grad = -1;
    for (int y = 0; y < H - 1; ++y) {
        for (int x = 0; x < W - 1; ++x) {
            size_t a = (size_t)y * W + x;
            double DA = L.img[a + W + 1] - L.img[a];
            double BC = L.img[a + 1] - L.img[a + W];
            double gx = DA + BC, gy = DA - BC;
            double norm = sqrt((gx * gx + gy * gy) / 4);
            L.modgrad[a] = norm;
            if (norm <= rho) L.angles[a] = NOTDEF;
            else {
                L.angles[a] = (double)orc_fast_atan2((float)gx, (float)-gy) * DEG_TO_RADS;
                if (norm > max_grad) max_grad = norm;
            }
        }
    }
    /* seed list */
    const size_t nseed_raster = (size_t)(W - 1) * (H - 1);
    int *seeds = (int *)malloc(sizeof(int) * (NP + 1));
    size_t nseeds = 0;
    if (seed_order == ORC_LSD_SEED_BINNED) {
        double bin_coef = (max_grad > 0) ? (double)(N_BINS - 1) / max_grad : 0;
        int *cnt = (int *)calloc(N_BINS + 1, sizeof(int));
        int *bin = (int *)malloc(sizeof(int) * nseed_raster);
        size_t q = 0;
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x, ++q) {
                int b = (int)(L.modgrad[(size_t)y * W + x] * bin_coef);
                if (b > N_BINS - 1) b = N_BINS - 1;
                bin[q] = b;
                cnt[N_BINS - 1 - b + 1]++; /* descending bins */
            }
        for (int b = 0; b < N_BINS; b++) cnt[b + 1] += cnt[b];
        q = 0;
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x, ++q) seeds[cnt[N_BINS - 1 - bin[q]]++] = y * W + x;
        nseeds = nseed_raster;
        free(cnt); free(bin);
    } else {
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) seeds[nseeds++] = y * W + x;
        /* upstream also visits w*h - (w-1)(h-1) default-constructed entries at (0,0): no effect */
    }
    L.LOG_NT = 5 * (log10((double)W) + log10((double)H)) / 2 + log10(11.0);
    const int min_reg_size = (int)(-L.LOG_NT / log10(p));
    regpt *reg = (regpt *)malloc(sizeof(regpt) * NP);
    int nlines = 0, nregions = 0;
    for (size_t i = 0; i < nseeds; ++i) {
        const int adx = seeds[i];
        if (L.used[adx] == NOTUSED && L.angles[adx] != NOTDEF) {
            int reg_size;
            double reg_angle;
            region_grow(&L, adx % W, adx / W, reg, &reg_size, &reg_angle, prec);
            nregions++;
            if (reg_size < min_reg_size) continue;
            rect_t rec;
            region2rect(reg, reg_size, reg_angle, prec, p, &rec);
            if (!refine(&L, reg, &reg_size, reg_angle, prec, p, &rec, DENSITY_TH)) continue;
            double log_nfa = rect_improve(&L, &rec, LOG_EPS);
            if (log_nfa <= LOG_EPS) continue;
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
            if (nlines < cap) {
                lines[4 * nlines] = (float)rec.x1; lines[4 * nlines + 1] = (float)rec.y1;
                lines[4 * nlines + 2] = (float)rec.x2; lines[4 * nlines + 3] = (float)rec.y2;
            }
            nlines++;
        }
    }
    if (dbg) {
        dbg->sw = W; dbg->sh = H; dbg->nregions = nregions; dbg->min_reg_size = min_reg_size; dbg->max_grad = max_grad;
        if (dbg->want_maps) {
            dbg->scaled = L.img; dbg->angles = L.angles; dbg->modgrad = L.modgrad; dbg->used = L.used;
            L.img = NULL; L.angles = NULL; L.modgrad = NULL; L.used = NULL;
        }
    }
    free(reg); free(seeds); free(L.img); free(L.angles); free(L.modgrad); free(L.used);
    return nlines;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Model of "banded speculative region growing" (DESIGN.md section 7): an exact parallelisation of the serial seed loop
 * above, evaluated here on the CPU before any kernel is written.  Not part of any parity path.
 *
 * Phase 1 (parallel over K row bands): band b runs the whole per-seed pipeline (grow, rect, refine) over ITS seeds in raster
 * order against a private, initially empty USED map -- i.e. it speculates that no earlier band touches what it touches.  Each
 * effective seed leaves a record: seed, every pixel the pipeline ever accepted ("touched"), the pixels still marked at the
 * end, the rectangle if one came out.
 * Phase 2 (bands in order): T = true USED map so far, S = band b's private map replayed along its timeline, D = S xor T.
 * Walking the band's pixels in raster order: a recorded region stands iff its seed is free in T and no pixel of the 3x3
 * dilation of its touched set is in D (then every test it made saw the true value) -- it is committed by copying its marks;
 * otherwise it is regrown serially on T.  A seed that speculation skipped but that is free in T is grown serially as well.
 * D is updated with the symmetric difference of the speculative and the true marks.
 * The model checks that T and the emitted rectangles equal the serial run bit for bit and counts accept steps:
 * stats[0] serial accepts, [1] max over bands of speculative accepts (phase 1 critical path), [2] accepts redone serially in
 * phase 2, [3] regions, [4] regions redone or new, [5] touched pixels validated, [6] 1 if identical to the serial result. */
typedef struct { int seed, t0, nt, has_rect; rect_t rec; } band_rec;

static int run_seed(lsd_t *L, int adx, regpt *reg, double prec, double p, int min_reg_size, int *touched, int *ntouched, rect_t *rec, long *accepts)
{
    int reg_size;
    double reg_angle;
    region_grow(L, adx % L->w, adx / L->w, reg, &reg_size, &reg_angle, prec);
    *accepts += reg_size;
    int nt = 0;
    for (int i = 0; i < reg_size; i++) touched[nt++] = reg[i].y * L->w + reg[i].x;
    *ntouched = nt;
    if (reg_size < min_reg_size) return 0;
    region2rect(reg, reg_size, reg_angle, prec, p, rec);
    g_last_regrow_n = -1;
    const int ok = refine(L, reg, &reg_size, reg_angle, prec, p, rec, 0.7);
    if (g_last_regrow_n >= 0) {   /* the regrown list (reduce_region_radius only permutes it and shortens reg_size) */
        *accepts += g_last_regrow_n;
        for (int i = 0; i < g_last_regrow_n; i++) touched[nt++] = reg[i].y * L->w + reg[i].x;
        *ntouched = nt;
    }
    return ok;
}

static int g_band_halo = 0;   /* rows above a band that its speculation grows first, unrecorded, to start from a realistic flag state */
void orc_lsd_band_speculation_halo(int rows) { g_band_halo = rows; }

int orc_lsd_band_speculation(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nbands, long *stats)
{
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5;
    lsd_t L;
    const double prec = CV_PI_ * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / sin(prec);
    double *img = (double *)malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) img[(size_t)y * w + x] = (double)gray[(size_t)y * pitch + x];
    const double sigma = SIGMA_SCALE / SCALE;
    const int ksize = 1 + 2 * (int)(unsigned)ceil(sigma * sqrt(2 * 3 * log(10.0)));
    double kern[64];
    orc_gauss_kernel_f64(ksize, sigma, kern);
    double *blur = (double *)malloc(sizeof(double) * (size_t)w * h);
    gaussian_blur_f64(img, blur, w, h, kern, ksize);
    L.w = (int)lrint(w * SCALE); L.h = (int)lrint(h * SCALE);
    L.img = (double *)malloc(sizeof(double) * (size_t)L.w * L.h);
    resize_linear_f64(blur, w, h, L.img, L.w, L.h, SCALE, SCALE);
    free(img); free(blur);
    const int W = L.w, H = L.h;
    const size_t NP = (size_t)W * H;
    L.angles = (double *)malloc(sizeof(double) * NP);
    L.modgrad = (double *)calloc(NP, sizeof(double));
    for (int x = 0; x < W; x++) L.angles[(size_t)(H - 1) * W + x] = NOTDEF;
    for (int y = 0; y < H; y++) L.angles[(size_t)y * W + (W - 1)] = NOTDEF;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            size_t a = (size_t)y * W + x;
            double DA = L.img[a + W + 1] - L.img[a], BC = L.img[a + 1] - L.img[a + W];
            double gx = DA + BC, gy = DA - BC, norm = sqrt((gx * gx + gy * gy) / 4);
            L.modgrad[a] = norm;
            L.angles[a] = norm <= rho ? NOTDEF : (double)orc_fast_atan2((float)gx, (float)-gy) * DEG_TO_RADS;
        }
    L.LOG_NT = 5 * (log10((double)W) + log10((double)H)) / 2 + log10(11.0);
    const int min_reg_size = (int)(-L.LOG_NT / log10(p));
    regpt *reg = (regpt *)malloc(sizeof(regpt) * NP);
    int *touched = (int *)malloc(sizeof(int) * 2 * NP);
    memset(stats, 0, sizeof(long) * 8);
    /* ---- serial run */
    uint8_t *used_serial = (uint8_t *)calloc(NP, 1);
    rect_t *rects_serial = (rect_t *)malloc(sizeof(rect_t) * NP / 4);
    int nrect_serial = 0;
    L.used = used_serial;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            const int adx = y * W + x;
            if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
            int nt; rect_t rec;
            stats[3]++;
            if (run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &stats[0])) rects_serial[nrect_serial++] = rec;
        }
    /* ---- phase 1: speculation per band */
    if (nbands < 1) nbands = 1;
    const int rows = H - 1;
    band_rec **recs = (band_rec **)calloc(nbands, sizeof(band_rec *));
    int *nrecs = (int *)calloc(nbands, sizeof(int));
    int **tl = (int **)calloc(nbands, sizeof(int *));
    uint8_t *priv = (uint8_t *)malloc(NP);
    uint8_t **halo = (uint8_t **)calloc(nbands, sizeof(uint8_t *));
    for (int b = 0; b < nbands; b++) {
        const int y0 = (int)((long)rows * b / nbands), y1 = (int)((long)rows * (b + 1) / nbands);
        memset(priv, 0, NP);
        L.used = priv;
        recs[b] = (band_rec *)malloc(sizeof(band_rec) * ((size_t)(y1 - y0) * W + 1));
        size_t tcap = 4 * NP, tn = 0;
        tl[b] = (int *)malloc(sizeof(int) * tcap);
        long acc = 0;
        if (b > 0 && g_band_halo > 0) {   /* warm-up over the rows just above the band; what it marks is the band's initial speculative state */
            const int yh = y0 - g_band_halo > 0 ? y0 - g_band_halo : 0;
            for (int y = yh; y < y0; ++y)
                for (int x = 0; x < W - 1; ++x) {
                    const int adx = y * W + x;
                    if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
                    int nt; rect_t rec;
                    run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &acc);
                }
        }
        halo[b] = (uint8_t *)malloc(NP);
        memcpy(halo[b], priv, NP);
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W - 1; ++x) {
                const int adx = y * W + x;
                if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
                band_rec *r = &recs[b][nrecs[b]++];
                int nt;
                r->seed = adx;
                r->has_rect = run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &r->rec, &acc);
                if (tn + (size_t)nt > tcap) { tcap = 2 * (tn + nt); tl[b] = (int *)realloc(tl[b], sizeof(int) * tcap); }
                r->t0 = (int)tn; r->nt = nt;
                /* keep, per touched pixel, whether it is still marked at the end: sign bit */
                for (int i = 0; i < nt; i++) tl[b][tn++] = touched[i] | (L.used[touched[i]] == USED ? (int)0x40000000 : 0);
            }
        if (acc > stats[1]) stats[1] = acc;
    }
    /* ---- phase 2: commit in band order */
    uint8_t *T = (uint8_t *)calloc(NP, 1), *S = (uint8_t *)malloc(NP), *D = (uint8_t *)malloc(NP);
    rect_t *rects_par = (rect_t *)malloc(sizeof(rect_t) * NP / 4);
    int nrect_par = 0;
    for (int b = 0; b < nbands; b++) {
        const int y0 = (int)((long)rows * b / nbands), y1 = (int)((long)rows * (b + 1) / nbands);
        memcpy(S, halo[b], NP);
        for (size_t i = 0; i < NP; i++) D[i] = S[i] != T[i];
        int ri = 0;
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W - 1; ++x) {
                const int adx = y * W + x;
                const band_rec *r = (ri < nrecs[b] && recs[b][ri].seed == adx) ? &recs[b][ri] : NULL;
                if (r) ri++;
                const int true_eff = T[adx] == NOTUSED && L.angles[adx] != NOTDEF;
                if (!r && !true_eff) continue;
                int valid = r && true_eff;
                if (valid) {
                    stats[5] += r->nt;
                    for (int i = 0; i < r->nt && valid; i++) {
                        const int q = tl[b][r->t0 + i] & 0x3FFFFFFF, qx = q % W, qy = q / W;
                        for (int dy = -1; dy <= 1 && valid; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                const int xx = qx + dx, yy = qy + dy;
                                if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
                                if (D[(size_t)yy * W + xx]) { valid = 0; break; }
                            }
                    }
                }
                if (valid) {   /* commit the speculative marks and rectangle */
                    for (int i = 0; i < r->nt; i++) { const int e = tl[b][r->t0 + i]; if (e & 0x40000000) { T[e & 0x3FFFFFFF] = USED; S[e & 0x3FFFFFFF] = USED; } }
                    if (r->has_rect) rects_par[nrect_par++] = r->rec;
                    continue;
                }
                if (r)   /* the speculative timeline moves on with its own marks */
                    for (int i = 0; i < r->nt; i++) { const int e = tl[b][r->t0 + i]; if (e & 0x40000000) { const int q = e & 0x3FFFFFFF; S[q] = USED; D[q] = S[q] != T[q]; } }
                if (true_eff) {   /* grow on the true state */
                    int nt; rect_t rec;
                    L.used = T;
                    stats[4]++;
                    if (run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &stats[2])) rects_par[nrect_par++] = rec;
                    for (int i = 0; i < nt; i++) D[touched[i]] = S[touched[i]] != T[touched[i]];
                }
            }
    }
    stats[6] = nrect_par == nrect_serial && memcmp(T, used_serial, NP) == 0 && memcmp(rects_par, rects_serial, sizeof(rect_t) * nrect_serial) == 0;
    stats[7] = nrect_serial;
    for (int b = 0; b < nbands; b++) { free(recs[b]); free(tl[b]); free(halo[b]); }
    free(halo);
    free(recs); free(nrecs); free(tl); free(priv); free(T); free(S); free(D); free(rects_par); free(rects_serial); free(used_serial);
    free(reg); free(touched); free(L.img); free(L.angles); free(L.modgrad);
    return (int)stats[6];
}

/* ---------------------------------------------------------------------------------------------------------------
 * Model of the round-3 scheme, "banded speculation with PARALLEL validation rounds" (DESIGN.md section 5): phase 1 as above but WITHOUT halo
 * rows (every band speculates against an empty map, so it does no extra work); the serial commit wave is replaced by rounds in which EVERY band
 * validates itself, all bands at once: band b takes E'_b = the union of what the bands before it mark according to their CURRENT logs, compares it
 * with E_b (the union its log was last made consistent with) and, where they differ, walks its records exactly like phase 2 above (stands / regrown
 * on the true flags / new seeds), which leaves a log that is the serial processing of the band's seeds from E'_b.  Band 0 never changes, so after
 * round r the bands 0..r are final: at most nbands rounds; the rounds stop when no band's marks changed.  The fixpoint is the serial result
 * (induction over the bands).  Not part of any parity path.
 * stats[0] serial accepts, [1] max over bands of speculative accepts (phase 1 critical path), [2] sum over rounds of (max over bands of accepts
 * redone in the round) = critical path of the rounds, [3] total accepts redone, [4] rounds until nothing changed (the last round only confirms),
 * [5] band validations that had to walk their records, [6] 1 if identical to the serial result, [7] regions redone or new, summed */
typedef struct { band_rec *recs; int nrecs, cap; int *tl; size_t tn, tcap; uint8_t *out, *E; rect_t *rects; int nrect; } band_log;

static void band_log_push(band_log *B, int seed, int has_rect, const rect_t *rec, const int *touched, int nt, const uint8_t *used)
{
    if (B->nrecs == B->cap) { B->cap = B->cap ? 2 * B->cap : 1024; B->recs = (band_rec *)realloc(B->recs, sizeof(band_rec) * B->cap); }
    if (B->tn + (size_t)nt > B->tcap) { B->tcap = 2 * (B->tn + nt) + 1024; B->tl = (int *)realloc(B->tl, sizeof(int) * B->tcap); }
    band_rec *r = &B->recs[B->nrecs++];
    r->seed = seed; r->has_rect = has_rect; r->t0 = (int)B->tn; r->nt = nt;
    if (has_rect) r->rec = *rec;
    for (int i = 0; i < nt; i++) B->tl[B->tn++] = touched[i] | (used[touched[i]] == USED ? (int)0x40000000 : 0);
}

static int g_rounds_mode = 0, g_rounds_refined = 0;
static long g_rounds_needless = 0;
long orc_lsd_band_rounds_needless(void) { const long v = g_rounds_needless; g_rounds_needless = 0; return v; }
static double g_fill_tol = 1.0;
void orc_lsd_band_rounds_fill_tol(double t) { g_fill_tol = t; }
void orc_lsd_band_rounds_refined(int m) { g_rounds_refined = m; }
void orc_lsd_band_rounds_mode(int m) { g_rounds_mode = m; }
/* model experiments: explicit band boundaries (n = bands + 1 row numbers, first 0, last rows; NULL / 0: the defined-pixel balance) and the phase-1 accepts per band */
static int g_rounds_by[258], g_rounds_nby = 0;
static long g_rounds_band_acc[257];
void orc_lsd_band_rounds_set_bounds(const int *by, int n) { g_rounds_nby = (by && n >= 2 && n <= 258) ? n : 0; for (int i = 0; i < g_rounds_nby; i++) g_rounds_by[i] = by[i]; }
void orc_lsd_band_rounds_get_band_accepts(long *out, int n) { for (int i = 0; i < n && i < 257; i++) out[i] = g_rounds_band_acc[i]; }

int orc_lsd_band_rounds(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nbands, long *stats)
{
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5;
    lsd_t L;
    const double prec = CV_PI_ * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / sin(prec);
    double *img = (double *)malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) img[(size_t)y * w + x] = (double)gray[(size_t)y * pitch + x];
    const double sigma = SIGMA_SCALE / SCALE;
    const int ksize = 1 + 2 * (int)(unsigned)ceil(sigma * sqrt(2 * 3 * log(10.0)));
    double kern[64];
    orc_gauss_kernel_f64(ksize, sigma, kern);
    double *blur = (double *)malloc(sizeof(double) * (size_t)w * h);
    gaussian_blur_f64(img, blur, w, h, kern, ksize);
    L.w = (int)lrint(w * SCALE); L.h = (int)lrint(h * SCALE);
    L.img = (double *)malloc(sizeof(double) * (size_t)L.w * L.h);
    resize_linear_f64(blur, w, h, L.img, L.w, L.h, SCALE, SCALE);
    free(img); free(blur);
    const int W = L.w, H = L.h;
    const size_t NP = (size_t)W * H;
    L.angles = (double *)malloc(sizeof(double) * NP);
    L.modgrad = (double *)calloc(NP, sizeof(double));
    for (int x = 0; x < W; x++) L.angles[(size_t)(H - 1) * W + x] = NOTDEF;
    for (int y = 0; y < H; y++) L.angles[(size_t)y * W + (W - 1)] = NOTDEF;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            size_t a = (size_t)y * W + x;
            double DA = L.img[a + W + 1] - L.img[a], BC = L.img[a + 1] - L.img[a + W];
            double gx = DA + BC, gy = DA - BC, norm = sqrt((gx * gx + gy * gy) / 4);
            L.modgrad[a] = norm;
            L.angles[a] = norm <= rho ? NOTDEF : (double)orc_fast_atan2((float)gx, (float)-gy) * DEG_TO_RADS;
        }
    L.LOG_NT = 5 * (log10((double)W) + log10((double)H)) / 2 + log10(11.0);
    const int min_reg_size = (int)(-L.LOG_NT / log10(p));
    regpt *reg = (regpt *)malloc(sizeof(regpt) * NP);
    int *touched = (int *)malloc(sizeof(int) * 2 * NP);
    memset(stats, 0, sizeof(long) * 8);
    /* ---- serial run */
    uint8_t *used_serial = (uint8_t *)calloc(NP, 1);
    rect_t *rects_serial = (rect_t *)malloc(sizeof(rect_t) * NP / 4);
    int nrect_serial = 0;
    L.used = used_serial;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            const int adx = y * W + x;
            if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
            int nt; rect_t rec;
            if (run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &stats[0])) rects_serial[nrect_serial++] = rec;
        }
    /* ---- bands: equal shares of the defined pixels, boundaries on multiples of 8 rows (as k_lsd_spec_bands) */
    const int rows = H - 1, units = (rows + 7) / 8;
    if (nbands < 1) nbands = 1;
    if (nbands > units) nbands = units;
    int *by = (int *)calloc(nbands + 1, sizeof(int));
    {
        long *cnt = (long *)calloc(units, sizeof(long)), total = 0;
        for (int y = 0; y < rows; y++) for (int x = 0; x < W - 1; x++) if (L.angles[(size_t)y * W + x] != NOTDEF) { cnt[y >> 3]++; total++; }
        long acc = 0; int u = 0;
        for (int b = 1; b < nbands; b++) {
            const long target = total * b / nbands;
            while (u < units && acc + cnt[u] / 2 < target) acc += cnt[u++];
            int uu = u, umin = (by[b - 1] >> 3) + 1;
            if (uu < umin) uu = umin;
            if (uu > units - (nbands - b)) uu = units - (nbands - b);
            by[b] = uu * 8 < rows ? uu * 8 : rows;
        }
        by[nbands] = rows;
        free(cnt);
    }
    if (g_rounds_nby == nbands + 1) for (int b = 0; b <= nbands; b++) by[b] = g_rounds_by[b] < rows ? g_rounds_by[b] : rows;
    /* ---- phase 1: every band against an empty map */
    band_log *B = (band_log *)calloc(nbands, sizeof(band_log)), *B2 = (band_log *)calloc(nbands, sizeof(band_log));
    uint8_t *priv = (uint8_t *)malloc(NP);
    for (int b = 0; b < nbands; b++) {
        memset(priv, 0, NP);
        /* initial guess of what the earlier bands mark: nothing (mode 0), or every defined pixel above the band (mode 1) -- in the serial run a defined
         * pixel above the current seed is free only if refine / reduce_region_radius released it again */
        if (g_rounds_mode == 1) for (int y = 0; y < by[b]; y++) for (int x = 0; x < W; x++) if (L.angles[(size_t)y * W + x] != NOTDEF) priv[(size_t)y * W + x] = USED;
        L.used = priv;
        long acc = 0;
        if (g_rounds_mode == 4 && b > 0) {
            /* mode 4: NO warm-up growth.  The guess: every defined pixel above the band is taken, and so is every pixel of the band's first g_band_halo rows that
             * hangs on such a pixel through a chain of 8-neighbours whose level-line angles differ by at most prec (pairwise: a cheap stand-in for "the region from
             * above pokes down to here"), found by a row-by-row fill with two horizontal sweeps per row */
            for (int y = 0; y < by[b]; y++) for (int x = 0; x < W; x++) if (L.angles[(size_t)y * W + x] != NOTDEF) priv[(size_t)y * W + x] = USED;
            const int yd = by[b] + g_band_halo < H - 1 ? by[b] + g_band_halo : H - 1;
            for (int y = by[b]; y < yd; y++) {
                for (int x = 0; x < W - 1; x++) {
                    const size_t a = (size_t)y * W + x;
                    if (L.angles[a] == NOTDEF) continue;
                    for (int dx = -1; dx <= 1; dx++) {
                        const int xx = x + dx;
                        if (xx < 0 || xx >= W) continue;
                        const size_t q = (size_t)(y - 1) * W + xx;
                        if (priv[q] == USED && L.angles[q] != NOTDEF && angle_diff(L.angles[a], L.angles[q]) <= prec * g_fill_tol) { priv[a] = USED; break; }
                    }
                }
                for (int x = 1; x < W - 1; x++) {
                    const size_t a = (size_t)y * W + x;
                    if (priv[a] != USED && L.angles[a] != NOTDEF && priv[a - 1] == USED && L.angles[a - 1] != NOTDEF && angle_diff(L.angles[a], L.angles[a - 1]) <= prec * g_fill_tol) priv[a] = USED;
                }
                for (int x = W - 3; x >= 0; x--) {
                    const size_t a = (size_t)y * W + x;
                    if (priv[a] != USED && L.angles[a] != NOTDEF && priv[a + 1] == USED && L.angles[a + 1] != NOTDEF && angle_diff(L.angles[a], L.angles[a + 1]) <= prec * g_fill_tol) priv[a] = USED;
                }
            }
        }
        if ((g_rounds_mode == 2 || g_rounds_mode == 3) && b > 0) {   /* the GPU's halo warm-up: the g_band_halo rows above the band grown first, unrecorded, on an empty map */
            const int yh = by[b] - g_band_halo > 0 ? by[b] - g_band_halo : 0;
            /* mode 3: ... on a map that has every defined pixel ABOVE the warm-up rows marked (in the serial run they all are, bar the few that refine released):
             * the warm-up regions cannot leak upwards, which cost more than the warm-up rows themselves */
            if (g_rounds_mode == 3) for (int y = 0; y < yh; y++) for (int x = 0; x < W; x++) if (L.angles[(size_t)y * W + x] != NOTDEF) priv[(size_t)y * W + x] = USED;
            for (int y = yh; y < by[b]; ++y)
                for (int x = 0; x < W - 1; ++x) {
                    const int adx = y * W + x;
                    if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
                    int nt; rect_t rec;
                    run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &acc);
                }
        }
        B[b].E = (uint8_t *)malloc(NP); memcpy(B[b].E, priv, NP);
        B[b].rects = (rect_t *)malloc(sizeof(rect_t) * NP / 4);
        for (int y = by[b]; y < by[b + 1]; ++y)
            for (int x = 0; x < W - 1; ++x) {
                const int adx = y * W + x;
                if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
                int nt; rect_t rec;
                const int ok = run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &acc);
                band_log_push(&B[b], adx, ok, &rec, touched, nt, L.used);
                if (ok) B[b].rects[B[b].nrect++] = rec;
            }
        B[b].out = (uint8_t *)malloc(NP);
        for (size_t i = 0; i < NP; i++) B[b].out[i] = (priv[i] == USED && B[b].E[i] != USED) ? USED : NOTUSED;
        if (getenv("ORC_ROUNDS_VERBOSE")) fprintf(stderr, "  band %d rows %d-%d: %ld accepts, %d records\n", b, by[b], by[b + 1], acc, B[b].nrecs);
        if (acc > stats[1]) stats[1] = acc;
        if (b < 257) g_rounds_band_acc[b] = acc;
    }
    /* ---- rounds */
    uint8_t *T = (uint8_t *)malloc(NP), *S = (uint8_t *)malloc(NP), *D = (uint8_t *)malloc(NP), *En = (uint8_t *)malloc(NP);
    uint8_t **newout = (uint8_t **)calloc(nbands, sizeof(uint8_t *));
    int rounds = 0;
    for (;;) {
        int changed = 0;
        long round_max = 0;
        rounds++;
        memset(En, 0, NP);   /* running union of the outs of the bands before b, all taken from the state BEFORE this round (Jacobi) */
        for (int b = 0; b < nbands; b++) {
            newout[b] = NULL;
            if (b > 0) for (size_t i = 0; i < NP; i++) En[i] |= B[b - 1].out[i];
            if (memcmp(En, B[b].E, NP) == 0) continue;   /* consistent already */
            stats[5]++;
            memcpy(T, En, NP); memcpy(S, B[b].E, NP);
            for (size_t i = 0; i < NP; i++) D[i] = S[i] != T[i];
            band_log N; memset(&N, 0, sizeof(N));
            N.rects = (rect_t *)malloc(sizeof(rect_t) * NP / 4);
            long redo = 0;
            int ri = 0;
            for (int y = by[b]; y < by[b + 1]; ++y)
                for (int x = 0; x < W - 1; ++x) {
                    const int adx = y * W + x;
                    const band_rec *r = (ri < B[b].nrecs && B[b].recs[ri].seed == adx) ? &B[b].recs[ri] : NULL;
                    if (r) ri++;
                    const int true_eff = T[adx] == NOTUSED && L.angles[adx] != NOTDEF;
                    if (!r && !true_eff) continue;
                    int valid = r && true_eff;
                    if (valid)
                        for (int i = 0; i < r->nt && valid; i++) {
                            const int q = B[b].tl[r->t0 + i] & 0x3FFFFFFF, qx = q % W, qy = q / W;
                            for (int dy = -1; dy <= 1 && valid; dy++)
                                for (int dx = -1; dx <= 1; dx++) {
                                    const int xx = qx + dx, yy = qy + dy;
                                    if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
                                    const size_t a = (size_t)yy * W + xx;
                                    if (!D[a]) continue;
                                    /* refined rule (g_rounds_refined): a neighbour the speculation saw FREE and that is truly USED changes nothing unless the
                                     * record ACCEPTED it (a free neighbour it rejected for its angle is skipped in the true run: same outcome) */
                                    if (g_rounds_refined && T[a] == USED && !(dx == 0 && dy == 0)) continue;
                                    valid = 0; break;
                                }
                        }
                    if (valid) {
                        for (int i = 0; i < r->nt; i++) { const int e = B[b].tl[r->t0 + i]; if (e & 0x40000000) { T[e & 0x3FFFFFFF] = USED; S[e & 0x3FFFFFFF] = USED; } }
                        /* the record moves to the new log unchanged */
                        for (int i = 0; i < r->nt; i++) touched[i] = B[b].tl[r->t0 + i] & 0x3FFFFFFF;
                        band_log_push(&N, adx, r->has_rect, &r->rec, touched, r->nt, T);
                        /* (marks: the pixels flagged in the old log are exactly those now USED in T among the touched ones) */
                        if (r->has_rect) N.rects[N.nrect++] = r->rec;
                        continue;
                    }
                    if (r)
                        for (int i = 0; i < r->nt; i++) { const int e = B[b].tl[r->t0 + i]; if (e & 0x40000000) { const int q = e & 0x3FFFFFFF; S[q] = USED; D[q] = S[q] != T[q]; } }
                    if (true_eff) {
                        int nt; rect_t rec;
                        L.used = T;
                        stats[7]++;
                        const long redo_before = redo;
                        const int ok = run_seed(&L, adx, reg, prec, p, min_reg_size, touched, &nt, &rec, &redo);
                        if (r && r->nt == nt) {   /* (diagnostics: accepts redone for a record that came out exactly as it was -- what a perfect validity test would save) */
                            int same_rec = 1;
                            for (int i = 0; i < nt && same_rec; i++) same_rec = (B[b].tl[r->t0 + i] & 0x3FFFFFFF) == touched[i];
                            if (same_rec) g_rounds_needless += redo - redo_before;
                        }
                        band_log_push(&N, adx, ok, &rec, touched, nt, T);
                        if (ok) N.rects[N.nrect++] = rec;
                        for (int i = 0; i < nt; i++) D[touched[i]] = S[touched[i]] != T[touched[i]];
                    }
                }
            /* what the band marks = T minus what the earlier bands mark */
            N.out = (uint8_t *)malloc(NP);
            for (size_t i = 0; i < NP; i++) N.out[i] = (T[i] == USED && En[i] != USED) ? USED : NOTUSED;
            N.E = (uint8_t *)malloc(NP); memcpy(N.E, En, NP);
            if (memcmp(N.out, B[b].out, NP) != 0) changed = 1;
            B2[b] = N; newout[b] = N.out;
            stats[3] += redo;
            if (redo > round_max) round_max = redo;
        }
        for (int b = 0; b < nbands; b++)
            if (newout[b]) { free(B[b].recs); free(B[b].tl); free(B[b].out); free(B[b].E); free(B[b].rects); B[b] = B2[b]; }
        stats[2] += round_max;
        if (getenv("ORC_ROUNDS_VERBOSE")) fprintf(stderr, "  round %d: max redo %ld, changed %d\n", rounds, round_max, changed);
        if (!changed || rounds > nbands + 1) break;
    }
    stats[4] = rounds;
    /* ---- result: the bands' rectangles in band order, the union of their marks */
    int nrect_par = 0, same = 1;
    memset(T, 0, NP);
    for (int b = 0; b < nbands; b++) {
        for (int i = 0; i < B[b].nrect; i++, nrect_par++)
            if (nrect_par >= nrect_serial || memcmp(&B[b].rects[i], &rects_serial[nrect_par], sizeof(rect_t)) != 0) same = 0;
        for (size_t i = 0; i < NP; i++) T[i] |= B[b].out[i];
    }
    stats[6] = same && nrect_par == nrect_serial && memcmp(T, used_serial, NP) == 0;
    for (int b = 0; b < nbands; b++) { free(B[b].recs); free(B[b].tl); free(B[b].out); free(B[b].E); free(B[b].rects); }
    free(B); free(B2); free(newout); free(by); free(priv); free(T); free(S); free(D); free(En); free(rects_serial); free(used_serial);
    free(reg); free(touched); free(L.img); free(L.angles); free(L.modgrad);
    return (int)stats[6];
}

/* cv::line_descriptor::LSDDetector::detectImpl KeyLine fill (octave 0, octaveScale = 1) */
int orc_keylines_from_segments(const float *lines, int n, int w, int h, orc_keyline *out)
{
    for (int k = 0; k < n; k++) {
        float e0 = lines[4 * k], e1 = lines[4 * k + 1], e2 = lines[4 * k + 2], e3 = lines[4 * k + 3];
        /* checkLineExtremes */
        if (e0 < 0) e0 = 0;
        if (e0 >= w) e0 = (float)w - 1.0f;
        if (e2 < 0) e2 = 0;
        if (e2 >= w) e2 = (float)w - 1.0f;
        if (e1 < 0) e1 = 0;
        if (e1 >= h) e1 = (float)h - 1.0f;
        if (e3 < 0) e3 = 0;
        if (e3 >= h) e3 = (float)h - 1.0f;
        orc_keyline *kl = &out[k];
        const float octaveScale = 1.0f; /* pow((float)scale, 0) */
        kl->startPointX = e0 * octaveScale; kl->startPointY = e1 * octaveScale;
        kl->endPointX = e2 * octaveScale; kl->endPointY = e3 * octaveScale;
        kl->sPointInOctaveX = e0; kl->sPointInOctaveY = e1; kl->ePointInOctaveX = e2; kl->ePointInOctaveY = e3;
        kl->lineLength = (float)sqrt((double)(e0 - e2) * (double)(e0 - e2) + (double)(e1 - e3) * (double)(e1 - e3));
        /* LineIterator(img, Point(cvRound..), Point(cvRound..)), 8-connected: count = max(|dx|,|dy|)+1 */
        int x0 = (int)lrintf(e0), y0 = (int)lrintf(e1), x1 = (int)lrintf(e2), y1 = (int)lrintf(e3);
        int dx = abs(x1 - x0), dy = abs(y1 - y0);
        kl->numOfPixels = (dx > dy ? dx : dy) + 1;
        kl->angle = (float)atan2((double)(kl->endPointY - kl->startPointY), (double)(kl->endPointX - kl->startPointX));
        kl->class_id = k;
        kl->octave = 0;
        kl->size = (kl->endPointX - kl->startPointX) * (kl->endPointY - kl->startPointY);
        kl->response = kl->lineLength / (float)(w > h ? w : h);
        kl->pt_x = (kl->endPointX + kl->startPointX) / 2;
        kl->pt_y = (kl->endPointY + kl->startPointY) / 2;
    }
    return n;
}
