// lsd_kernels.hip -- LSD line-segment detector on gfx950, restating what the reference reaches through
// LineSegment::ExtractLineSegment -> cv::line_descriptor::LSDDetector::detect -> cv::LineSegmentDetector
// (include/ExtractLineSegment.h:38; OpenCV 3.3 imgproc/lsd.cpp, LSD_REFINE_ADV, default parameters).
//
// Stage map (B frames per launch):
//   k_lsd_pre                           GaussianBlur(7x7, sigma 0.75) on the image converted to double, resize(0.8,
//                                       INTER_LINEAR), ll_angle: 2x2 gradient, modgrad, level-line angle (cv::fastAtan2)
//   k_lsd_regions                       seed scan + region_grow + region2rect + refine   (ORDER-DEPENDENT, see below)
//   k_nfa_init / k_nfa_count / k_nfa_math   rect_improve = 5 search stages of (pixel count, NFA math) per rectangle
//   k_lsd_finalize                      ordered compaction -> segments -> KeyLine fields -> top-N by response
//
// The only inherently sequential part of LSD is k_lsd_regions: regions are grown greedily in seed order, every
// accepted pixel changes the running region angle (float sums + fastAtan2), and `used` marks made by one region
// (and un-made by refine) decide what later seeds see.  That chain is kept EXACTLY: one wave owns one frame and
// performs the accept steps in the reference order, while its 64 lanes cooperate on everything that is
// order-free inside a step (the 3x3 neighbourhood tests, pixel gathers, min/max extents, the rectangle pixel
// counts).  Floating-point sums whose order matters are accumulated serially in the reference order.
// The NFA validation does not touch `used`, so it is split off and runs one lane per rectangle.
// Frames are independent, so a batch keeps 1 wave x B frames busy (the batch is the parallel axis).
#include "plf_common.h"
#include "lsd_geom.h"

#define NOTDEF_F (-1024.0f)
// issue priority of the THROUGHPUT kernels of the line stream (k_lsd_pre, the NFA kernels): the line stream is the critical path of the large-batch step and its
// front / tail run beside ORB tiles (priority 2) and matcher waves (0)
#ifndef PLF_LINE_PRIO
#define PLF_LINE_PRIO 0
#endif
#define PLF_LINE_SETPRIO() do { if (PLF_LINE_PRIO) __builtin_amdgcn_s_setprio(PLF_LINE_PRIO); } while (0)
#ifndef PLF_SPEC_PF_REFINE
#define PLF_SPEC_PF_REFINE 1   // (experiment: fetch-ahead in refine's regrowth of the band waves, STG == 0)
#endif
#define PI_D 3.1415926535897932384626433832795
#define M_3_2_PI_D (3 * 3.14159265358979323846 / 2)
#define M_2__PI_D (2 * 3.14159265358979323846)
#define DEG2RAD_D (PI_D / 180)

// ------------------------------------------------------------------------------------------------
// k_lsd_pre: GaussianBlur(7x7, sigma 0.75) of the image converted to double, resize(0.8, INTER_LINEAR) with float
// coefficients, and ll_angle, fused per 64 x 16 tile of the scaled image; every intermediate lives in LDS
// (identical operation order to cv::RowFilter / SymmColumnFilter / resize / ll_angle, see oracle/lsd_oracle.c).
//   row pass   tmp(y, x)  = sum_i k[i] * double(u8(y, refl101(x - 3 + i)))                      i = 0..6, in this order
//   col pass   blur(y, x) = k[3] * tmp(y, x) + 0.0, then += k[3+i] * (tmp(refl(y+i)) + tmp(refl(y-i)))   i = 1..3
//   resize     r0 = S0[sx] * a.x + S0[sx+1] * a.y (or S0[sx] * 1.0 for dx >= xmax), same for r1; out = r0 * b.x + r1 * b.y
//   ll_angle   2x2 gradient, modgrad (double), level-line angle in DEGREES as cv::fastAtan2 returns it (the reference
//              stores double(deg) * DEG_TO_RADS, recomputed on use), NOTDEF_F where the gradient is too small or on
//              the last row/column.  cs: cos and sin of the float-rounded angle, the increments region_grow adds to
//              its sums.  cs0: float(cos(angle)), float(sin(angle)) of the un-rounded angle, the initial sums of a
//              region seeded here.
// A tile of 65 x 17 scaled pixels (one more column/row for the gradient) needs at most PRE_SC x PRE_SR blurred source
// pixels (checked on the host for the actual geometry) and 6 more rows of the row pass.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long __attribute__((aligned(1))) plf_u64u;   // 8-byte access at byte alignment (legal on gfx950 global memory)
typedef uint32_t __attribute__((aligned(1))) plf_u32u_pre;
// (PRE_TW, PRE_TH, PRE_SC, PRE_SR, PRE_NT: lsd_geom.h -- the host sizes the launch and checks the geometry with the same constants)
__global__ void __launch_bounds__(PRE_NT) k_lsd_pre(const uint8_t *__restrict__ in, ptrdiff_t pitch, ptrdiff_t fstride, float *__restrict__ ang,
                                                 double *__restrict__ modgrad, double2 *__restrict__ cs, float2 *__restrict__ cs0, LsdGeom g,
                                                 LsdTaps t, const int *__restrict__ xofs, const float2 *__restrict__ xa,
                                                 const int *__restrict__ yofs, const float2 *__restrict__ yb, int *__restrict__ defcount)
{
    PLF_LINE_SETPRIO();
    // LDS: the row-pass tile (22.5 KB; the column pass overwrites it IN PLACE with the blurred tile -- every thread first reads the 14 rows behind its
    // 8 outputs into registers, one barrier, then writes -- and the list of defined pixels reuses it at the end) + the scaled tile (8.8 KB): 31 KB per
    // workgroup, 4 resident tiles of 8 waves per CU (a separate 18 KB blurred tile made it 40 KB / 3 tiles)
    constexpr int PRE_TMP = (PRE_SR + 12) * PRE_SC * 8 >= PRE_TW * PRE_TH * 20 ? (PRE_SR + 12) * PRE_SC : (PRE_TW * PRE_TH * 20 + 7) / 8;
    __shared__ double s_tmp[PRE_TMP];   // (+6 rows for the row pass, +6 that only keep the column pass's unconditional reads in bounds)
    __shared__ double s_sc[(PRE_TH + 1) * (PRE_TW + 1)];
    double *s_blur = s_tmp;   // blurred row r at s_tmp row r (after the column pass)
    const int f = blockIdx.z, tid = threadIdx.x;
    const int dx0 = blockIdx.x * PRE_TW, dy0 = blockIdx.y * PRE_TH;
    const int dx1 = min(dx0 + PRE_TW, g.sw - 1), dy1 = min(dy0 + PRE_TH, g.sh - 1);
    const int c_lo = xofs[dx0], c_hi = min(xofs[dx1] + 1, g.w - 1);
    const int r_lo = min(max(yofs[dy0], 0), g.h - 1), r_hi = min(max(yofs[dy1] + 1, 0), g.h - 1);
    const int nc = c_hi - c_lo + 1, nr = r_hi - r_lo + 1;
    const uint8_t *img = in + (size_t)f * fstride;
    // row pass, one item = 4 consecutive columns of one row: their 10 source bytes are converted once (7 conversions per output before: the pass
    // was a third of the kernel's instructions); per output the same products and the same order of additions as cv::RowFilter
    for (int i = tid; i < (nr + 6) * (PRE_SC / 4); i += PRE_NT) {
        const int r = i / (PRE_SC / 4), c = 4 * (i - r * (PRE_SC / 4));
        if (c >= nc) continue;
        const uint8_t *row = img + (size_t)plf_reflect101(r_lo - 3 + r, g.h) * pitch;
        const int x = c_lo + c;
        double o[4];
        if (x >= 3 && x + 8 < g.w) {   // interior: bytes x-3 .. x+6 from one unaligned 8-byte and one 4-byte load
            const unsigned long long lo8 = *(const plf_u64u *)(row + x - 3);
            const uint32_t hi4 = *(const plf_u32u_pre *)(row + x + 5);
            double d[10];
#pragma unroll
            for (int q = 0; q < 8; q++) d[q] = (double)(int)((lo8 >> (8 * q)) & 0xFF);
            d[8] = (double)(int)(hi4 & 0xFF); d[9] = (double)(int)((hi4 >> 8) & 0xFF);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double s_ = t.k[0] * d[j];
#pragma unroll
                for (int q = 1; q < 7; q++) s_ += t.k[q] * d[j + q];
                o[j] = s_;
            }
        } else if (g.w >= 16) {
            // image border (the first / last groups of a row): the 10 reflected source bytes are picked out of one 16-byte window that lies inside the
            // row.  (A per-tap reflected byte load here -- 28 loads, ~450 instructions -- ran on every wave of the tiles at the left / right image edge,
            // because each of their waves holds a border group: a third of the kernel's time, 13.5 -> 9.1 ms per 4096 frames.)
            const int base = min(max(x - 3, 0), g.w - 16);
            const unsigned long long w0 = *(const plf_u64u *)(row + base), w1 = *(const plf_u64u *)(row + base + 8);
            double d[10];
#pragma unroll
            for (int q = 0; q < 10; q++) {
                const int pq = plf_reflect101(x - 3 + q, g.w) - base;
                d[q] = (double)(int)((((pq & 8) ? w1 : w0) >> (8 * (pq & 7))) & 0xFF);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double s_ = t.k[0] * d[j];
#pragma unroll
                for (int q = 1; q < 7; q++) s_ += t.k[q] * d[j + q];
                o[j] = s_;
            }
        } else {   // images narrower than 16 pixels
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                double s_ = t.k[0] * (double)row[plf_reflect101(x + j - 3, g.w)];
#pragma unroll 1
                for (int q = 1; q < 7; q++) s_ += t.k[q] * (double)row[plf_reflect101(x + j - 3 + q, g.w)];
                o[j] = s_;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s_tmp[r * PRE_SC + c + j] = o[j];
    }
    __syncthreads();
    {
        // column pass, one item = (column c, band of 8 blurred rows): at most 4 bands x 88 columns <= PRE_NT items, so one round
        static_assert(((PRE_SR + 7) / 8) * PRE_SC <= PRE_NT, "one column-pass item per thread");
        const int cb = tid / PRE_SC, cc = tid - cb * PRE_SC, rb0 = 8 * cb;
        const bool item = cc < nc && rb0 < nr;
        double v[14];
        if (item) {
            // (no row tests: the 6 spare rows of s_tmp keep rb0 + 13 inside the array; rows past nr + 6 hold stale values and only feed outputs nobody reads)
#pragma unroll
            for (int j = 0; j < 14; j++) v[j] = s_tmp[(rb0 + j) * PRE_SC + cc];
        }
        __syncthreads();
        if (item) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                double s_ = t.k[3] * v[j + 3] + 0.0;
#pragma unroll
                for (int q = 1; q <= 3; q++) s_ += t.k[3 + q] * (v[j + 3 + q] + v[j + 3 - q]);
                s_blur[(rb0 + j) * PRE_SC + cc] = s_;
            }
        }
    }
    __syncthreads();
    const int ncs = dx1 - dx0 + 1, nrs = dy1 - dy0 + 1;
    {
        // resize: a lane owns one column of the scaled tile (its source column and coefficients are loaded once), a wave walks rows -- the row's source rows
        // and coefficients are wave-uniform (scalar loads); per pixel two LDS reads of two doubles and the nine products / sums of cv::resize.
        // (One item per (row, column) with the four table loads per pixel was 60 instructions per pixel, a sixth of the kernel.)
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        auto px = [&](int rx, int sx, double ax, double ay, bool lin, int ry, int sy, float2 b) {
            const double *S0 = s_blur + (min(max(sy, 0), g.h - 1) - r_lo) * PRE_SC + sx, *S1 = s_blur + (min(max(sy + 1, 0), g.h - 1) - r_lo) * PRE_SC + sx;
            double r0, r1;
            if (lin) {
                r0 = S0[0] * ax + S0[1] * ay;
                r1 = S1[0] * ax + S1[1] * ay;
            } else {
                r0 = S0[0] * 1.0;
                r1 = S1[0] * 1.0;
            }
            s_sc[ry * (PRE_TW + 1) + rx] = r0 * (double)b.x + r1 * (double)b.y;
        };
        if (lane < ncs) {
            const int dx = dx0 + lane, sx = xofs[dx] - c_lo;
            const float2 a = xa[dx];
            const double ax = (double)a.x, ay = (double)a.y;
            const bool lin = dx < g.xmax;
            for (int ry = wv; ry < nrs; ry += PRE_NT / 64) px(lane, sx, ax, ay, lin, ry, yofs[dy0 + ry], yb[dy0 + ry]);
        }
        if (wv == PRE_NT / 64 - 1 && ncs == PRE_TW + 1 && lane < nrs) {   // the 65th column (the gradient's right neighbour), one lane per row
            const int dx = dx0 + PRE_TW;
            const float2 a = xa[dx];
            px(PRE_TW, xofs[dx] - c_lo, (double)a.x, (double)a.y, dx < g.xmax, lane, yofs[dy0 + lane], yb[dy0 + lane]);
        }
    }
    __syncthreads();
    // ll_angle for the tile.  Only the 2x2 gradient and its squared norm are computed for every pixel: norm = sqrt(q) <= rho (no level-line angle) is decided
    // on q itself -- sqrt is correctly rounded and monotone, so sqrt(q) <= rho  <=>  q <= g.rho_q, the largest double whose root does not exceed rho (found
    // on the host) -- and the pixels with a defined angle (about a quarter) are compacted (wave ballots) into LDS lists, so that the double-precision sqrt,
    // fastAtan2 and sincos below run on full waves of defined pixels only.  modgrad is written for those pixels alone: nothing reads it elsewhere
    // (region2rect weighs region points, k_lsd_maxgrad / k_lsd_seedkeys test the angle first).
    __shared__ int s_ndef;
    __shared__ float s_angt[PRE_TW * PRE_TH];                    // the tile's level-line angles (NOTDEF_F where there is none): the static singles below
    __shared__ unsigned long long s_rowbits[PRE_TH];             // static singles of the tile, one word per tile row
    double *s_q = s_tmp;                                         // the blurred tile is dead after the resize: q, offset inside the frame, (float)gx, (float)-gy
    int *s_off = reinterpret_cast<int *>(s_tmp + PRE_TW * PRE_TH);
    float *s_fx = reinterpret_cast<float *>(s_off + PRE_TW * PRE_TH), *s_fy = s_fx + PRE_TW * PRE_TH;
    if (tid == 0) s_ndef = 0;
    if (tid < PRE_TH) s_rowbits[tid] = 0ull;
    __syncthreads();
    const int tx = tid & 63;
    float *angf = ang + (size_t)f * g.s_stride;
#pragma unroll
    for (int k = 0; k < (PRE_TH + PRE_NT / 64 - 1) / (PRE_NT / 64); k++) {
        const int ty = __builtin_amdgcn_readfirstlane(tid >> 6) + (PRE_NT / 64) * k;
        if (PRE_TH % (PRE_NT / 64) != 0 && ty >= PRE_TH) break;
        const int x = dx0 + tx, y = dy0 + ty;
        bool def = false;
        double q = 0.0;
        float fx = 0.f, fy = 0.f;
        s_angt[ty * PRE_TW + tx] = NOTDEF_F;
        if (x < g.sw && y < g.sh) {
            if (x < g.sw - 1 && y < g.sh - 1) {
                const double *im = s_sc + ty * (PRE_TW + 1) + tx;
                const double DA = im[PRE_TW + 2] - im[0];
                const double BC = im[1] - im[PRE_TW + 1];
                const double gx = DA + BC, gy = DA - BC;
                q = (gx * gx + gy * gy) / 4;
                def = q > g.rho_q;
                fx = (float)gx; fy = (float)-gy;
            }
            if (!def) angf[(unsigned)(y * g.sw + x)] = NOTDEF_F;
        }
        const unsigned long long m = __ballot(def);
        int base = 0;
        if (plf_lane() == 0 && m) base = atomicAdd(&s_ndef, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (def) {
            const int e = base + __popcll(m & ((1ull << plf_lane()) - 1ull));
            s_q[e] = q; s_off[e] = ty * PRE_TW + tx; s_fx[e] = fx; s_fy[e] = fy;   // (the position inside the tile: the frame offset follows from it)
        }
    }
    __syncthreads();
    const int ndef = s_ndef;
    // pixels with a level-line angle = the length of the frame's region-growing chain to within the few pixels refine releases (every one of them ends up USED):
    // the cost by which k_lsd_balance deals the frames of a large batch to the waves of k_lsd_regions2
    if (defcount && tid == 0 && ndef) atomicAdd(&defcount[f], ndef);
    double *mgf = modgrad + (size_t)f * g.s_stride;
    double2 *csf = cs + (size_t)f * g.s_stride;
    float2 *cs0f = cs0 + (size_t)f * g.s_stride;
    for (int i = tid; i < ndef; i += PRE_NT) {
        const int loc = s_off[i];
        const unsigned o = (unsigned)((dy0 + loc / PRE_TW) * g.sw + dx0 + loc % PRE_TW);   // (32-bit offsets from the frame's scalar base pointers)
        mgf[o] = sqrt(s_q[i]);
        const float deg = plf_fast_atan2_1div(s_fx[i], s_fy[i]);
        angf[o] = deg;
        s_angt[loc] = deg;
        const double ad = (double)deg * DEG2RAD_D;
        const double af = (double)(float)ad;
        double sf, cf;
        sincos(af, &sf, &cf);   // cs: cos / sin of the FLOAT-rounded angle, the increments region_grow adds
        csf[o] = make_double2(cf, sf);
        // cs0: float(cos(ad)), float(sin(ad)) of the un-rounded angle.  ad = af + eps with |eps| <= 2^-25 |ad|: the second-order expansion around af is
        // within ~2 ulp (double) of cos / sin (ad), i.e. as close to the correctly rounded value as a second libm call is, at a twentieth of its cost
        const double eps = ad - af, h = 0.5 * eps * eps;
        cs0f[o] = make_float2((float)(cf - eps * sf - h * cf), (float)(sf + eps * cf - h * sf));
    }
    // Static singles (LsdGeom::sgl; see singles_run in the region-growing section): one BIT per pixel, set iff the pixel has an angle and none of its 8
    // neighbours can pass the first alignment test of a region seeded there -- no neighbour's angle is within the tolerance + 0.001 degrees of the pixel's
    // (circular difference in single precision: a superset of the reference's double test).  Pixels on the rim of this tile (neighbours outside: unknown, 11 % of
    // the pixels) never get the bit.  Interior pixels read their neighbours from the tile of angles without bounds tests; the bits of a tile row are collected
    // in one 64-bit LDS word and written as (up to) two dwords -- a byte per pixel, stored from the compacted lanes, cost the kernel 1.5 ms per 8192 natural
    // frames in partial-line writes alone.
#ifndef PLF_PRE_NOSGL
    if (g.sgl) {
        __syncthreads();
        const float tol = (float)(g.prec * (180.0 / PI_D)) + 1.0e-3f;
        for (int i = tid; i < ndef; i += PRE_NT) {
            const int loc = s_off[i], ty = loc / PRE_TW, tx = loc % PRE_TW;
            if (tx > 0 && tx < PRE_TW - 1 && ty > 0 && ty < PRE_TH - 1) {
                const float a = s_angt[loc];
                const float *nb = s_angt + loc;
                float best = 1.0e9f;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int off = (k < 3 ? k - 1 - PRE_TW : k == 3 ? -1 : k == 4 ? 1 : k - 6 + PRE_TW);
                    const float b = nb[off];
                    const float d = fabsf(a - b), e = 360.f - d;
                    const float m = d < e ? d : e;
                    best = (b >= 0.f && m < best) ? m : best;
                }
                if (!(best < tol)) atomicOr(&s_rowbits[ty], 1ull << tx);
            }
        }
        __syncthreads();
        if (tid < min(PRE_TH, g.sh - dy0)) {
            // (row tid of the tile = 64 bits at pixel index (dy0 + tid) * sw + dx0 of the frame's bitmap)
            const unsigned long long bits = s_rowbits[tid];
            uint32_t *bmf = g.sgl + (size_t)f * (g.s_stride >> 5);
            const unsigned idx = (unsigned)((dy0 + tid) * g.sw + dx0);
            if ((g.sw & 31) == 0) {   // the words lie inside the row and belong to this tile alone: plain stores, zero words included (nothing clears the map)
                bmf[idx >> 5] = (uint32_t)bits;
                if (dx0 + 32 < g.sw) bmf[(idx >> 5) + 1] = (uint32_t)(bits >> 32);
            } else if (bits) {        // any width: the host cleared the map, the row's bits are OR-ed into up to three words
                const unsigned o = idx & 31u;
                atomicOr(&bmf[idx >> 5], (uint32_t)(bits << o));
                const unsigned long long hi = o ? bits >> (32 - o) : bits >> 32;
                if ((uint32_t)hi) atomicOr(&bmf[(idx >> 5) + 1], (uint32_t)hi);
                if (o && (uint32_t)(hi >> 32)) atomicOr(&bmf[(idx >> 5) + 2], (uint32_t)(hi >> 32));
            }
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Published LSD seed order (seed_order = 1; every OpenCV release except 3.0-3.3): pixels sorted by gradient-magnitude
// bin, strongest bin first, raster order inside a bin (a stable counting sort upstream).  The key packs
// (1023 - bin) << 20 | pixel, so an ascending sort of the keys of a frame is exactly that order.
// bin = int(modgrad * (1023 / max_grad)), max_grad over the pixels with a defined angle (oracle/lsd_oracle.c).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lsd_maxgrad(const float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                                     double *__restrict__ maxgrad, LsdGeom g)
{
    __shared__ double red[256];
    const int f = blockIdx.x, t = threadIdx.x, NP = g.sw * g.sh;
    const float *ang = ang_all + (size_t)f * g.s_stride;
    const double *mg = modgrad_all + (size_t)f * g.s_stride;
    double m = -1.0;
    for (int a = t; a < NP; a += 256)
        if (ang[a] != NOTDEF_F && mg[a] > m) m = mg[a];
    red[t] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o && red[t + o] > red[t]) red[t] = red[t + o];
        __syncthreads();
    }
    if (t == 0) maxgrad[f] = red[0];
}

__global__ void __launch_bounds__(256) k_lsd_seedkeys(const float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double *__restrict__ maxgrad,
                                                      uint32_t *__restrict__ keys_all, LsdGeom g)
{
    const int a = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y, NP = g.sw * g.sh;
    if (a >= NP) return;
    const double mx = maxgrad[f];
    const double bin_coef = (mx > 0) ? 1023.0 / mx : 0.0;
    const int x = a % g.sw, y = a / g.sw;
    int b = 0;   // pixels without a level-line angle (the last row / column among them) never seed and have no modgrad: parked in the weakest bin
    if (x < g.sw - 1 && y < g.sh - 1 && ang_all[(size_t)f * g.s_stride + a] != NOTDEF_F) {
        b = (int)(modgrad_all[(size_t)f * g.s_stride + a] * bin_coef);
        if (b > 1023) b = 1023;
    }
    keys_all[(size_t)f * g.s_stride + a] = ((uint32_t)(1023 - b) << 20) | (uint32_t)a;
}

// diagnostics (plf_line_chain_lengths): pixels of a frame that region growing left marked USED (sign bit of the angle word of a defined pixel) = the accept
// steps of the frame's chain minus what refine / reduce_region_radius released again.  Counted after the fact so that the chain kernel carries no counter.
__global__ void __launch_bounds__(256) k_lsd_count_used(const float *__restrict__ ang_all, int *__restrict__ out, LsdGeom g)
{
    __shared__ int red[256];
    const int f = blockIdx.x, t = threadIdx.x, NP = g.sw * g.sh;
    const uint32_t *ang = reinterpret_cast<const uint32_t *>(ang_all) + (size_t)f * g.s_stride;
    int c = 0;
    for (int a = t; a < NP; a += 256) { const uint32_t w = ang[a]; c += (w >= 0x80000000u && w != __float_as_uint(NOTDEF_F)) ? 1 : 0; }
    red[t] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    if (t == 0) out[f] = red[0];
}

// ------------------------------------------------------------------------------------------------
// region growing (one wave per frame)
// ------------------------------------------------------------------------------------------------
// LDS pointers carry their address space in the type: a pointer that may be LDS or global would be lowered to FLAT
// accesses, whose `s_waitcnt vmcnt(0)` also drains the outstanding neighbourhood loads.
#define LDS_PTR(T) __attribute__((address_space(3))) T *
// The USED flag of a pixel is the SIGN BIT of its entry in the frame's angle map (angles are in [0, 360), NOTDEF is
// -1024): a pixel is a candidate iff its word is < 0x80000000.  The map lives in HBM/L2, not LDS, so that a
// workgroup needs only a few KB of LDS and a CU hosts many frames at once -- the kernel is a chain of dependent
// memory accesses per frame, and its throughput is the number of frames in flight.  ONE wave owns a frame for the whole kernel, so its flag updates and reads are
// relaxed WORKGROUP-scope atomics, i.e. plain loads and stores that the compiler may neither cache in registers nor reorder: per-location program order is all that
// is needed.  (Rounds 1-3 used device scope: on gfx950 that marks every access sc1 -- the loads bypass the CU's L1 and the 4-byte stores are written THROUGH the L2,
// 64 bytes of HBM write traffic each: 3.9 MB per frame, a third of the kernel's traffic.)  k_nfa_count strips the sign afterwards.
struct RegCtx {
    int W, H;
    uint32_t *ang;             // angle map words (float bits), sign bit = USED
    const double *modgrad;
    const double2 *cs;
    const float2 *cs0;
    const uint32_t *sgl;       // bitmap of the frame's static singles (k_lsd_pre): pixels with an angle none of whose neighbours can pass the first test of a region seeded there
    int cbase;                 // first pixel of the 64-pixel seed chunk being scanned (regions_body; chunk_taken)
    LDS_PTR(uint32_t) rxy_l;   // LDS part of the region list (x | y << 16)
    uint32_t *rxy_g;           // global overflow of the region list (entries >= rcap)
    int rcap;
    int gcap;                  // entries of rxy_g (scratch beyond the list: reduce_region_radius)
    int use_bm;                // speculative mode: the USED flags live in an LDS bitmap (bm), the angle words are read-only
    LDS_PTR(uint32_t) bm;
    int regrow_n;              // size of the list refine() regrew (-1: it did not regrow)
    LDS_PTR(double) stg;       // 96 doubles of staging for region2rect's ordered sums (32 points x 3), or null: the sums are then replayed with v_readlane
    unsigned long long t_dead; // time budget (plf_line_params.max_ms): wall_clock64() value after which the frame stops; 0 = no budget (a constant in every
                               // kernel without the budget, so the test folds away)
};

#ifdef PLF_LSD_TIMING
__device__ long long g_lsd_t[24];
#define TIC(v) const long long v = clock64()
#define TOC(slot, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lsd_t[slot] += clock64() - (v); } while (0)
#ifdef PLF_LSD_TIMING_NOCNT
#define CNT(slot, k)
#else
#define CNT(slot, k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lsd_t[slot] += (k); } while (0)   // (a global read-modify-write: counts distort the cycle figures)
#endif
// (band waves of the speculative schedule: band PLF_TIMING_BAND of frame 0)
#ifndef PLF_TIMING_BAND
#define PLF_TIMING_BAND 20
#endif
#define TOCB(slot, v) do { if (blockIdx.x == PLF_TIMING_BAND && blockIdx.y == 0 && threadIdx.x == 0) g_lsd_t[slot] += clock64() - (v); } while (0)
#define CNTB(slot, k) do { if (blockIdx.x == PLF_TIMING_BAND && blockIdx.y == 0 && threadIdx.x == 0) g_lsd_t[slot] += (k); } while (0)
#else
#define CNTB(slot, k)
#define TIC(v)
#define TOC(slot, v)
#define CNT(slot, k)
#define TOCB(slot, v)
#endif
#define CBAR() asm volatile("" ::: "memory")   // single-wave kernel: LDS ops stay in program order; only the compiler must not reorder

// (bitmap mode returns the same word the flag-in-sign-bit mode would hold: every caller is unchanged)
__device__ __forceinline__ uint32_t ang_load(const RegCtx &C, int a)
{
    if (C.use_bm) {
        uint32_t w = C.ang[a];
        if ((C.bm[a >> 5] >> (a & 31)) & 1u) w |= 0x80000000u;
        return w;
    }
    return __hip_atomic_load(&C.ang[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float ang_value(uint32_t w) { return __uint_as_float(w & 0x7FFFFFFFu); }   // of a defined pixel
// w = the pixel's angle word as this wave last read it (sign clear: it was a candidate).  The wave that owns the frame is the only writer of its
// map, so "set the sign bit" is a plain 4-byte store of the known word -- no read-modify-write at the L2 -- and per-location program order keeps
// its own later loads coherent with it (relaxed device-scope store / loads).
__device__ __forceinline__ void used_set(RegCtx &C, int a, uint32_t w)
{
    if (C.use_bm) { __hip_atomic_fetch_or(&C.bm[a >> 5], 1u << (a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
    __hip_atomic_store(&C.ang[a], w | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void used_clr(RegCtx &C, int a)
{
    if (C.use_bm) { __hip_atomic_fetch_and(&C.bm[a >> 5], ~(1u << (a & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
    // (a load and a store, not a read-modify-write at the L2: the L2 would change the word behind the back of the CU's L1, which the plain loads above may hit)
    const uint32_t w = __hip_atomic_load(&C.ang[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&C.ang[a], w & 0x7FFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t rxy_get(const RegCtx &C, int i)
{
    uint32_t v = C.rxy_l[i < C.rcap ? i : 0];   // always an LDS read
    if (i >= C.rcap) v = C.rxy_g[i];            // rare: regions longer than the LDS list
    return v;
}
__device__ __forceinline__ void rxy_put(RegCtx &C, int i, uint32_t v)
{
    if (i < C.rcap) C.rxy_l[i] = v;
    else C.rxy_g[i] = v;
}

__device__ __forceinline__ double shfl_d(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}
// broadcast from a wave-uniform lane (v_readlane: no LDS crossbar round trip)
__device__ __forceinline__ double readlane_d(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}


// Wave-wide min / max by DPP (round 5): row_shr 1, 2, 4, 8 inside the rows of 16 lanes, row_bcast:15 and row_bcast:31 across them -- lane 63 ends up with the
// value over all 64 lanes and is broadcast with v_readlane.  __shfl_xor butterflies compile to ds_bpermute_b32 (~60 cycles each, six dependent levels: the four
// double extents of region2rect cost a lone band wave ~800 cycles per call, the bounding box of a seed's record ~600); min and max are order-free, so the result
// is the same bit for bit.  A lane without a source in a step keeps its own value (old = the value itself: op(v, v) = v).
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_mov_d(double v)
{
    return __hiloint2double(dpp_mov_i<CTRL, ROWMASK>(__double2hiint(v)), dpp_mov_i<CTRL, ROWMASK>(__double2loint(v)));
}
#define PLF_DPP_REDUCE(v, OP, MOV)                                                                            \
    v = OP(v, MOV<0x111, 0xf>(v)); v = OP(v, MOV<0x112, 0xf>(v)); v = OP(v, MOV<0x114, 0xf>(v)); v = OP(v, MOV<0x118, 0xf>(v)); \
    v = OP(v, MOV<0x142, 0xa>(v)); v = OP(v, MOV<0x143, 0xc>(v));
__device__ __forceinline__ int wave_min_i(int v) { PLF_DPP_REDUCE(v, min, dpp_mov_i) return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ int wave_max_i(int v) { PLF_DPP_REDUCE(v, max, dpp_mov_i) return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ double wave_min_d(double v) { PLF_DPP_REDUCE(v, fmin, dpp_mov_d) return readlane_d(v, 63); }
__device__ __forceinline__ double wave_max_d(double v) { PLF_DPP_REDUCE(v, fmax, dpp_mov_d) return readlane_d(v, 63); }

// Thresholds of the cheap alignment pre-test of region_grow (see there): t1 <= tan(prec - delta), t2 >= tan(prec + delta), delta = 0.05 degrees.
// The pre-test only sorts candidates into "surely aligned", "surely not" and "border" (decided by the reference's own test), so ANY t1 below and
// t2 above those tangents is sound: single-precision tanf with a 1e-4 relative safety factor (its error is ~1e-7; the band is ~2.5e-3 wide) keeps
// the double-precision tan -- a double-double routine that alone costs ~40 VGPRs -- out of this kernel.
// NaN thresholds switch the pre-test off (every decision is then taken by the exact test): both comparisons of the classification are false for a NaN, which
// leaves no lane "surely aligned" and none "surely not" -- without a test of its own in the accept loop.
struct GrowTh { float t1, t2; };
__device__ __forceinline__ GrowTh grow_thresholds(double prec)
{
    const double delta = 8.7266462599716e-4;
    GrowTh t;
    t.t1 = __uint_as_float(0x7FC00000u); t.t2 = t.t1;
    if (prec - delta > 0.0 && prec + delta < 1.55) {
        t.t1 = tanf((float)(prec - delta)) * (1.0f - 1.0e-4f);
        t.t2 = tanf((float)(prec + delta)) * (1.0f + 1.0e-4f);
    }
    return t;
}

// 3x3 neighbourhood data of up to 7 queued region points: lane = slot * 9 + k9, neighbours in (yy, xx) order
struct Grp { uint32_t w; uint32_t fl; double csx, csy; int a; uint32_t xy; };   // w: angle word, fl: USED flag of the bitmap mode (candidate iff w < 0x80000000, fl == 0 and the lane is valid)
// Round 5: in the bitmap mode (band waves, validation rounds) the flag stays a value of its own.  ang_load() merges it into the sign of the angle word, i.e. the
// merged word depends on the GLOBAL load of the angle and the compiler placed `s_waitcnt vmcnt(1)` right behind the prefetch of the next group -- every group
// of a lone band wave waited a full L2 round trip for a word it needs one group later.  With the flag apart nothing reads the prefetched registers before the hand-over.
// The cos/sin increment is fetched together with the angle word (fetching it only for candidates, after the word has arrived, saves HBM traffic but puts a
// second dependent round trip -- and an s_waitcnt that also stalls the group being processed -- into every step of the chain: 77.2 -> 71.4 ms per 4096
// frames).
// Round 4: EVERY lane loads (from a clamped address) and the lanes that hold nothing come back as a wave-wide mask (`valid`, scalar registers), instead of three
// nested EXEC regions that each re-materialised five default values: ~35 vector + ~20 scalar instructions per group became ~15 + 6.  One test covers the image
// border: a region point is a DEFINED pixel, and ll_angle leaves the last row and the last column NOTDEF, so x <= W-2 and y <= H-2; x-1 = -1 addresses the last
// column of the row above (NOTDEF: never a candidate), and only y-1 = -1 leaves the frame -- with a negative index.
__device__ __forceinline__ Grp group_at(const RegCtx &C, uint32_t pxy, int kx, int ky, int nlanes, unsigned long long &valid)
{
    Grp G;
    const int xx = (int)(pxy & 0xFFFF) + kx, yy = (int)(pxy >> 16) + ky;
    G.a = yy * C.W + xx;
    G.xy = (uint32_t)xx | ((uint32_t)yy << 16);
    valid = __ballot(G.a >= 0);
    // lanes past the group's last point hold whatever their list slot holds: they read pixel 0 (one cached line for all of them) -- with the address of that stale
    // slot they pulled 25 GB of unrelated lines per 8192-frame launch through the L2
    // (no upper clamp: a region point is a defined pixel, x <= W-2 and y <= H-2, so its neighbours end at the last pixel of the frame)
    const int af = plf_lane() < nlanes ? max(G.a, 0) : 0;
    const double2 c = C.cs[af];
    if (C.use_bm) { G.w = C.ang[af]; G.fl = (C.bm[af >> 5] >> (af & 31)) & 1u; }
    else { G.w = ang_load(C, af); G.fl = 0u; }
    G.csx = c.x; G.csy = c.y;
    return G;
}
__device__ __forceinline__ Grp load_group(const RegCtx &C, int first, int cnt, int slot, int kx, int ky, unsigned long long &valid)
{
    const int idx = first + slot;
    uint32_t pxy = C.rxy_l[min(idx, C.rcap)];      // (the word at rcap is the spare mailbox word: readable)
    Grp G;
    if (first + cnt > C.rcap) {                    // rare, wave-uniform: part of the group lives in the global part of the list.  A branch of its own, with the
        asm volatile("");                          // dependent load and its wait inside: merged into the common path it would put an s_waitcnt vmcnt(0) -- i.e. a
        if (slot < cnt && idx >= C.rcap) pxy = C.rxy_g[idx];   // wait for the CURRENT group's data -- in front of every prefetch
        G = group_at(C, pxy, kx, ky, 9 * cnt, valid);
    } else G = group_at(C, pxy, kx, ky, 9 * cnt, valid);
    valid &= (1ull << (9 * cnt)) - 1ull;           // cnt <= 7: lane 63 never
    return G;
}

// LineSegmentDetectorImpl::region_grow.  All lanes return the same (n, reg_angle).
// The accept steps happen strictly in the reference order (centre by centre, neighbours in (yy, xx) order, the
// region angle a function of the sums after every accepted pixel).  Up to 7 queued centres are handled as one
// group of 63 lanes whose lane order IS the reference's test order; the data of the next group (angle + cos/sin
// increment per neighbour) is loaded while the current group is processed.
//
// The reference recomputes reg_angle = fastAtan2(sumdy, sumdx) after every accept and tests every later neighbour
// against it.  Here the fastAtan2 + exact double test is only evaluated when it can matter: with S = (sumdx, sumdy)
// and u = (cos a, sin a) of a candidate, the angle between S and u is atan2(|S x u|, S . u); fastAtan2 differs from
// the true angle of S by < 0.0096 degrees, so a candidate whose angle to S is below prec - 0.05 deg is aligned and
// one above prec + 0.05 deg is not, whatever the exact test would compute.  Only candidates inside that 0.1 degree
// band ("border") are decided by the exact test.  The USED flags arrive with the angle words of a group, i.e. they
// are as old as the group's load: a pixel accepted since then is removed from the later lanes of the current group
// and from the lanes of the already-loaded next group by comparing addresses.
// Round 4 (the step is bound by instruction issue, vector and scalar alike): two tests of the classification went -- `S . u > 0` is implied by both cone
// comparisons (|S x u| >= 0 and t1, t2 > 0), and |S|^2 >= 0.25 always holds: S starts as a unit vector and every accepted lane has S . u > 0 (it passed the
// cone test, or it was a border lane, and border lanes pass `|S x u| < t2 * S . u` as well), so |S| only grows; and the pixels a region takes out of the seed
// chunk are no longer tracked per accept (seven scalar instructions each) but read off the list once, when a small region ends (regions_body).
template <int PF>
__device__ int region_grow(RegCtx &C, int sx, int sy, float deg0, float2 cs0, double prec, GrowTh th, double &reg_angle_out, int min_n)
{
    const int lane = plf_lane();
    double reg_angle = (double)deg0 * DEG2RAD_D;
    int n_theta = 1;                      // reg_angle is the value the reference holds for the sums of the first n_theta pixels (scalar; only the rare border path and the end touch it)
    float sumdx = cs0.x, sumdy = cs0.y;   // float(cos(reg_angle)), float(sin(reg_angle))
    const uint32_t sxy = (uint32_t)sx | ((uint32_t)sy << 16);
    if (lane == 0) {
        rxy_put(C, 0, sxy);
        used_set(C, sy * C.W + sx, __float_as_uint(deg0));
    }
    CBAR();
    int n = 1, i = 0;
    const int slot = lane / 9, k9 = lane - slot * 9;
    const int kx = k9 % 3 - 1, ky = k9 / 3 - 1;
    // (the first group is the seed itself: its neighbourhood is addressed from (sx, sy) directly, not through the list entry lane 0 has just written)
    // The group being processed (`cur`) is only ever written by the hand-over copy at the top of the loop, never by a load: with the seed's group loaded straight
    // into `cur` in front of the loop (rounds 1-3) the compiler had to place the wait for THOSE loads at the first use of `cur` inside the loop -- behind the
    // prefetch of the next group -- and that one static s_waitcnt vmcnt(0) made every iteration wait for the prefetch it had just issued.
    int nx_n = 1, cur_n;
    unsigned long long nx_valid, nx_stale = 0ull;   // nx_stale: lanes whose pixel was accepted after its word was loaded
    Grp nx = group_at(C, sxy, kx, ky, 9, nx_valid);
    nx_valid &= 0x1FFull;
    // All per-lane predicates of the accept loop are kept as wave-uniform 64-bit masks (the compares write them
    // directly), so the loop control is scalar and nothing bounces between VGPR booleans and masks.
    for (;;) {
        CNT(7, 1);
        // ---- hand-over: the group fetched ahead becomes the current one (the wait for its loads lands here)
        const Grp cur = nx;
        const unsigned long long cur_valid = nx_valid, cur_stale = nx_stale;
        cur_n = nx_n;
#ifdef PLF_LSD_TIMING
        { TIC(tw); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); TOC(20, tw); TOCB(22, tw); }
#endif
        // (everything that reads what the current group's loads brought is computed HERE, in front of the prefetch, with a compiler barrier behind it: the wait for
        // those loads must not end up behind the loads of the next group -- vmcnt counts in order, so it would wait for them as well)
        unsigned long long candm = __ballot(cur.w < 0x80000000u && cur.fl == 0u) & cur_valid & ~cur_stale;
        const float ux = (float)cur.csx, uy = (float)cur.csy;
        asm volatile("" : : "v"(ux), "v"(uy), "v"(cur.w) : "memory");
        // ---- issue the loads of the next group: list entries that exist now
        // PF == 0 (k_lsd_regions2: eight chains per SIMD): NO fetch ahead -- the next group is loaded when this one is done and then holds every centre accepted
        // by now, i.e. fewer, fuller groups (a group costs ~95 instructions whatever it holds; 2.7 centres per group with the fetch ahead).  The exposed round
        // trip costs nothing there: the kernel is bound by instruction issue and seven other chains fill the wait.  73.7 -> 69.2 ms per 8192 frames.
        nx_n = PF ? min(7, n - (i + cur_n)) : 0;
        if (nx_n < 0) nx_n = 0;
        nx.w = 0xFFFFFFFFu; nx.fl = 0u; nx.csx = 0.0; nx.csy = 0.0; nx.a = -1; nx.xy = 0u;
        nx_valid = 0ull; nx_stale = 0ull;
        if (nx_n > 0) nx = load_group(C, i + cur_n, nx_n, slot, kx, ky, nx_valid);
        // ---- process the current group
        CNT(8, cur_n);
        TIC(tacc);
        // The accepted lanes of the group are collected as a mask and their USED flags and list entries written ONCE, behind the loop (they are accepted in
        // ascending lane order, so a lane's list position is the count before the group + its rank in the mask): a one-lane EXEC region with two stores per
        // accept -- 15 instructions of the 125 an accept costs -- became one scalar OR.  Nothing reads them earlier: the next group was fetched before this loop
        // (its lanes are struck out by address), the one after is fetched at the top of the next trip.
        unsigned long long accm = 0ull;
        const int n0 = n;
        // classification of the remaining candidates against the current sums: mA surely aligned, mB border.  The first remaining candidate that is not surely
        // misaligned is next; border lanes are rare (the cone pre-test decides ~99.8 % of the candidates), so the loop is flat: one rarely taken branch for the
        // reference's own test instead of an inner loop over the masks.  A candidate the exact test rejects is decided for good -- it precedes every lane that
        // can still be accepted, and a later accept drops the lanes before it anyway.  (Rotated loop: one scalar exit test per trip, at the bottom.)
        unsigned long long mB, mAB;
#ifndef PLF_GROW_PK_MUL   // the four products as plain v_mul_f32, written as assembly: the compiler pairs them into two v_pk_mul_f32, and packed fp32 forms issue at the slow
                          // rate and lengthen the trip's dependent chain (k_lsd_regions2 66.7 -> 65.5 ms per 8192 frames; -DPLF_GROW_PK_MUL restores the C form)
#define PLF_GROW_PRODUCTS() float p0_, p1_, p2_, p3_;                                                                                        \
            asm("v_mul_f32 %0, %1, %2" : "=v"(p0_) : "v"(sumdx), "v"(ux)); asm("v_mul_f32 %0, %1, %2" : "=v"(p1_) : "v"(sumdy), "v"(uy));    \
            asm("v_mul_f32 %0, %1, %2" : "=v"(p2_) : "v"(sumdx), "v"(uy)); asm("v_mul_f32 %0, %1, %2" : "=v"(p3_) : "v"(sumdy), "v"(ux));    \
            const float dot = p0_ + p1_, acr = fabsf(p2_ - p3_);
#else
#define PLF_GROW_PRODUCTS() const float dot = sumdx * ux + sumdy * uy, acr = fabsf(sumdx * uy - sumdy * ux);
#endif
#define PLF_GROW_CLASSIFY()                                                                                   \
        {                                                                                                     \
            PLF_GROW_PRODUCTS()                                                                               \
            const unsigned long long mA = candm & __ballot(acr <= th.t1 * dot);                               \
            mB = candm & ~mA & ~__ballot(acr >= th.t2 * dot);                                                 \
            mAB = mA | mB;                                                                                    \
        }
        PLF_GROW_CLASSIFY()
        if (mAB) do {
            const int k = __ffsll((long long)mAB) - 1;
            if ((mB >> k) & 1ull) {
                if (n_theta != n) {
                    // (the empty asm pins the fastAtan2 -- two IEEE divisions -- inside this rarely taken branch; the
                    // compiler would otherwise evaluate it speculatively on every accept step)
                    float fx = sumdx, fy = sumdy;
                    asm volatile("" : "+v"(fx), "+v"(fy));
                    reg_angle = (double)plf_fast_atan2(fy, fx) * DEG2RAD_D;
                    n_theta = n;
                }
                bool al = false;
                if (lane == k) {
                    double n_th = reg_angle - (double)__uint_as_float(cur.w) * DEG2RAD_D;
                    if (n_th < 0) n_th = -n_th;
                    if (n_th > M_3_2_PI_D) {
                        n_th -= M_2__PI_D;
                        if (n_th < 0) n_th = -n_th;
                    }
                    al = n_th <= prec;
                }
                if (!__ballot(al)) { candm &= ~(1ull << k); PLF_GROW_CLASSIFY() continue; }   // not aligned: the sums did not change
            }
            CNT(9, 1);
            accm |= 1ull << k;
            const double cc = readlane_d(cur.csx, k), ss = readlane_d(cur.csy, k);
            const int ka = __builtin_amdgcn_readlane(cur.a, k);
            // `sumdx += cos(float(angle))`: ::cos(double) of the float-rounded angle, float accumulator
            sumdx = (float)((double)sumdx + cc);
            sumdy = (float)((double)sumdy + ss);
            ++n;
            candm &= ~((2ull << k) - 1ull) & ~__ballot(cur.a == ka);   // lanes up to k are decided; the pixel is taken
            if (PF) nx_stale |= __ballot(nx.a == ka);
            PLF_GROW_CLASSIFY()
        } while (mAB);
#undef PLF_GROW_CLASSIFY
#undef PLF_GROW_PRODUCTS
        if (accm) {
            if (__builtin_amdgcn_inverse_ballot_w64(accm)) {
                used_set(C, cur.a, cur.w);
                rxy_put(C, n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(accm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)accm, 0u)), cur.xy);
            }
        }
        CBAR();
        TOC(19, tacc); TOCB(21, tacc);
        i += cur_n;
        if (nx_n == 0) {
            if (i >= n) break;
            // nothing could be loaded ahead (short list): load the next group now
            nx_n = min(7, n - i);
            nx = load_group(C, i, nx_n, slot, kx, ky, nx_valid);
            nx_stale = 0ull;
        }
    }
    // (round 5: the final angle only for a region somebody looks at -- min_n = the caller's minimum region size.  Two regions in three stay below it, and the
    // fastAtan2 with its two IEEE divisions was ~50 instructions of every one of them)
    if (n_theta != n && n >= min_n) reg_angle = (double)plf_fast_atan2(sumdy, sumdx) * DEG2RAD_D;
    reg_angle_out = reg_angle;
    return n;
}

__device__ __forceinline__ double dist_d(double x1, double y1, double x2, double y2) { return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }
__device__ __forceinline__ double distsq_d(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }

__device__ __forceinline__ double angle_diff_signed_d(double a, double b)
{
    double diff = a - b;
    while (diff <= -PI_D) diff += M_2__PI_D;
    while (diff > PI_D) diff -= M_2__PI_D;
    return diff;
}

// region2rect incl. get_theta.  The per-point products are order-free and computed by all lanes (64 points at a
// time, modgrad gathered by coordinate); only the running sums are accumulated serially in list order, exactly the
// reference's additions.  Extents (min/max) are order-free and reduced across the wave.
template <int STG>
__device__ void region2rect(RegCtx &C, int n, double reg_angle, double prec, double p, LsdRect &rec)
{
    const int lane = plf_lane();
    double x = 0, y = 0, sum = 0;
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    if (STG || C.stg) {   // (STG: known at compile time -- the kernel then holds no copy of the v_readlane path)
        // Round 4: the running sums are the reference's additions in list order, but ONE SUM PER LANE: 32 lanes compute the products of 32 points (order-free)
        // and stage them in LDS, lanes 0-2 read their component of the 32 triples back and add them one after the other -- 2 instructions per point for all three
        // sums instead of 9 (two v_readlane and an add per sum).  Points past the end add +0.0, which leaves a sum unchanged bit for bit (it is never -0.0).
        const int Lc = min(lane, 2);
        double acc = 0.0;
        for (int base = 0; base < n; base += 32) {
            const int i = base + lane;
            double wx = 0.0, wy = 0.0, w = 0.0;
            if (lane < 32 && i < n) {
                const uint32_t q = rxy_get(C, i);
                w = C.modgrad[(int)(q >> 16) * C.W + (int)(q & 0xFFFF)];
                wx = (double)(int)(q & 0xFFFF) * w;
                wy = (double)(int)(q >> 16) * w;
            }
            if (lane < 32) { C.stg[lane * 3 + 0] = wx; C.stg[lane * 3 + 1] = wy; C.stg[lane * 3 + 2] = w; }
            CBAR();
#pragma unroll
            for (int q = 0; q < 32; q++) acc += C.stg[q * 3 + Lc];
            CBAR();
        }
        x = readlane_d(acc, 0); y = readlane_d(acc, 1); sum = readlane_d(acc, 2);
        x /= sum;
        y /= sum;
        acc = 0.0;
        for (int base = 0; base < n; base += 32) {
            const int i = base + lane;
            double txx = 0.0, tyy = 0.0, txy = 0.0;
            if (lane < 32 && i < n) {
                const uint32_t q = rxy_get(C, i);
                const double w = C.modgrad[(int)(q >> 16) * C.W + (int)(q & 0xFFFF)];
                const double dx = (double)(int)(q & 0xFFFF) - x, dy = (double)(int)(q >> 16) - y;
                txx = dy * dy * w;
                tyy = dx * dx * w;
                txy = -(dx * dy * w);    // (`Ixy -= t` is `Ixy += -t`, exactly)
                if (txy == 0.0) txy = 0.0;   // (-0.0 -> +0.0: keeps "a running sum is never -0.0" true for the padding argument above)
            }
            if (lane < 32) { C.stg[lane * 3 + 0] = txx; C.stg[lane * 3 + 1] = tyy; C.stg[lane * 3 + 2] = txy; }
            CBAR();
#pragma unroll
            for (int q = 0; q < 32; q++) acc += C.stg[q * 3 + Lc];
            CBAR();
        }
        Ixx = readlane_d(acc, 0); Iyy = readlane_d(acc, 1); Ixy = readlane_d(acc, 2);
    } else {
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double wx = 0.0, wy = 0.0, w = 0.0;
        if (i < n) {
            const uint32_t q = rxy_get(C, i);
            w = C.modgrad[(int)(q >> 16) * C.W + (int)(q & 0xFFFF)];
            wx = (double)(int)(q & 0xFFFF) * w;
            wy = (double)(int)(q >> 16) * w;
        }
        // (in list order, 8 points per loop trip; the lanes past the end hold +0.0, which leaves these non-negative sums unchanged bit for bit)
        const int cnt = min(64, n - base);
        for (int k0 = 0; k0 < cnt; k0 += 8) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                x += readlane_d(wx, k0 + q);
                y += readlane_d(wy, k0 + q);
                sum += readlane_d(w, k0 + q);
            }
        }
    }
    x /= sum;
    y /= sum;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double txx = 0.0, tyy = 0.0, txy = 0.0;
        if (i < n) {
            const uint32_t q = rxy_get(C, i);
            const double w = C.modgrad[(int)(q >> 16) * C.W + (int)(q & 0xFFFF)];
            const double dx = (double)(int)(q & 0xFFFF) - x, dy = (double)(int)(q >> 16) - y;
            txx = dy * dy * w;
            tyy = dx * dx * w;
            txy = dx * dy * w;
        }
        const int cnt = min(64, n - base);
        for (int k0 = 0; k0 < cnt; k0 += 8) {   // (Ixy - (+0.0) = Ixy as well: a running sum is never -0.0)
#pragma unroll
            for (int q = 0; q < 8; q++) {
                Ixx += readlane_d(txx, k0 + q);
                Iyy += readlane_d(tyy, k0 + q);
                Ixy -= readlane_d(txy, k0 + q);
            }
        }
    }
    }
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)plf_fast_atan2((float)(lambda - Ixx), (float)Ixy)
                                           : (double)plf_fast_atan2((float)Ixy, (float)(lambda - Iyy));
    theta *= DEG2RAD_D;
    {
        double d = angle_diff_signed_d(theta, reg_angle);
        if (d < 0) d = -d;
        if (d > prec) theta += PI_D;
    }
    const double dx = cos(theta), dy = sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = lane; i < n; i += 64) {
        const uint32_t q = rxy_get(C, i);
        const double regdx = (double)(int)(q & 0xFFFF) - x, regdy = (double)(int)(q >> 16) - y;
        const double l = regdx * dx + regdy * dy;
        const double w = -regdx * dy + regdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l);
        w_max = fmax(w_max, w); w_min = fmin(w_min, w);
    }
    if (STG == 0) {   // lone band waves: the DPP reduction is ~450 cycles shorter than six dependent ds_bpermute levels
        l_max = wave_max_d(l_max); l_min = wave_min_d(l_min); w_max = wave_max_d(w_max); w_min = wave_min_d(w_min);
    } else {          // issue-bound large-batch kernel: the butterfly runs on the LDS pipe; the DPP form adds vector instructions (68.2 -> 70.1 ms per 8192 frames)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l_max = fmax(l_max, shfl_d(l_max, lane ^ o)); l_min = fmin(l_min, shfl_d(l_min, lane ^ o));
            w_max = fmax(w_max, shfl_d(w_max, lane ^ o)); w_min = fmin(w_min, shfl_d(w_min, lane ^ o));
        }
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

// reduce_region_radius: the swap-with-last removal is replayed literally (it permutes the list, and the
// order of the list decides the rounding of the next region2rect sums).
template <int STG>
__device__ bool reduce_region_radius(RegCtx &C, int &n, double reg_angle, double prec, double p, LsdRect &rec, double density,
                                     double density_th)
{
    const int lane = plf_lane();
    const uint32_t q0 = rxy_get(C, 0);
    const double xc = (double)(int)(q0 & 0xFFFF), yc = (double)(int)(q0 >> 16);
    const double radSq1 = distsq_d(xc, yc, rec.x1, rec.y1), radSq2 = distsq_d(xc, yc, rec.x2, rec.y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
        if (C.t_dead && wall_clock64() > C.t_dead) return false;   // budgeted frames only: a pathological region can spend seconds in this loop
        radSq *= 0.75 * 0.75;
        const int m = n;
        if (m + (m >> 1) + 64 <= C.gcap) {
            // The reference removes a point by swapping the LAST point into its place and re-examining that slot.  The kept prefix that leaves
            // behind is: every kept point below the new size stays where it is, and the removed ones there ("holes", ascending) are filled with the
            // kept points from above the new size taken from the END downwards (checked against the literal loop on random cases).  All lanes work.
            int m_new = 0;
            for (int base = 0; base < m; base += 64) {
                const int i = base + lane;
                bool in = false;
                if (i < m) {
                    const uint32_t q = rxy_get(C, i);
                    in = !(distsq_d(xc, yc, (double)(int)(q & 0xFFFF), (double)(int)(q >> 16)) > radSq);
                    if (!in) used_clr(C, (int)(q >> 16) * C.W + (int)(q & 0xFFFF));
                }
                m_new += __popcll(__ballot(in));
            }
            uint32_t *tmp = C.rxy_g + m;   // hole positions (the list never reaches beyond m)
            int hcount = 0;
            for (int base = 0; base < m_new; base += 64) {
                const int i = base + lane;
                bool out = false;
                if (i < m_new) {
                    const uint32_t q = rxy_get(C, i);
                    out = distsq_d(xc, yc, (double)(int)(q & 0xFFFF), (double)(int)(q >> 16)) > radSq;
                }
                const unsigned long long om = __ballot(out);
                if (out) __hip_atomic_store(&tmp[hcount + __popcll(om & ((1ull << lane) - 1ull))], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                hcount += __popcll(om);
            }
            CBAR();
            int fcount = 0;
            for (int base = ((m - 1) >> 6) << 6; base >= 0 && base + 64 > m_new && fcount < hcount; base -= 64) {
                const int i = base + lane;
                bool in = false;
                uint32_t q = 0u;
                if (i < m && i >= m_new) {
                    q = rxy_get(C, i);
                    in = !(distsq_d(xc, yc, (double)(int)(q & 0xFFFF), (double)(int)(q >> 16)) > radSq);
                }
                const unsigned long long im = __ballot(in);
                if (in) {
                    const int k = fcount + __popcll(im & ~((2ull << lane) - 1ull));   // kept points with a higher index come first
                    // (round 5: a SWAP, as in the reference's loop -- the removed point goes to the slot the kept one leaves.  Until then the hole was simply
                    // overwritten: the prefix [0, n) was right, but the list as a whole was no longer a permutation of what the regrowth had accepted, and the
                    // few-frames schedule logs exactly that list AFTER refine() as "every pixel the seed ever accepted": a pixel accepted by the regrowth and
                    // released here was missing from the record, i.e. its neighbourhood was not covered by the validity test of the validation rounds.)
                    const int hole = (int)__hip_atomic_load(&tmp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t hq = rxy_get(C, hole);
                    rxy_put(C, hole, q);
                    rxy_put(C, i, hq);
                }
                fcount += __popcll(im);
            }
            CBAR();
            n = m_new;
        } else {
        if (lane == 0) {
            int mm = n;
            for (int i = 0; i < mm; ++i) {
                const uint32_t q = rxy_get(C, i);
                if (distsq_d(xc, yc, (double)(int)(q & 0xFFFF), (double)(int)(q >> 16)) > radSq) {
                    used_clr(C, (int)(q >> 16) * C.W + (int)(q & 0xFFFF));
                    const uint32_t ql = rxy_get(C, mm - 1);
                    rxy_put(C, mm - 1, q);
                    rxy_put(C, i, ql);
                    --mm;
                    --i;
                }
            }
            C.rxy_l[C.rcap] = (uint32_t)mm;  // one spare LDS word carries the new size to the other lanes
        }
        CBAR();   // one wave owns the frame: LDS accesses of a wave are performed in order
        n = (int)C.rxy_l[C.rcap];
        CBAR();   // one wave owns the frame: LDS accesses of a wave are performed in order
        }
        if (n < 2) return false;
        region2rect<STG>(C, n, reg_angle, prec, p, rec);
        density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
}

template <int STG>
__device__ bool refine(RegCtx &C, int &n, double reg_angle, double prec, double p, LsdRect &rec, double density_th)
{
    const int lane = plf_lane();
    double density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const uint32_t q0 = rxy_get(C, 0);
    const int a0 = (int)(q0 >> 16) * C.W + (int)(q0 & 0xFFFF);
    const double xc = (double)(int)(q0 & 0xFFFF), yc = (double)(int)(q0 >> 16);
    const float deg_c = ang_value(ang_load(C, a0));
    const double ang_c = (double)deg_c * DEG2RAD_D;
    double sum = 0, s_sum = 0;
    int cnt = 0;
    if (STG || C.stg) {
        // (as in region2rect: the two ordered sums are owned by lanes 0 and 1, fed through LDS; a point outside the radius adds +0.0 -- neither sum is ever -0.0:
        // a difference of two non-negative angles is never -0.0 -- and the count is a popcount)
        const int Lc = min(lane, 1);
        double acc = 0.0;
        for (int base = 0; base < n; base += 48) {
            const int i = base + lane;
            bool inc = false;
            double ang_d = 0.0;
            if (lane < 48 && i < n) {
                const uint32_t q = rxy_get(C, i);
                const int a = (int)(q >> 16) * C.W + (int)(q & 0xFFFF);
                used_clr(C, a);
                if (dist_d(xc, yc, (double)(int)(q & 0xFFFF), (double)(int)(q >> 16)) < rec.width) {
                    inc = true;
                    ang_d = angle_diff_signed_d((double)ang_value(ang_load(C, a)) * DEG2RAD_D, ang_c);
                }
            }
            cnt += __popcll(__ballot(inc));
            if (lane < 48) { C.stg[lane * 2 + 0] = ang_d; C.stg[lane * 2 + 1] = ang_d * ang_d; }
            CBAR();
#pragma unroll
            for (int q = 0; q < 48; q++) acc += C.stg[q * 2 + Lc];
            CBAR();
        }
        sum = readlane_d(acc, 0); s_sum = readlane_d(acc, 1);
    } else
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool inc = false;
        double ang_d = 0.0;
        if (i < n) {
            const uint32_t q = rxy_get(C, i);
            const int a = (int)(q >> 16) * C.W + (int)(q & 0xFFFF);
            used_clr(C, a);
            if (dist_d(xc, yc, (double)(int)(q & 0xFFFF), (double)(int)(q >> 16)) < rec.width) {
                inc = true;
                ang_d = angle_diff_signed_d((double)ang_value(ang_load(C, a)) * DEG2RAD_D, ang_c);
            }
        }
        unsigned long long m = __ballot(inc);
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const double d = readlane_d(ang_d, k);
            sum += d;
            s_sum += d * d;
            ++cnt;
        }
    }
    CBAR();   // one wave owns the frame: LDS accesses of a wave are performed in order
    const double mean_angle = sum / (double)cnt;
    const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)cnt + mean_angle * mean_angle);
    n = region_grow<(STG != 2) && (STG != 0 || PLF_SPEC_PF_REFINE)>(C, (int)(q0 & 0xFFFF), (int)(q0 >> 16), deg_c, C.cs0[a0], tau, grow_thresholds(tau), reg_angle, 2);
    C.regrow_n = n;
    if (n < 2) return false;
    region2rect<STG>(C, n, reg_angle, prec, p, rec);
    density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return reduce_region_radius<STG>(C, n, reg_angle, prec, p, rec, density, density_th);
    return true;
}

// The pixels of the 64-pixel seed chunk at C.cbase that the SMALL region just grown (n < 32 list entries, all of them in LDS) has taken: its list entries are
// scattered into 64 flag words parked in the unused part of the list (entries 32..95) and read back as a ballot.  Replaces a mask that region_grow updated with
// seven scalar instructions per accepted pixel.
__device__ __forceinline__ unsigned long long chunk_taken(RegCtx &C, int n, int lane)
{
    C.rxy_l[32 + lane] = 0u;
    CBAR();
    if (lane >= 1 && lane < n) {
        const uint32_t q = C.rxy_l[lane];
        const int d = (int)(q >> 16) * C.W + (int)(q & 0xFFFF) - C.cbase;
        if ((unsigned)d < 64u) C.rxy_l[32 + d] = 1u;
    }
    CBAR();
    const uint32_t v = C.rxy_l[32 + lane];
    CBAR();
    return __ballot(v != 0u);
}

// Seeds that can only grow a ONE-pixel region (round 5).  On natural-image-like frames 44 % of the regions region growing starts are a single pixel (21 % on the
// polygon scenes; tools/singleton_stats.py): the seed's 8 neighbours are undefined, taken, or not aligned with the seed's own level-line angle -- and the first
// step of region_grow tests exactly that, every neighbour against reg_angle = the seed's angle (the sums only move after the first accept).  Such a seed costs the
// whole per-seed path plus one group (~250 instructions) to mark one pixel.  k_lsd_pre therefore leaves one BIT per pixel (LsdGeom::sgl), set iff the pixel has
// an angle and NO neighbour may pass that first test -- none has an angle within the tolerance + 0.001 degrees (single precision: a superset of the reference's
// double-precision test |theta - a| <= prec, wrap at 3/2 pi = the circular difference; the float differences are good to 5e-5 degrees); pixels on the rim of
// the tile k_lsd_pre was looking at never get it (neighbours unknown: 11 % of the pixels).  A seed whose bit is set is a STATIC SINGLE:
// whatever has been marked or released by the time its turn comes, no neighbour can be accepted, its region is the seed alone -- 72 % of the single-pixel regions
// of natural-image-like frames, 46 % on the polygon scenes.  At its turn it sets its flag and nothing else (n = 1 < min_reg_size: no rectangle, no chunk_taken);
// a run of singles in front of the next ordinary seed is marked in one step -- the same flags in the same order as the serial loop.  A single that an earlier
// region takes (a region's mean angle may accept what none of its pixels would) leaves the seed mask the usual way.
// (Measured and dropped: looking at the neighbours' USED flags as well when a chunk is loaded -- eight more loads and ~90 instructions per chunk of 64 seeds
// with mostly 1-3 candidate lanes -- finds the other singles too and costs the polygon scenes more than it saves: k_lsd_regions2 64.5 -> 66.7 ms per 8192
// frames, natural 209 -> 191.5 ms; tools/experiments/README.md.)
// the leading run of singles of the seed mask: the seeds in front of the first one that needs the pipeline
__device__ __forceinline__ unsigned long long singles_run(unsigned long long mask, unsigned long long smask)
{
    const unsigned long long ns = mask & ~smask;
    const unsigned long long below = ns ? ((ns & (0ull - ns)) - 1ull) : ~0ull;
    return mask & smask & below;
}
#ifndef PLF_LSD_SINGLES
#define PLF_LSD_SINGLES 1   // (0: every seed takes the pipeline -- the A/B switch, tools/ab.sh)
#endif

// LDS of one wave of the large-batch region kernel: the first 1280 words of the region list (rcap <= 1279 + the mailbox word) and 1 KB for the parked seed chunk
// (PLF_LSD_WAVE_LIST / PLF_LSD_WAVE_LDS: lsd_geom.h, shared with the host)
#ifndef PLF_REGIONS_PRIO
#define PLF_REGIONS_PRIO 3
#endif
// BUDGET (plf_line_params.max_ms > 0; separate kernel instances, the default ones carry no clock reads): the wave reads the 100 MHz clock before every seed
// and, past the deadline, leaves the frame with the rectangles found so far -- status bit 8 (PLF_W_TRUNCATED), per-frame flag in status[16 + f]
template <int LDSOFF, int FPW, bool BUDGET>
__device__ __forceinline__ void regions_body(float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                             const double2 *__restrict__ cs_all, const float2 *__restrict__ cs0_all,
                                             uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all, int *__restrict__ nrect,
                                             int *__restrict__ status, const LsdGeom &g, const uint32_t *__restrict__ seeds_all, int nframes,
                                             const int *__restrict__ perm = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char smem_base[];
    // LDSOFF < 0 (k_lsd_regions2): blockDim.x / 64 frames per workgroup, one wave each, the LDS offset of a wave is a run-time scalar
    const int wv = LDSOFF < 0 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (LDSOFF ? 1 : 0);
    const int fpw = LDSOFF < 0 ? (int)(blockDim.x >> 6) : FPW;
    int f = FPW == 1 ? (int)blockIdx.x : (int)blockIdx.x * fpw + wv;
    const int lane = FPW == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    // (large batches: which frame this wave slot works on is dealt by k_lsd_balance -- frames of similar chain length share a workgroup, the workgroups of a CU get
    // equal sums; slots past the batch hold -1)
    if (LDSOFF < 0 && perm) f = f < nframes + fpw ? __builtin_amdgcn_readfirstlane(perm[f]) : -1;
    if (FPW != 1 && (f < 0 || f >= nframes)) return;
    LDS_PTR(char) smem = (LDS_PTR(char))smem_base + (LDSOFF < 0 ? wv * PLF_LSD_WAVE_LDS : LDSOFF);
    const uint32_t *seeds = seeds_all ? seeds_all + (size_t)f * g.s_stride : nullptr;   // sorted keys (seed_order 1) or raster
    const int W = g.sw, H = g.sh, NP = W * H;
    RegCtx C;
    C.W = W; C.H = H;
    C.ang = reinterpret_cast<uint32_t *>(ang_all) + (size_t)f * g.s_stride;
    C.modgrad = modgrad_all + (size_t)f * g.s_stride;
    C.cs = cs_all + (size_t)f * g.s_stride;
    C.cs0 = cs0_all + (size_t)f * g.s_stride;
    C.sgl = g.sgl + (size_t)f * (g.s_stride >> 5);
    C.rxy_l = (LDS_PTR(uint32_t))smem;
    C.rcap = LDSOFF < 0 ? min(g.rcap, PLF_LSD_WAVE_LIST / 4 - 1) : g.rcap;   // (the large-batch kernel keeps a shorter head of the list in LDS)
    C.gcap = (int)g.s_stride;
    C.use_bm = 0; C.bm = (LDS_PTR(uint32_t))smem; C.regrow_n = -1;
    // region2rect's staging: behind the parked seed chunk (large-batch kernel), or behind the list and its mailbox word (one frame per workgroup)
    C.stg = (LDS_PTR(double))(smem + (LDSOFF < 0 ? PLF_LSD_WAVE_LIST + 1024 : (((g.rcap + 1) * 4 + 7) & ~7)));
    C.t_dead = BUDGET ? wall_clock64() + g.budget_ticks : 0ull;
    bool truncated = false;
    C.rxy_g = rxy_all + (size_t)f * g.s_stride;
    // This wave is one long dependent chain; waves of other kernels sharing its SIMD only ever delay it.
    // Raise its issue priority so that co-running throughput kernels (ORB, matchers, NFA) fill the idle slots instead.
    __builtin_amdgcn_s_setprio(PLF_REGIONS_PRIO);
    LsdRect *rects = rects_all + (size_t)f * g.rect_cap;
    int nr = 0;
    const double prec = g.prec, p = g.p;
    const GrowTh th0 = grow_thresholds(prec);
    TIC(tall);
    if (LDSOFF < 0) {
    // Large-batch kernel: the per-lane state of the seed chunk (pixel, angle word, seed sums) is parked in 1 KB of LDS behind the list instead of in
    // 4 VGPRs that live across the whole per-seed pipeline: 96 instead of 99 VGPRs, i.e. 128 instead of 96 registers per SIMD left next to the four region
    // waves -- with the list cut to 1280 entries (6144 bytes of LDS per wave, 64 KB per CU left) a second k_orb_level tile fits beside them.
    // One 16-byte broadcast read per seed replaces four readlanes; the kernel's own time is unchanged (69.5 vs 69.0 ms per 4096 frames), the step
    // goes from 131.7 to 129.3 ms.  (With two-wave workgroups this build lost: a fifth workgroup pair per CU became possible and the 2048 workgroups
    // of a launch were placed unevenly; an 8-wave workgroup cannot be placed a third time on a CU.)
    // (Round 3, end: the kernel is capped at 64 VGPRs and the list head at 768 entries -- PLF_REGIONS_WPE, PLF_LSD_WAVE_LIST -- so that FOUR 8-wave workgroups fit a
    // CU; the parked chunk matters more than before: the four registers would be four more spills.)
    typedef uint32_t park_t __attribute__((ext_vector_type(4)));
    LDS_PTR(park_t) park = (LDS_PTR(park_t))(smem + PLF_LSD_WAVE_LIST);
    // (raster order: the coordinates of a seed follow from those of its chunk's first pixel -- no integer division by W per seed, ~30 instructions on this path)
    const bool walk = !seeds && W >= 64;
    int bx = 0, by = 0;
    for (int base = 0; base < NP; base += 64, bx += 64) {
        if (bx >= W) { bx -= W; by++; }
        unsigned long long mask, smask = 0ull;
        {
            int px = base + lane;
            if (seeds) px = px < NP ? (int)(seeds[px] & 0xFFFFFu) : NP;
            float2 c0 = make_float2(0.f, 0.f);
            if (px < NP) c0 = C.cs0[px];
            // (the singles' bits are fetched WITH the angle word, not behind it: a byte per pixel loaded under `w < 0x80000000` was a second, dependent round trip
            // per chunk -- the kernel is a chain of round trips as much as of instructions: 207 -> 215 ms per 8192 natural frames instead of a gain.  Raster
            // order: the chunk's 64 bits are two words at a wave-uniform address.)
            // (every lane loads the word that holds its own bit -- the same two words for the whole chunk in raster order -- and the bits are taken out only
            // after the angle words have arrived: pinning the words to scalar registers right behind their load made the compiler wait for them BEFORE it
            // issued the angle-word load, the dependent round trip again)
            uint32_t sgw = 0u;
            if (PLF_LSD_SINGLES && px < NP) sgw = C.sgl[px >> 5];
            const uint32_t w = px < NP ? ang_load(C, px) : 0xFFFFFFFFu;
            mask = __ballot(w < 0x80000000u);
            if (PLF_LSD_SINGLES) smask = __ballot(w < 0x80000000u && ((sgw >> (px & 31)) & 1u));
            park[lane] = park_t{(uint32_t)px, w, __float_as_uint(c0.x), __float_as_uint(c0.y)};
        }
        CBAR();
        C.cbase = seeds ? -0x40000000 : base;
        while (mask) {
#ifndef PLF_NO_UNI_MASK
            // Round 5: the seed mask is pinned to scalar registers.  It is wave-uniform by construction (ballots), but the compiler kept it in a VGPR pair and ran this
            // loop as a divergent one -- v_ffbl, 64-bit vector shifts and EXEC save / restore around every seed: k_lsd_regions2 198 -> 181 ms per 8192
            // natural-image-like frames (38 k seeds per frame), VGPR spills 21 -> 9 (tools/r05_ab_quick.sh; -DPLF_NO_UNI_MASK restores the old code).
            mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mask >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)mask);
#endif
            if (PLF_LSD_SINGLES) {
                // the static singles in front of the next ordinary seed: their flags, nothing else (n = 1).  Every lane of the run reads its own parked entry into
                // registers of its own.  (A variant that took one single per trip, flagged in its parked pixel index and stored from the registers of the broadcast
                // read `park[j]`, made the compiler put `s_waitcnt vmcnt(0)` in front of that read -- the pending flag stores of the previous seed -- for EVERY seed:
                // 207 -> 213 ms per 8192 natural frames instead of a gain.)
                const unsigned long long run = smask ? singles_run(mask, smask) : 0ull;   // (most chunks of a hard-edged scene hold no single: one scalar test per seed)
                if (run) {
                    if (__builtin_amdgcn_inverse_ballot_w64(run)) { const park_t me = park[lane]; used_set(C, (int)me.x, me.y); }
                    CNT(23, __popcll(run));
                    mask &= ~run; smask &= ~run;
                    if (!mask) break;
                }
            }
            if (BUDGET && wall_clock64() > C.t_dead) { truncated = true; break; }
            const int j = __ffsll((long long)mask) - 1;
            const park_t sv = park[j];
            const int seed = __builtin_amdgcn_readfirstlane((int)sv.x);
            const float sdeg = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)sv.y));
            const float2 sc0 = make_float2(__uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)sv.z)),
                                           __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)sv.w)));
            double reg_angle;
            TIC(t0);
            int sx, sy;
            if (walk) { sx = bx + j; sy = by; if (sx >= W) { sx -= W; sy++; } }
            else { sx = seed % W; sy = seed / W; }
            int n = region_grow<(LDSOFF >= 0)>(C, sx, sy, sdeg, sc0, prec, th0, reg_angle, g.min_reg_size);
            TOC(0, t0); CNT(4, 1); CNT(5, n);
            const bool big = n >= g.min_reg_size;
            if (big) {
                LsdRect rec;
                TIC(t1);
                region2rect<(LDSOFF < 0 ? 2 : 1)>(C, n, reg_angle, prec, p, rec);
                TOC(1, t1); CNT(6, 1);
                TIC(t2);
                const bool okr = refine<(LDSOFF < 0 ? 2 : 1)>(C, n, reg_angle, prec, p, rec, 0.7);
                TOC(2, t2);
                if (okr) {
                    if (nr < g.rect_cap) { if (lane == 0) rects[nr] = rec; }
                    else if (lane == 0) atomicOr(status, 1);
                    nr++;
                }
            }
            CBAR();
            mask &= ~((2ull << j) - 1ull);
            if (big || seeds || n >= 32) {   // refine / reduce may have released pixels again: take the flags from memory
                const int px = (int)park[lane].x;
                const uint32_t w = px < NP ? ang_load(C, px) : 0xFFFFFFFFu;
                mask &= __ballot(w < 0x80000000u);
            } else if (n > 1 && mask) {
                mask &= ~chunk_taken(C, n, lane);
            }
        }
        if (BUDGET && truncated) break;
    }
    } else
    for (int base = 0; base < NP; base += 64) {
        int px = base + lane;
        if (seeds) px = px < NP ? (int)(seeds[px] & 0xFFFFFu) : NP;
        // (the seed sums are fetched with the angle words, not after them: one round trip per chunk of 64 seeds instead of two dependent ones)
        float2 c0 = make_float2(0.f, 0.f);
        if (px < NP) c0 = C.cs0[px];
        const bool sgb = PLF_LSD_SINGLES && px < NP && ((C.sgl[px >> 5] >> (px & 31)) & 1u);   // (static single; fetched with the angle word, not behind it)
        uint32_t w = px < NP ? ang_load(C, px) : 0xFFFFFFFFu;
        bool ok = w < 0x80000000u;
        const float deg = __uint_as_float(w);
        C.cbase = seeds ? -0x40000000 : base;   // (list order: the chunk is not contiguous, flags are re-read after every region)
        unsigned long long mask = __ballot(ok), smask = 0ull;
        if (PLF_LSD_SINGLES && mask) smask = __ballot(ok && sgb);
        while (mask) {
            if (PLF_LSD_SINGLES) {
                const unsigned long long run = singles_run(mask, smask);
                if (run) {
                    if (__builtin_amdgcn_inverse_ballot_w64(run)) used_set(C, px, w);
                    ok = ok && !((run >> lane) & 1ull);
                    mask &= ~run;
                    if (!mask) break;
                }
            }
            if (BUDGET && wall_clock64() > C.t_dead) { truncated = true; break; }
            const int j = __ffsll((long long)mask) - 1;
            const int seed = __builtin_amdgcn_readlane(px, j);
            const float sdeg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(deg), j));
            const float2 sc0 = make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0.x), j)),
                                           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0.y), j)));
            double reg_angle;
            TIC(t0);
            int n = region_grow<(LDSOFF >= 0)>(C, seed % W, seed / W, sdeg, sc0, prec, th0, reg_angle, g.min_reg_size);
            TOC(0, t0); CNT(4, 1); CNT(5, n);
            const bool big = n >= g.min_reg_size;
            if (big) {
                LsdRect rec;
                TIC(t1);
                region2rect<(LDSOFF < 0 ? 2 : 1)>(C, n, reg_angle, prec, p, rec);
                TOC(1, t1); CNT(6, 1);
                TIC(t2);
                const bool okr = refine<(LDSOFF < 0 ? 2 : 1)>(C, n, reg_angle, prec, p, rec, 0.7);
                TOC(2, t2);
                if (okr) {
                    if (nr < g.rect_cap) { if (lane == 0) rects[nr] = rec; }
                    else if (lane == 0) atomicOr(status, 1);
                    nr++;
                }
            }
            CBAR();
            if (big || seeds || n >= 32) {   // refine / reduce may have released pixels again: take the flags from memory
                w = px < NP ? ang_load(C, px) : 0xFFFFFFFFu;
                ok = ok && lane > j && w < 0x80000000u;
            } else {
                const unsigned long long tk = n > 1 ? chunk_taken(C, n, lane) : 0ull;
                ok = ok && lane > j && !((tk >> lane) & 1ull);
            }
            mask = __ballot(ok);
        }
        if (BUDGET && truncated) break;
    }
    TOC(3, tall);
    if (lane == 0) nrect[f] = min(nr, g.rect_cap);
    if (BUDGET && truncated && lane == 0) { atomicOr(status, 8); status[16 + f] = 1; }
}

// Which frame goes to which wave of k_lsd_regions2 (round 5).  All waves of the launch are resident at once -- 8 per SIMD: waves v and v + 4 of each of the CU's four
// workgroups (tools/wave_placement.hip reads HW_ID of this launch shape) -- and the launch lasts until the slowest SIMD has worked off its 8 chains.  In batch order
// the sums of the chain lengths of a workgroup differ by 5 % (polygon scenes) to 30 % (natural-image-like frames) from the mean.  Here the frames are sorted by cost
// (defined pixels, counted by k_lsd_pre: exactly the chain length but for the pixels refine releases), cut into fpw tiles of G = B / fpw ranks, and workgroup w
// takes rank w of the even tiles and rank G - 1 - w of the odd ones (serpentine: every workgroup gets about the same sum), tile k and tile fpw - 1 - k on the
// two waves that share a SIMD -- so every SIMD gets about the same sum WHEREVER the dispatcher puts the workgroup (other kernels run beside this one).
// Measured, 8192 VGA frames (tools/balance_probe.py re-orders the batch on the host, tools/balance_dbg.py compares permutations): region kernel 68.2 -> 61.2 ms
// on polygon scenes, 218 -> 198 ms on natural-image-like frames.  Workgroups of ADJACENT ranks dealt to the CUs in serpentine order -- equal sums per CU, like
// chains in a workgroup -- gain nothing as a permutation (215 ms) although the same order laid out in memory by the host runs in 181 ms, and lose 60 % beside the
// ORB kernels of the step, where workgroup w no longer lands on CU w mod 256.
// Rank by counting, one thread per frame over workgroups of 256 threads and 1 KB of LDS: rank(i) = frames with a larger cost (ties: batch order) -- B^2 comparisons,
// ~30 us for 8192 frames -- because the kernel runs beside the ORB tiles and the matchers of the step: a one-block bitonic sort (1024 threads, 64 KB of LDS) waited
// ~40 ms for a CU with that much room (region stage 68 -> 108 ms inside bench.py although it took 0.1 ms alone).  perm[] is pre-set to -1 by the host (slots past
// the batch).  Only which wave does which frame.
__global__ void __launch_bounds__(256) k_lsd_balance(const int *__restrict__ cost, int *__restrict__ perm, int B, int fpw)
{
    __shared__ int sc[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ci = i < B ? cost[i] : -1;
    int rank = 0;
    for (int base = 0; base < B; base += 256) {
        __syncthreads();
        sc[threadIdx.x] = base + (int)threadIdx.x < B ? cost[base + threadIdx.x] : -1;
        __syncthreads();
        const int n = min(256, B - base);
        for (int j = 0; j < n; j++) { const int cj = sc[j]; rank += (cj > ci || (cj == ci && base + j < i)) ? 1 : 0; }
    }
    if (i >= B) return;
    const int G = (B + fpw - 1) / fpw;                       // workgroups; tile k = ranks [k G, (k + 1) G)
    const int k = rank / G, pos = rank - k * G;
    const int w = (k & 1) ? G - 1 - pos : pos;
    // waves v and v + fpw / 2 share a SIMD (fpw = 8: v and v + 4): tiles k and fpw - 1 - k
    const int half = fpw >> 1;
    const int v = (fpw & 1) ? k : (k < half ? k : half + (fpw - 1 - k));
    perm[w * fpw + v] = i;
}

// VGPR cap of the large-batch region kernel = waves per SIMD it is built for (tools/variant_build.sh overrides it).  8: 64 VGPRs, 19 of them spilled (80 bytes of
// scratch per lane) -- the kernel is slower per wave, and eight chains per SIMD instead of four more than make up for it (end of round 3: 8192 frames in
// flight 35.3 k frames/s against 33.1 k with 4096 at 96 VGPRs; at 4096 in flight the two builds are equal, the co-runners get the registers)
#ifndef PLF_REGIONS_WPE
#define PLF_REGIONS_WPE 8
#endif
#if PLF_REGIONS_WPE > 0 && defined(PLF_REGIONS_NSGPR)   // (experiment: an explicit scalar-register budget)
#define PLF_REGIONS_OCC __attribute__((amdgpu_waves_per_eu(PLF_REGIONS_WPE, PLF_REGIONS_WPE), amdgpu_num_sgpr(PLF_REGIONS_NSGPR)))
#elif PLF_REGIONS_WPE > 0
#define PLF_REGIONS_OCC __attribute__((amdgpu_waves_per_eu(PLF_REGIONS_WPE, PLF_REGIONS_WPE)))
#else
#define PLF_REGIONS_OCC
#endif
// Large batches: blockDim.x / 64 frames per workgroup (the host launches 8), one wave each; a wave finds its 6400 bytes of LDS at a run-time
// scalar offset.  (Round 2 had two frames per workgroup with a constant offset per wave, i.e. two copies of the body in one kernel: 97 spilled
// SGPRs instead of 38, 71.8 instead of 69.5 ms per 4096 frames.  Frames per workgroup, whole step at 4096 in flight: 2: 136.4, 4: 135.2,
// 8: 135.0, 16: 136.0 ms.)
__global__ void PLF_REGIONS_OCC __launch_bounds__(1024) k_lsd_regions2(float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                                      const double2 *__restrict__ cs_all, const float2 *__restrict__ cs0_all,
                                                      uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all, int *__restrict__ nrect,
                                                      int *__restrict__ status, LsdGeom g, const uint32_t *__restrict__ seeds_all, int nframes,
                                                      const int *__restrict__ perm)
{
    regions_body<-1, 0, false>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, seeds_all, nframes, perm);
}
__global__ void __launch_bounds__(1024) k_lsd_regions2_budget(float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                                      const double2 *__restrict__ cs_all, const float2 *__restrict__ cs0_all,
                                                      uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all, int *__restrict__ nrect,
                                                      int *__restrict__ status, LsdGeom g, const uint32_t *__restrict__ seeds_all, int nframes,
                                                      const int *__restrict__ perm)
{
    regions_body<-1, 0, true>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, seeds_all, nframes, perm);
}

// Latency mode (a handful of frames in flight, e.g. the live SLAM loop): the chain of one frame is all there is to run, so its memory round
// trips are the run time.  Wave 0 is the region wave (cos/sin fetched eagerly); waves 1..3 of the workgroup only pull the frame's angle and
// increment maps into this XCD's L2 (they were written by k_lsd_pre on all XCDs, i.e. they sit in HBM / Infinity Cache), then leave.
template <bool BUDGET>
__device__ __forceinline__ void regions_lat_body(float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                                         const double2 *__restrict__ cs_all, const float2 *__restrict__ cs0_all,
                                                         uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all, int *__restrict__ nrect,
                                                         int *__restrict__ status, LsdGeom g, const uint32_t *__restrict__ seeds_all, int *__restrict__ sink)
{
    if (threadIdx.x >= 64) {
        const int f = blockIdx.x, t = threadIdx.x - 64, NP = g.sw * g.sh;
        const uint32_t *a = reinterpret_cast<const uint32_t *>(ang_all) + (size_t)f * g.s_stride;
        const uint32_t *c = reinterpret_cast<const uint32_t *>(cs_all + (size_t)f * g.s_stride);
        uint32_t acc = 0;
        // 128-byte lines in raster order, the two maps interleaved band by band (32 rows)
        const int band = 32 * g.sw;
        for (int b0 = 0; b0 < NP; b0 += band) {
            const int b1 = min(NP, b0 + band);
            for (int i = b0 + t * 32; i < b1; i += 192 * 32) acc ^= a[i];
            for (int i = b0 * 4 + t * 32; i < b1 * 4; i += 192 * 32) acc ^= c[i];
        }
        if (acc == 0x9E3779B9u) *sink = (int)acc;   // keeps the loads alive
        return;
    }
    regions_body<0, 1, BUDGET>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, seeds_all, 0);
}

__global__ void __launch_bounds__(256) k_lsd_regions_lat(float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                                         const double2 *__restrict__ cs_all, const float2 *__restrict__ cs0_all,
                                                         uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all, int *__restrict__ nrect,
                                                         int *__restrict__ status, LsdGeom g, const uint32_t *__restrict__ seeds_all, int *__restrict__ sink)
{
    regions_lat_body<false>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, seeds_all, sink);
}
__global__ void __launch_bounds__(256) k_lsd_regions_lat_budget(float *__restrict__ ang_all, const double *__restrict__ modgrad_all,
                                                         const double2 *__restrict__ cs_all, const float2 *__restrict__ cs0_all,
                                                         uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all, int *__restrict__ nrect,
                                                         int *__restrict__ status, LsdGeom g, const uint32_t *__restrict__ seeds_all, int *__restrict__ sink)
{
    regions_lat_body<true>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, seeds_all, sink);
}

// ------------------------------------------------------------------------------------------------
// Banded speculative region growing (few frames in flight).  The serial seed loop is exact but one wave retires it at
// ~11 cycles per instruction; this is the same loop made parallel WITHOUT changing any result (model and proof by
// execution: oracle/lsd_oracle.c, orc_lsd_band_speculation):
//   k_lsd_spec_grow   one wave per (band of rows, frame): the whole per-seed pipeline over the band's seeds against a private USED bitmap in
//                     LDS, started from what growing the rows just above the band (unrecorded "halo") leaves marked.  Every effective
//                     seed leaves a record (seed, every pixel the pipeline ever accepted, which of them are still marked, the rectangle if any).
//   k_lsd_spec_commit one wave per frame walks the bands in order with T (true flags, LDS) and S (the band's speculative flags replayed;
//                     LDS or, for large frames, global memory); a pixel is dirty (D) where the two differ.  A record stands iff its seed is
//                     free in T and the 3x3 dilation of its accepted set holds no dirty pixel -- then every flag it read had the true value
//                     and its marks / rectangle are copied; otherwise the seed is regrown on T.  Seeds speculation skipped but that are
//                     free in T are grown as well.
//   k_lsd_spec_fused  both phases in one launch: the commit wave follows the (staggered) band waves through per-band done flags.
// Overflowing record buffers only disable the records of that frame: the commit kernel then IS the serial loop.
// ------------------------------------------------------------------------------------------------
// (SpecRec / SpecBufs: lsd_geom.h, shared with line_host.hip)

__device__ __forceinline__ unsigned long long spec_bits64(const uint32_t *__restrict__ map, int p, int words);
// rows [reach[0], reach[1]] a band's log can depend on (SpecBufs::reach; fb = frame * nbands + band < slots of the allocation)
__device__ __forceinline__ int *spec_reach(const SpecBufs &SB, size_t fb) { return SB.reach + fb * 2; }


// append the current region list [0, n) as pixel indices
// (mark: OR-ed into every entry -- the band waves log "still marked at the end of the seed" optimistically, see spec_grow_body; bb: per-lane partial bounding box
// x0, y0, x1, y1 of the appended pixels, or nullptr)
struct SpecBB { int x0, y0, x1, y1; };
__device__ __forceinline__ void spec_append(const RegCtx &C, int n, uint32_t *__restrict__ dst, int &tn, int cap, int &ovf, uint32_t mark, SpecBB &bb)
{
    if (tn + n > cap) { ovf = 1; return; }
    for (int i = plf_lane(); i < n; i += 64) {
        const uint32_t q = rxy_get(C, i);
        const int qx = (int)(q & 0xFFFFu), qy = (int)(q >> 16);
        dst[tn + i] = ((uint32_t)qy * (uint32_t)C.W + (uint32_t)qx) | mark;
        bb.x0 = min(bb.x0, qx); bb.y0 = min(bb.y0, qy); bb.x1 = max(bb.x1, qx); bb.y1 = max(bb.y1, qy);
    }
    tn += n;
}

// the per-seed pipeline of the serial loop; the accepted pixels go to dst[t0 ..) (first growth, then the regrowth of refine)
__device__ __forceinline__ bool spec_seed(RegCtx &C, const LsdGeom &g, GrowTh th0, int seed, float sdeg, float2 sc0, LsdRect &rec, uint32_t *__restrict__ dst,
                                          int &tn, int cap, int &ovf, uint32_t mark, SpecBB &bb, bool warm = false)
{
    double reg_angle;
    C.regrow_n = -1;
    TIC(ts0);
#ifndef PLF_SPEC_PF
#define PLF_SPEC_PF 0   // (round 5: no fetch-ahead in the band waves either -- fuller groups; the neighbourhood loads hit the L1: 2-3 % per call, tools/ab_few.sh)
#endif
    int n = region_grow<PLF_SPEC_PF>(C, seed % C.W, seed / C.W, sdeg, sc0, g.prec, th0, reg_angle, g.min_reg_size);
    CBAR();
    TOCB(10, ts0);
    CNTB(16, 1); CNTB(17, n);
    TIC(ts1);
    spec_append(C, n, dst, tn, cap, ovf, mark, bb);
    TOCB(11, ts1);
    if (n < g.min_reg_size) return false;
#ifdef PLF_WARM_NOREFINE
    if (warm) return false;   // (experiment: the unrecorded warm-up regions stand as grown -- no rectangle, no refine; only the quality of the band's guess)
#endif
    TIC(ts2);
    region2rect<0>(C, n, reg_angle, g.prec, g.p, rec);
    TOCB(12, ts2);
    TIC(ts3);
    const bool okr = refine<0>(C, n, reg_angle, g.prec, g.p, rec, 0.7);
    CBAR();
    TOCB(13, ts3);
    if (C.regrow_n >= 0) spec_append(C, C.regrow_n, dst, tn, cap, ovf, mark, bb);   // (reduce_region_radius only permutes that list)
    return okr;
}

__device__ __forceinline__ bool spec_seed(RegCtx &C, const LsdGeom &g, GrowTh th0, int seed, float sdeg, float2 sc0, LsdRect &rec, uint32_t *__restrict__ dst,
                                          int &tn, int cap, int &ovf)
{
    SpecBB bb = {0, 0, 0, 0};
    return spec_seed(C, g, th0, seed, sdeg, sc0, rec, dst, tn, cap, ovf, 0u, bb);
}

__device__ __forceinline__ void spec_ctx(RegCtx &C, const LsdGeom &g, int f, float *ang_all, const double *modgrad_all, const double2 *cs_all, const float2 *cs0_all,
                                         uint32_t *rxy_g, LDS_PTR(uint32_t) list, LDS_PTR(uint32_t) bm)
{
    C.W = g.sw; C.H = g.sh;
    C.ang = reinterpret_cast<uint32_t *>(ang_all) + (size_t)f * g.s_stride;
    C.modgrad = modgrad_all + (size_t)f * g.s_stride;
    C.cs = cs_all + (size_t)f * g.s_stride;
    C.cs0 = cs0_all + (size_t)f * g.s_stride;
    C.sgl = g.sgl + (size_t)f * (g.s_stride >> 5);
    C.rxy_l = list; C.rcap = g.rcap; C.gcap = (int)g.s_stride; C.rxy_g = rxy_g;
    C.use_bm = 1; C.bm = bm; C.regrow_n = -1;
    C.cbase = -0x40000000;
    // region2rect's staging (96 doubles = 192 words) is carved out of the END of the list's LDS words: the list keeps rcap - 192 entries (+ its mailbox word) there
    C.stg = nullptr;
    if (g.rcap >= 192 + 127 && ((g.rcap - 191) & 1) == 0) { C.rcap = g.rcap - 192; C.stg = (LDS_PTR(double))(list + (g.rcap - 191)); }
    C.t_dead = 0ull;
}

// Rows of the bands.  A band wave grows its own rows AND the halo rows above them (unrecorded warm-up), so its run time follows the defined pixels of
// [y0 - halo_rows, y1): the boundaries (any row, since round 3 -- multiples of 8 rows until then, which left bands of 8, 16 or 24 rows next to each other
// when 24 bands share 383 rows: the slowest band wave, i.e. the whole launch, took up to 1.6x the mean) minimise the largest weighted band cost
//     cost(b) = defined pixels in rows [max(0, y0_b - halo_rows), y1_b)   against the weight   w_b = 1 + stagger * b
// (late bands may be longer: the commit wave reaches them later).  The minimum is found by bisection on the cost bound: the 64 lanes of wave 0 try 64 bounds
// at once with a greedy sweep (binary searches in the prefix sums), twice.
__device__ __forceinline__ int spec_band_sweep(const int *__restrict__ P, int rows, int nb, int halo, float stagger, float C, int *__restrict__ by_out)
{
    int a = 0;
    for (int b = 0; b < nb; b++) {
        const int base = b > 0 ? P[max(0, a - halo)] : 0;
        const float lim = C * (1.f + stagger * (float)b) + (float)base;
        const int emax = rows - (nb - 1 - b);          // every later band keeps at least one row
        int lo = a + 1, hi = emax;                     // the band takes at least one row, whatever it costs
        if (b == nb - 1) lo = hi = rows;
        while (lo < hi) {                              // largest e in [lo, hi] with P[e] <= lim (P is non-decreasing); lo if none
            const int mid = (lo + hi + 1) >> 1;
            if ((float)P[mid] <= lim) lo = mid; else hi = mid - 1;
        }
        if (b == nb - 1 && (float)P[rows] > lim) return 0;   // the last band cannot absorb the rest within the bound
        a = lo;
        if (by_out) by_out[b + 1] = a;
    }
    return 1;
}

// (1) per-row counts of the defined pixels + the bitmap of the defined pixels (the angle words are read-only in this mode: the commit / validation waves walk
// the bitmap instead of the angle map).  One wave per 64 pixels, the whole frame in parallel.  rowcnt[f][unit] must be zero on entry (the host clears it).
__global__ void __launch_bounds__(256) k_lsd_spec_rows(const float *__restrict__ ang_all, LsdGeom g, SpecBufs SB, int *__restrict__ rowcnt)
{
    const int f = blockIdx.y, W = g.sw, H = g.sh, rows = H - 1;
    const int ru = (rows + 1023) / 1024;      // rows per counting unit: 1 up to 1024 rows
    const uint32_t *ang = reinterpret_cast<const uint32_t *>(ang_all) + (size_t)f * g.s_stride;
    uint32_t *dm = SB.defmap + (size_t)f * SB.bm_words;
    const int p0 = (blockIdx.x * 256 + (threadIdx.x & ~63));      // first pixel of this wave's 64
    if (p0 >= SB.bm_words * 32) return;
    const int p = p0 + (threadIdx.x & 63);
    const bool def = p < W * H && ang[p] < 0x80000000u;
    const unsigned long long m = __ballot(def);
    if ((threadIdx.x & 63) == 0) {
        dm[p0 >> 5] = (uint32_t)m;
        if ((p0 >> 5) + 1 < SB.bm_words) dm[(p0 >> 5) + 1] = (uint32_t)(m >> 32);
        // the 64 pixels lie in one row or straddle a few (narrow frames): count per row
        unsigned long long rest = m;
        int q = p0;
        while (rest && q < W * rows) {
            const int row = q / W, in_row = min(64 - (q - p0), (row + 1) * W - q);
            const unsigned long long mk = in_row >= 64 ? ~0ull : (((1ull << in_row) - 1ull) << (q - p0));
            const int c = __popcll(rest & mk);
            if (c) atomicAdd(&rowcnt[f * 1024 + min(row / ru, 1023)], c);
            rest &= ~mk;
            q += in_row;
        }
    }
}

// (2) the boundaries: prefix sums of the row counts, then the 64 lanes try 64 cost bounds at once (greedy sweep, binary searches in the prefix sums); the lane
// with the smallest feasible bound has the answer (each lane keeps the boundaries of its own sweep in a scratch row)
__global__ void __launch_bounds__(64) k_lsd_spec_bands(LsdGeom g, SpecBufs SB, const int *__restrict__ rowcnt, int *__restrict__ sweep_scratch)
{
    __shared__ int P[1024 + 1];
    const int f = blockIdx.x, t = threadIdx.x, H = g.sh;
    const int rows = H - 1;
    const int ru = (rows + 1023) / 1024;
    const int units = (rows + ru - 1) / ru;
    // exclusive prefix sums over the units, 16 per lane + a wave scan of the lane totals
    int v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int u = t * 16 + k; v[k] = u < units ? rowcnt[f * 1024 + u] : 0; sum += v[k]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(incl, o, 64); if (t >= o) incl += x; }
    int run = incl - sum;
    if (t == 0) P[0] = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { run += v[k]; P[t * 16 + k + 1] = run; }
    __syncthreads();
    const int nb = SB.nbands, halo_u = (SB.halo_rows + ru - 1) / ru;
    const int total = P[units];
    int *by = SB.band_y + f * (nb + 1);
    int *mine = sweep_scratch + ((size_t)f * 64 + t) * 65;
    // bound per unit weight: between the mean own cost of a band and (generously) three times that plus the heaviest warm-up
    const float wsum = (float)nb + SB.stagger * (float)nb * (float)(nb - 1) * 0.5f;
    const float lo = (float)total / wsum * 0.98f, hi = (float)total / wsum * 3.0f + (float)total / (float)max(units, 1) * (float)(halo_u + 2) + 64.f;
    int ok = 0;
    float C = lo + (hi - lo) * (float)t / 63.f;
    if (t == 63) C = (float)total + 1.f;               // always feasible
    if (units >= nb) ok = spec_band_sweep(P, units, nb, halo_u, SB.stagger, C, mine);
    const unsigned long long m = __ballot(ok != 0);
    if (units >= nb && m) {
        const int win = __ffsll((long long)m) - 1;
        __threadfence();
        const int *src = sweep_scratch + ((size_t)f * 64 + win) * 65;
        for (int b = 1 + t; b <= nb; b += 64) by[b] = min(__hip_atomic_load(&src[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * ru, rows);
    } else {
        for (int b = 1 + t; b <= nb; b += 64) by[b] = (int)((long long)rows * b / nb);   // fewer units than bands (the host excludes it): equal rows
    }
    if (t == 0) { by[0] = 0; }
    __syncthreads();
    if (t == 0) { by[nb] = rows; for (int b = 1; b <= nb; b++) by[b] = max(by[b], by[b - 1]); }
}

template <bool BUDGET>
__device__ __forceinline__ void spec_grow_body(int band, int f, float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                               const float2 *__restrict__ cs0_all, const LsdGeom &g, const SpecBufs &SB)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int W = g.sw, H = g.sh;
    if (threadIdx.x >= 64) {
        // waves 1..3 (when the launch has them): pull the rows this band works on -- warm-up rows, own rows and a margin -- of the four maps into this XCD's L2
        // (k_lsd_pre wrote them from all XCDs), then leave.  The band wave's neighbourhood loads are then L2 hits instead of trips to HBM / Infinity Cache.
        const int t = threadIdx.x - 64;
        const int y0 = SB.band_y[f * (SB.nbands + 1) + band], y1 = SB.band_y[f * (SB.nbands + 1) + band + 1];
        const int ra = max(0, y0 - SB.halo_rows - 12) * W, rb = min(H, y1 + 12) * W;
        const uint32_t *a = reinterpret_cast<const uint32_t *>(ang_all) + (size_t)f * g.s_stride;
        const uint32_t *c = reinterpret_cast<const uint32_t *>(cs_all + (size_t)f * g.s_stride);
        const uint32_t *m = reinterpret_cast<const uint32_t *>(modgrad_all + (size_t)f * g.s_stride);
        const uint32_t *c0 = reinterpret_cast<const uint32_t *>(cs0_all + (size_t)f * g.s_stride);
        uint32_t acc = 0;
        for (int i = ra + t * 32; i < rb; i += 192 * 32) acc ^= a[i];          // one word per 128-byte line
        for (int i = ra * 4 + t * 32; i < rb * 4; i += 192 * 32) acc ^= c[i];
        for (int i = ra * 2 + t * 32; i < rb * 2; i += 192 * 32) acc ^= m[i];
        for (int i = ra * 2 + t * 32; i < rb * 2; i += 192 * 32) acc ^= c0[i];
        if (acc == 0x9E3779B9u) SB.cnt[((size_t)f * SB.nbands + band) * 4 + 3] = (int)acc;   // keeps the loads alive
        return;
    }
    const unsigned long long t_start = wall_clock64();
    LDS_PTR(uint32_t) list = (LDS_PTR(uint32_t))smem;
    LDS_PTR(uint32_t) bm = list + ((g.rcap + 1 + 15) & ~15);
    for (int i = lane; i < SB.bm_words; i += 64) bm[i] = 0u;
    CBAR();
    const size_t fb = (size_t)f * SB.nbands + band;
    RegCtx C;
    spec_ctx(C, g, f, ang_all, modgrad_all, cs_all, cs0_all, SB.rxy + fb * g.s_stride, list, bm);
    if (BUDGET) C.t_dead = wall_clock64() + g.budget_ticks;   // past it the band wave stops; what it has not grown is left to the commit wave, which stops as well
    bool truncated = false;
    __builtin_amdgcn_s_setprio(3);
    uint32_t *tl = SB.tl + fb * SB.tcap;
    SpecRec *recs = SB.recs + fb * SB.rcap_rec;
    uint32_t *seedmap = SB.seedmap + (size_t)f * SB.bm_words;
    const GrowTh th0 = grow_thresholds(g.prec);
    const int y0 = SB.band_y[f * (SB.nbands + 1) + band], y1 = SB.band_y[f * (SB.nbands + 1) + band + 1];
    int nrec = 0, tn = 0, ovf = 0, nrect_band = 0;
    int reach0 = y0, reach1 = max(y1 - 1, y0);   // rows the log depends on: the band's own rows, widened by every record's dilated box
    if (band > 0 && (SB.halo_rows > 0 || SB.fill_rows > 0)) {
        // The warm-up starts from "every defined pixel ABOVE the warm-up rows is taken" (in the serial run they all are when this band's turn comes, bar the
        // few that refine released) instead of an empty map: on an empty map the regions of the top warm-up rows grew upwards without bound -- work that
        // cost more than the warm-up rows themselves (band waves with 5 own + 12 warm-up rows ran 1.6x as long as the 31-row first band).  Like everything the
        // warm-up leaves, this is only the band's GUESS of what the earlier bands mark (its initial S): the commit / the validation rounds compare it with the truth.
        const uint32_t *dm = SB.defmap + (size_t)f * SB.bm_words;
        const int pa = max(0, y0 - (SB.fill_rows > 0 ? 0 : SB.halo_rows)) * W;                  // first pixel of the warm-up rows
        for (int i = lane; i < ((pa + 31) >> 5); i += 64) {
            uint32_t v = dm[i];
            if (i == (pa >> 5) && (pa & 31)) v &= (1u << (pa & 31)) - 1u;
            bm[i] = v;
        }
        CBAR();
    }
    if (band > 0 && SB.fill_rows > 0) {
        // NO warm-up growth (validation rounds only; model: orc_lsd_band_rounds mode 4).  A region that crosses into the band from above was regrown, lower part by lower
        // part, by every band it crosses -- with 48 bands the band waves did ~2.6x the serial work, most of it in the warm-up.  The guess is made without growing
        // anything: a pixel of the band's first fill_rows rows is presumed taken if it hangs on a taken pixel of the row above, or of its own row, through
        // neighbours whose level-line angles differ by at most fill_tol_deg (half the region tolerance) -- a stand-in for "the region from above reaches down to here".
        // Row by row: vertical step per pixel (one lane each), then the closure along the row as a carry chain over 64-bit masks (Kogge-Stone inside a
        // chunk, the carry handed from chunk to chunk), left to right and right to left.  A dozen microseconds per band; what it gets wrong the rounds redo.
        const uint32_t *angw = C.ang;
        const float tol = SB.fill_tol_deg;
        const int r_end = min(y0 + SB.fill_rows, H - 1), nch = (W + 63) >> 6;
        LDS_PTR(unsigned long long) sc = (LDS_PTR(unsigned long long))list;      // per chunk: vertical marks, links to the left neighbour (the region list is idle)
        for (int r = y0; r < r_end; r++) {
            for (int c = 0; c < nch; c++) {
                const int x = c * 64 + lane;
                bool vm = false, lk = false;
                if (x < W) {
                    const int pp = r * W + x;
                    const uint32_t wp = angw[pp];
                    if (wp < 0x80000000u) {
                        const float ap = __uint_as_float(wp);
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            const int xx = x + dx;
                            if (xx < 0 || xx >= W) continue;
                            const int q = pp - W + dx;
                            if (!((bm[q >> 5] >> (q & 31)) & 1u)) continue;
                            const uint32_t wq = angw[q];
                            if (wq >= 0x80000000u) continue;
                            float d = fabsf(ap - __uint_as_float(wq));
                            if (d > 180.f) d = 360.f - d;
                            vm |= d <= tol;
                        }
                        if (x > 0) {
                            const uint32_t wl = angw[pp - 1];
                            if (wl < 0x80000000u) { float d = fabsf(ap - __uint_as_float(wl)); if (d > 180.f) d = 360.f - d; lk = d <= tol; }
                        }
                    }
                }
                const unsigned long long V = __ballot(vm), Lk = __ballot(lk);
                if (lane == 0) { sc[2 * c] = V; sc[2 * c + 1] = Lk; }
            }
            CBAR();
            // closure along the row (wave-uniform 64-bit arithmetic)
            unsigned long long carry = 0ull;
            for (int c = 0; c < nch; c++) {              // left to right: bit i of Lk links pixel i to pixel i - 1
                unsigned long long fl = sc[2 * c], pr = sc[2 * c + 1];
                fl |= carry & pr & 1ull;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) { fl |= pr & (fl << sft); pr &= pr << sft; }
                if (lane == 0) sc[2 * c] = fl;
                carry = fl >> 63;
            }
            CBAR();
            carry = 0ull;
            for (int c = nch - 1; c >= 0; c--) {         // right to left: pixel i hangs on pixel i + 1 through link bit i + 1
                unsigned long long fl = sc[2 * c];
                const unsigned long long lk_here = sc[2 * c + 1], lk_next = c + 1 < nch ? sc[2 * c + 3] : 0ull;
                unsigned long long pr = (lk_here >> 1) | ((lk_next & 1ull) << 63);
                fl |= (carry << 63) & pr;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) { fl |= pr & (fl >> sft); pr &= pr >> sft; }
                carry = fl & 1ull;
                // the chunk's marks into the band's flags: 64 bits at pixel r * W + c * 64, i.e. up to three words
                const int p0 = r * W + c * 64, w0 = p0 >> 5, o = p0 & 31;
                const int npx = min(64, W - c * 64);
                if (npx < 64) fl &= (1ull << npx) - 1ull;
                if (lane < 3) {
                    uint32_t part;
                    if (lane == 0) part = (uint32_t)(fl << o);
                    else if (lane == 1) part = o ? (uint32_t)(fl >> (32 - o)) : (uint32_t)(fl >> 32);
                    else part = o ? (uint32_t)(fl >> (64 - o)) : 0u;
                    const int wi = w0 + lane;
                    if (part && wi < SB.bm_words) __hip_atomic_fetch_or(&bm[wi], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            CBAR();
        }
    }
    // phase 0 (bands > 0): the rows just above the band, unrecorded -- what they mark (regions poking into the band) is the state the band's
    // speculation starts from, handed to the commit wave as the initial S; phase 1: the band itself, recorded
    uint32_t *halo = SB.halo + fb * SB.bm_words;
    for (int phase = (band > 0 && SB.halo_rows > 0) ? 0 : 1; phase < 2; phase++) {
    const bool record = phase == 1;
    TIC(tph);
    // The warm-up regions are clipped below the band: a region that crosses the warm-up rows from above is regrown here from its first pixel in those rows, and
    // without a bound EVERY band a 100-row region crosses grew all of its lower part again -- the 4 warm-up rows cost a band wave more (1.6 ms) than its 8 own
    // rows (1.0 ms).  Only what the warm-up marks inside and just below the band can matter to the band's own seeds; the rows further down are made to look taken
    // while the warm-up runs and are cleared again before the band's own seeds start (nothing real was marked there).  Again only the quality of a guess.
    const int clip_px = (SB.halo_clip >= 0 && y1 + SB.halo_clip < H) ? (y1 + SB.halo_clip) * W : W * H;
    if (!record && clip_px < W * H) {
        for (int i = (clip_px >> 5) + lane; i < SB.bm_words; i += 64) {
            uint32_t v = 0xFFFFFFFFu;
            if (i == (clip_px >> 5) && (clip_px & 31)) v = bm[i] | (0xFFFFFFFFu << (clip_px & 31));
            bm[i] = v;
        }
        CBAR();
    }
    if (record && clip_px < W * H && band > 0 && SB.halo_rows > 0) {
        for (int i = (clip_px >> 5) + lane; i < SB.bm_words; i += 64) {
            uint32_t v = 0u;
            if (i == (clip_px >> 5) && (clip_px & 31)) v = bm[i] & ((1u << (clip_px & 31)) - 1u);
            bm[i] = v;
        }
        CBAR();
    }
    const int ya = record ? y0 : max(0, y0 - SB.halo_rows), yb = record ? y1 : y0;
    if (record) { CBAR(); for (int i = lane; i < SB.bm_words; i += 64) halo[i] = bm[i]; }
    for (int base = ya * W; base < yb * W; base += 64) {
        const int px = base + lane;
        // (round 5: the seed sums and the singles' bits are fetched WITH the angle word, for every lane of the chunk -- under `ok` the sums were a second,
        // dependent round trip per chunk of a lone wave)
        float2 c0 = make_float2(0.f, 0.f);
        if (px < yb * W) c0 = C.cs0[px];
        const unsigned long long sbits = PLF_LSD_SINGLES ? spec_bits64(C.sgl, base, SB.bm_words) : 0ull;
        uint32_t w = px < yb * W ? ang_load(C, px) : 0xFFFFFFFFu;
        bool ok = w < 0x80000000u;
        unsigned long long mask = __ballot(ok), smask = mask & sbits;
        while (mask) {
            if (BUDGET && wall_clock64() > C.t_dead) { truncated = true; break; }
            const int j = __ffsll((long long)mask) - 1;
            const int seed = base + j;
            if (PLF_LSD_SINGLES && ((smask >> j) & 1ull)) {
                // a seed that can only grow a one-pixel region (static single, see singles_run): its flag, and in the recorded phase the record region_grow + spec_append would
                // have left -- one log entry, still marked; bounding box = the pixel; no rectangle
                if (!record) {
                    const unsigned long long run = singles_run(mask, smask);
                    if (__builtin_amdgcn_inverse_ballot_w64(run)) used_set(C, px, w);
                    ok = ok && !((run >> lane) & 1ull);
                    mask &= ~run;
                    continue;
                }
                if (lane == j) used_set(C, px, w);
                if (tn + 1 > SB.tcap || nrec >= SB.rcap_rec) ovf = 1;
                if (!ovf) {
                    if (lane == 0) {
                        const int qy = seed / W, qx = seed - qy * W;
                        tl[tn] = (uint32_t)seed | 0x40000000u;
                        int4 *dst = reinterpret_cast<int4 *>(&recs[nrec]);
                        dst[0] = make_int4(seed, tn, 1, 0);
                        dst[1] = make_int4(max(qx - 1, 0), max(qy - 1, 0), min(qx + 1, W - 1), min(qy + 1, H - 1));
                        atomicOr(&seedmap[seed >> 5], 1u << (seed & 31));
                    }
                    reach0 = min(reach0, max(seed / W - 1, 0)); reach1 = max(reach1, min(seed / W + 1, H - 1));
                    nrec++; tn++;
                }
                CBAR();
                ok = ok && lane > j;
                mask &= ~((2ull << j) - 1ull);
                continue;
            }
            const float sdeg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(__uint_as_float(w)), j));
            const float2 sc0 = make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0.x), j)),
                                           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0.y), j)));
            LsdRect rec;
            const int t0 = tn;
            if (lane == 0 && !SB.out) __hip_atomic_store(&SB.cnt[fb * 4 + 3], seed + 1 + (phase << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // heartbeat (spec_wait_band; nobody waits for a band wave when validation rounds follow)
            // every accepted pixel is logged as "still marked at the end of the seed" right away -- true unless refine released pixels (it regrew the region:
            // C.regrow_n >= 0, ~10 % of the large regions), in which case the entries are re-read and corrected; the bounding box comes out of the append
            // itself.  (Reading every log back to set the flag and find the box cost a global round trip per seed.)
            SpecBB bb = {W, H, -1, -1};
            const bool okr = spec_seed(C, g, th0, seed, sdeg, sc0, rec, tl, tn, SB.tcap, ovf, record ? 0x40000000u : 0u, bb, !record);
            TIC(trec);
            if (!record) { tn = 0; ovf = 0; }
            if (record && nrec >= SB.rcap_rec) ovf = 1;
            if (record && okr) nrect_band++;
            if (record && !ovf) {
                int bx0 = bb.x0, by0 = bb.y0, bx1 = bb.x1, by1 = bb.y1;
                if (C.regrow_n >= 0) {
                    for (int i = t0 + lane; i < tn; i += 64) {
                        const uint32_t q = tl[i] & 0x3FFFFFFFu;
                        if (!((bm[q >> 5] >> (q & 31)) & 1u)) tl[i] = q;
                    }
                }
                bx0 = wave_min_i(bx0); by0 = wave_min_i(by0); bx1 = wave_max_i(bx1); by1 = wave_max_i(by1);
                if (lane == 0) {
                    // (the 32-byte header always, the 96-byte rectangle only when there is one: nothing reads `rec` of a record without -- nine records in ten)
                    int4 *dst = reinterpret_cast<int4 *>(&recs[nrec]);
                    dst[0] = make_int4(seed, t0, tn - t0, okr ? 1 : 0);
                    dst[1] = make_int4(max(bx0 - 1, 0), max(by0 - 1, 0), min(bx1 + 1, W - 1), min(by1 + 1, H - 1));
                    if (okr) recs[nrec].rec = rec;
                    atomicOr(&seedmap[seed >> 5], 1u << (seed & 31));
                }
                reach0 = min(reach0, max(by0 - 1, 0)); reach1 = max(reach1, min(by1 + 1, H - 1));
                nrec++;
            }
            CBAR();
            // flags from the bitmap: cheap, always current (only the LDS word is read again: the angle word itself never changes in this mode, and re-loading it from
            // global memory put an exposed round trip behind every seed)
            ok = ok && lane > j && !((bm[px >> 5] >> (px & 31)) & 1u);
            mask = __ballot(ok);
            TOCB(18, trec);
        }
        if (BUDGET && truncated) break;
    }
    TOCB(14 + phase, tph);
    if (BUDGET && truncated) break;
    }
    if (SB.band_ticks && lane == 0) { SB.band_ticks[fb * 2] = (int)(wall_clock64() - t_start); SB.band_ticks[fb * 2 + 1] = tn; }
    if (SB.out) {   // validation rounds follow: what the band's own records mark (its flags minus the state its warm-up rows left) and its rectangle count
        uint32_t *outb = SB.out + fb * SB.bm_words;
        for (int i = lane; i < SB.bm_words; i += 64) outb[i] = bm[i] & ~halo[i];   // (halo[] = the band's initial state, written when its own seeds started: zeros for band 0)
        if (lane == 0) { SB.nrects[fb] = nrect_band; int *rc = spec_reach(SB, fb); rc[0] = reach0; rc[1] = reach1; }
    }
    // A band wave that ran out of its time budget leaves an INCOMPLETE log: the seeds it did not reach are in neither S nor T, so the commit wave -- a later launch
    // with a clock of its own in the two-launch schedule -- would never see them as candidates and could finish "in time" with regions missing.  Such a
    // log is published like an overflowed one (cnt[2] != 0): the commit wave then grows the band itself, from T, under its own budget -- exact if it finishes,
    // truncated and reported if it does not.
    if (BUDGET && truncated) ovf = 2;
    if (lane == 0) { SB.cnt[fb * 4 + 0] = nrec; SB.cnt[fb * 4 + 1] = tn; SB.cnt[fb * 4 + 2] = ovf; SB.cnt[fb * 4 + 3] = (int)(wall_clock64() & 0x7fffffff); }   // ([3]: 100 MHz timestamp, diagnostics)
    if (ovf && SB.round_state && lane == 0) SB.round_state[f * 4 + 3] = 1;   // an incomplete log cannot be validated: the frame takes the serial commit
    __threadfence();   // every lane's log entries are visible device-wide before the flag
    if (lane == 0) __hip_atomic_store(&SB.done[fb], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) k_lsd_spec_grow(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                      const float2 *__restrict__ cs0_all, LsdGeom g, SpecBufs SB)
{
    spec_grow_body<false>(blockIdx.x, blockIdx.y, ang_all, modgrad_all, cs_all, cs0_all, g, SB);
}
__global__ void __launch_bounds__(256) k_lsd_spec_grow_budget(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                             const float2 *__restrict__ cs0_all, LsdGeom g, SpecBufs SB)
{
    spec_grow_body<true>(blockIdx.x, blockIdx.y, ang_all, modgrad_all, cs_all, cs0_all, g, SB);
}

__device__ __forceinline__ bool bm_get(LDS_PTR(uint32_t) b, int a) { return (b[a >> 5] >> (a & 31)) & 1u; }
__device__ __forceinline__ void bm_set(LDS_PTR(uint32_t) b, int a) { __hip_atomic_fetch_or(&b[a >> 5], 1u << (a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void dc_mark(LDS_PTR(uint32_t) Dc, int a, int W, int ctx)
{
    const int t = ((a / W) >> 3) * ctx + ((a % W) >> 3);
    __hip_atomic_fetch_or(&Dc[t >> 5], 1u << (t & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the band's speculative flags S: in LDS next to T, or (frames whose two bitmaps exceed the LDS) in global memory
// Does a flag that differs between the truth T and the band's speculative state S at a pixel of the 3x3 dilation of a record's accepted set invalidate the record?
// Round 5, the refined rule (model: oracle/lsd_oracle.c, orc_lsd_band_rounds_refined; tests/test_models.py): only if
//   * the pixel is one the record ACCEPTED (centre) and it is truly taken -- an earlier band owns it --, or
//   * the speculation saw the pixel TAKEN and it is truly free: the record skipped a pixel it might have accepted.
// A neighbour the speculation saw FREE and did not accept was rejected for its angle every time it was tested; truly taken, it is skipped instead: the same
// outcome.  (An accepted pixel is free in S when its record is checked: the record found it free, and the marks of the records before it are already in S.)
// The rule "any difference invalidates" redid 5-10 % more accepts (tools/refined_rule_model.py; one frame 4.04 -> 3.94 ms).
// The rule needs the record to list EVERY pixel the seed accepted -- an accepted pixel that is missing counts as a mere neighbour.  A 12,288-frame soak found
// the one place where that failed (profiles/r05_soak_long_refined_rule.txt: texture_frame(61546), 48 bands; tools/repro_lines.py): reduce_region_radius
// overwrote list entries instead of swapping them, so a pixel accepted by refine()'s regrowth and released again dropped out of the list that is logged
// afterwards.  Fixed there (the list stays a permutation); with it the rule is exact on that frame and on 17,600 more
// (profiles/r05_soak_refined_rule_fixed.txt).  -DPLF_SPEC_REFINED_RULE=0 restores the old rule.
#ifndef PLF_SPEC_REFINED_RULE
#define PLF_SPEC_REFINED_RULE 1
#endif
__device__ __forceinline__ bool spec_flag_matters(bool t, bool sv, bool centre)
{
    if (!PLF_SPEC_REFINED_RULE) return t != sv;
    return centre ? (t || sv) : (sv && !t);
}

template <bool SG> struct SpecS {
    LDS_PTR(uint32_t) l;
    uint32_t *g;
    __device__ __forceinline__ bool get(int a) const
    {
        if (SG) return (__hip_atomic_load(&g[a >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (a & 31)) & 1u;
        return (l[a >> 5] >> (a & 31)) & 1u;
    }
    __device__ __forceinline__ void set(int a) const
    {
        if (SG) __hip_atomic_fetch_or(&g[a >> 5], 1u << (a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_or(&l[a >> 5], 1u << (a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ uint32_t word(int i) const { return SG ? __hip_atomic_load(&g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : l[i]; }
    __device__ __forceinline__ void load_from(const uint32_t *__restrict__ src, int words, int lane) const
    {
        for (int i0 = lane; i0 < words; i0 += 512) {   // eight independent loads per lane in flight (one per iteration cost a round trip each: 6144 words = 96 trips per band)
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = i0 + 64 * u < words ? src[i0 + 64 * u] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 64 * u;
                if (i < words) { if (SG) __hip_atomic_store(&g[i], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else l[i] = v[u]; }
            }
        }
    }
    __device__ __forceinline__ void clear_all(int words, int lane) const
    {
        for (int i = lane; i < words; i += 64) { if (SG) __hip_atomic_store(&g[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else l[i] = 0u; }
    }
};

__device__ __forceinline__ bool spec_rounds_converged(const SpecBufs &SB, int f, int last);
// the band's CURRENT log: side 0 = tl / recs / cnt (what the band wave wrote), side 1 = the buffers a validation round wrote (SB.side is zeroed per call)
__device__ __forceinline__ const uint32_t *spec_log_tl(const SpecBufs &SB, size_t fb) { return (SB.side[fb] ? SB.tl_alt : SB.tl) + fb * SB.tcap; }
__device__ __forceinline__ const SpecRec *spec_log_recs(const SpecBufs &SB, size_t fb) { return (SB.side[fb] ? SB.recs_alt : SB.recs) + fb * SB.rcap_rec; }
__device__ __forceinline__ const int *spec_log_cnt(const SpecBufs &SB, size_t fb) { return (SB.side[fb] ? SB.cnt_alt : SB.cnt) + fb * 4; }

// The one-launch schedule makes the commit wave of a frame wait for the band waves of the same launch.  The host only uses it when ALL workgroups of
// the launch can be resident at once (occupancy query), so every band wave is dispatched whatever the dispatch order; the spin is bounded all the
// same (~2^21 x 3 us WITHOUT a heartbeat of the awaited band wave -- it bumps cnt[3] once per seed): a scheduling surprise then surfaces as status
// bit 4 -> PLF_E_HIP for the batch instead of a hung GPU.
__device__ __forceinline__ bool spec_wait_band(const SpecBufs &SB, size_t fb, int *status)
{
    int spins = 0, beat = 0;
    while (__hip_atomic_load(&SB.done[fb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(127);
        if (++spins > SB.spin_bound) { atomicOr(status, 4); return false; }
        if ((spins & (SB.spin_bound >= 4096 ? 1023 : 15)) == 0) {   // a band wave that is still retiring seeds is slow, not missing (quantised images take tens of seconds per frame)
            const int b = __hip_atomic_load(&SB.cnt[fb * 4 + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b != beat) { beat = b; spins = 0; }
        }
    }
    __threadfence();
    return true;
}

// 64 bits of a bitmap starting at bit p (any alignment); words past the end read as zero
__device__ __forceinline__ unsigned long long spec_bits64(const uint32_t *__restrict__ map, int p, int words)
{
    const int i = p >> 5, o = p & 31;
    const unsigned long long w0 = map[i], w1 = i + 1 < words ? map[i + 1] : 0u;
    unsigned long long v = (w0 | (w1 << 32)) >> o;
    if (o && i + 2 < words) v |= (unsigned long long)map[i + 2] << (64 - o);
    return v;
}

__device__ __forceinline__ unsigned long long spec_readlane64(unsigned long long v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

#define SPEC_HCAP 512    // record headers of a commit segment held in LDS
#define SPEC_NRW 1024    // words (32 pixels each) of a commit segment
#define SPEC_COMMIT_EXTRA_WORDS (5 * SPEC_HCAP + SPEC_HCAP / 32 + SPEC_NRW + 1 + 16)   // line_host.hip sizes the dynamic LDS with the same expression
template <bool SG, bool BUDGET>
__device__ __forceinline__ void spec_commit_body(int f, float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                 const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                 int *__restrict__ nrect, int *__restrict__ status, const LsdGeom &g, const SpecBufs &SB, int *__restrict__ stats, int zlast = -1)
{
    if (zlast >= 0 && spec_rounds_converged(SB, f, zlast)) return;   // validation rounds ran and reached their fixpoint: k_lsd_spec_assemble has written the rectangles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int W = g.sw, H = g.sh;
    if (threadIdx.x >= 64) {
        // waves 1..3 only pull what the commit wave will read (written by the band waves on other XCDs) into this XCD's L2, then leave
        const int t = threadIdx.x - 64;
        uint32_t acc = 0;
        const uint32_t *sm = SB.seedmap + (size_t)f * SB.bm_words;
        for (int b = 0; b < SB.nbands; b++) {
            const size_t fb = (size_t)f * SB.nbands + b;
            if (!spec_wait_band(SB, fb, status)) return;
            const int nrec = spec_log_cnt(SB, fb)[0], tn = spec_log_cnt(SB, fb)[1];
            const uint32_t *r = reinterpret_cast<const uint32_t *>(spec_log_recs(SB, fb));
            for (int i = t * 32; i < nrec * (int)(sizeof(SpecRec) / 4); i += 192 * 32) acc ^= r[i];
            const uint32_t *q = spec_log_tl(SB, fb);
            for (int i = t * 32; i < tn; i += 192 * 32) acc ^= q[i];
            const int y0 = SB.band_y[f * (SB.nbands + 1) + b], y1 = SB.band_y[f * (SB.nbands + 1) + b + 1];
            const uint32_t *a = reinterpret_cast<const uint32_t *>(ang_all) + (size_t)f * g.s_stride;
            const uint32_t *c = reinterpret_cast<const uint32_t *>(cs0_all + (size_t)f * g.s_stride);
            for (int i = y0 * W + t * 32; i < y1 * W; i += 192 * 32) acc ^= a[i];
            for (int i = (y0 * W >> 5) + t * 32; i < (y1 * W >> 5); i += 192 * 32) acc ^= sm[i];
            for (int i = y0 * W * 2 + t * 32; i < y1 * W * 2; i += 192 * 32) acc ^= c[i];
        }
        if (acc == 0x9E3779B9u) atomicOr(status, 0);   // keeps the loads alive
        return;
    }
    LDS_PTR(uint32_t) list = (LDS_PTR(uint32_t))smem;
    LDS_PTR(uint32_t) T = list + ((g.rcap + 1 + 15) & ~15);
    // D = S xor T is never stored: a pixel is dirty when its two flags differ.  Sticky coarse map of D: one bit per 8x8-pixel tile (set when
    // a pixel of the tile turns dirty, rebuilt per band).
    SpecS<SG> S;
    S.l = T + SB.bm_words; S.g = SB.sglob + (size_t)f * SB.bm_words;
    // (a row of tiles starts on a word boundary: a record's bounding box is tested one tile ROW at a time -- one or two masked words -- instead of tile by
    // tile; the dependent LDS reads of that test were most of the commit wave's "walking" time: a 200 x 100 pixel box is 325 tiles but 13 rows)
    const int ctx = (((W + 7) >> 3) + 31) & ~31, cty = (H + 7) >> 3, cwords = (ctx >> 5) * cty;
    LDS_PTR(uint32_t) Dc = T + (SG ? 1 : 2) * SB.bm_words;
    // record headers of the segment being committed (seed, first log entry, log entries | has_rect << 31, dilated bounding box), its SUSPECT mask, and the
    // "defined, no record" bits of the segment's pixels
    LDS_PTR(uint32_t) Hseed = Dc + ((cwords + 15) & ~15);
    LDS_PTR(uint32_t) Ht0 = Hseed + SPEC_HCAP;
    LDS_PTR(uint32_t) Hnt = Ht0 + SPEC_HCAP;
    LDS_PTR(uint32_t) Hb0 = Hnt + SPEC_HCAP;
    LDS_PTR(uint32_t) Hb1 = Hb0 + SPEC_HCAP;
    LDS_PTR(uint32_t) Hsus = Hb1 + SPEC_HCAP;
    LDS_PTR(uint32_t) NR = Hsus + SPEC_HCAP / 32;
    for (int i = lane; i < SB.bm_words; i += 64) T[i] = 0u;
    CBAR();
    RegCtx C;
    spec_ctx(C, g, f, ang_all, modgrad_all, cs_all, cs0_all, rxy_all + (size_t)f * g.s_stride, list, T);
    if (BUDGET) C.t_dead = wall_clock64() + g.budget_ticks;
    bool truncated = false;
    __builtin_amdgcn_s_setprio(3);
    LsdRect *rects = rects_all + (size_t)f * g.rect_cap;
    uint32_t *tl2 = SB.tl2 + (size_t)f * 2 * g.s_stride;
    const uint32_t *seedmap = SB.seedmap + (size_t)f * SB.bm_words;
    const uint32_t *defmap = SB.defmap + (size_t)f * SB.bm_words;
    const GrowTh th0 = grow_thresholds(g.prec);
    int nr = 0, n_commit = 0, n_redo = 0, n_fast = 0, n_slow = 0;
    long long c_redo = 0, c_val = 0, c_setup = 0, c_bulk = 0, c_scan = 0, c_rescan = 0, c_seg = 0;
    const long long c_t0 = clock64();
    for (int band = 0; band < SB.nbands; band++) {
        const size_t fb = (size_t)f * SB.nbands + band;
        if (!spec_wait_band(SB, fb, status)) return;
        const long long c_s0 = clock64();
        if (stats && f == 0 && lane == 0 && band < 64) { stats[8 + 3 * band] = SB.cnt[fb * 4 + 3]; stats[8 + 3 * band + 1] = (int)(wall_clock64() & 0x7fffffff); }
        const int use_recs = spec_log_cnt(SB, fb)[2] == 0;   // a band whose log overflowed is simply grown here
        const uint32_t *tl = spec_log_tl(SB, fb);
        const SpecRec *recs = spec_log_recs(SB, fb);
        const int y0 = SB.band_y[f * (SB.nbands + 1) + band], y1 = SB.band_y[f * (SB.nbands + 1) + band + 1];
        if (band > 0 && (SB.halo_rows > 0 || zlast >= 0)) S.load_from(SB.halo + fb * SB.bm_words, SB.bm_words, lane);   // the band's flags after its warm-up rows (after validation rounds: the state its log is consistent with)
        else S.clear_all(SB.bm_words, lane);
        for (int i = lane; i < cwords; i += 64) Dc[i] = 0u;
        CBAR();
        for (int wi = lane; wi < SB.bm_words; wi += 64) {
            uint32_t bits = T[wi] ^ S.word(wi);
            if (!bits) continue;
            if ((W & 31) == 0) {   // a word is 32 pixels of one row = four tiles: one mark per dirty byte
                for (int k = 0; k < 4; k++) if ((bits >> (8 * k)) & 0xFFu) dc_mark(Dc, wi * 32 + 8 * k, W, ctx);
            } else {               // any width: every dirty bit marks its own tile
                while (bits) { const int a = wi * 32 + __ffs((int)bits) - 1; bits &= bits - 1; dc_mark(Dc, a, W, ctx); }
            }
        }
        CBAR();
        c_setup += clock64() - c_s0;
        // ---- The band's records are committed EVENT BY EVENT, not pixel by pixel (round 3; the chunk walk of round 2 paid two or three dependent
        // global round trips -- record headers, accepted-pixel log, rectangle -- for each of ~1800 chunks of a frame: ~4 ms, more than the band waves take).
        // A segment = up to SPEC_HCAP consecutive records and SPEC_NRW words of pixels.  Its record headers are loaded into LDS at once.  A record
        // whose dilated bounding box is clear of dirty tiles is CLEAN: every flag it read had the true value, it stands (the tile map only grows inside
        // a band, and every change is followed by a re-scan of the records still to come).  Runs of CLEAN records are committed in bulk: their marks are
        // one contiguous range of the accepted-pixel log, streamed with independent loads.  Only three things are events, handled in raster order:
        //   * a SUSPECT record (box touches a dirty tile): checked pixel by pixel; it stands, or its marks stay in S only and its seed -- if free in T --
        //     is regrown on T;
        //   * a CANDIDATE pixel: defined, no record, marked in S, free in T -- speculation skipped it because something it believed in took it; it is
        //     grown on T once every record before it is committed and it is still free;
        //   * (a band whose log overflowed has no records: every defined pixel that is free in T when its turn comes is a candidate.)
        // The result is the walk's, statement for statement: same validity rule, same order of rectangles.
        const int p_end = y1 * W;
        const int nrec_band = use_recs ? spec_log_cnt(SB, fb)[0] : 0;
        int r_lo = 0, p_lo = y0 * W;
        while (p_lo < p_end) {
            // ---- segment [p_lo, p_hi) x records [r_lo, r_hi)
            const long long c_g0 = clock64();
            const int nh = min(SPEC_HCAP, nrec_band - r_lo);
            for (int i = lane; i < nh; i += 64) {
                const int4 *hp = reinterpret_cast<const int4 *>(&recs[r_lo + i]);
                const int4 h0 = hp[0], h1 = hp[1];     // seed, t0, nt, has_rect | bx0, by0, bx1, by1
                Hseed[i] = (uint32_t)h0.x; Ht0[i] = (uint32_t)h0.y; Hnt[i] = (uint32_t)h0.z | (h0.w ? 0x80000000u : 0u);
                Hb0[i] = (uint32_t)h1.x | ((uint32_t)h1.y << 16); Hb1[i] = (uint32_t)h1.z | ((uint32_t)h1.w << 16);
            }
            int p_hi = min(p_end, ((p_lo >> 5) + SPEC_NRW) << 5);
            if (r_lo + nh < nrec_band) p_hi = min(p_hi, recs[r_lo + nh].seed);      // the first record that did not fit bounds the segment
            CBAR();
            int nseg = nh;                                    // records of the segment: those with seed < p_hi
            for (int base = 0; base < nh; base += 64) {
                const unsigned long long m = __ballot(base + lane < nh && (int)Hseed[base + lane] >= p_hi);
                if (m) { nseg = base + __ffsll((long long)m) - 1; break; }
            }
            const int w_lo = p_lo >> 5, nw = ((p_hi + 31) >> 5) - w_lo;
            for (int i = lane; i < nw; i += 64) NR[i] = defmap[w_lo + i] & ~(use_recs ? seedmap[w_lo + i] : 0u);
            for (int i = lane; i < SPEC_HCAP / 32; i += 64) Hsus[i] = 0u;
            CBAR();
            bool rescan = true;                               // classify the records against the tile map
            int cur = 0, pos = p_lo;                          // next record of the segment, next pixel position
            c_seg += clock64() - c_g0;
            for (;;) {
                if (BUDGET && wall_clock64() > C.t_dead) { truncated = true; break; }
                const long long c_e0 = clock64();
                if (rescan) {
                    for (int i = cur + lane; i < nseg; i += 64) {
                        const uint32_t b0 = Hb0[i], b1 = Hb1[i];
                        const int tx0 = (int)(b0 & 0xFFFFu) >> 3, ty0 = (int)(b0 >> 16) >> 3, tx1 = (int)(b1 & 0xFFFFu) >> 3, ty1 = (int)(b1 >> 16) >> 3;
                        const int w0 = tx0 >> 5, w1 = tx1 >> 5;
                        const uint32_t m0 = 0xFFFFFFFFu << (tx0 & 31), m1 = 0xFFFFFFFFu >> (31 - (tx1 & 31));
                        bool clean = true;
                        for (int ty = ty0; ty <= ty1 && clean; ty++) {
                            LDS_PTR(uint32_t) row = Dc + ty * (ctx >> 5);
                            for (int wq = w0; wq <= w1; wq++) {
                                uint32_t bits = row[wq];
                                if (wq == w0) bits &= m0;
                                if (wq == w1) bits &= m1;
                                if (bits) { clean = false; break; }
                            }
                        }
                        if (!clean) __hip_atomic_fetch_or(&Hsus[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    CBAR();
                    rescan = false;
                }
                const long long c_e1 = clock64();
                c_rescan += c_e1 - c_e0;
                // next SUSPECT record at or after cur
                int rs = nseg;
                {
                    uint32_t wv = 0u;
                    if (lane < SPEC_HCAP / 32 && lane >= (cur >> 5)) { wv = Hsus[lane]; if (lane == (cur >> 5)) wv &= 0xFFFFFFFFu << (cur & 31); }
                    const unsigned long long m = __ballot(wv != 0u);
                    if (m) {
                        const int l0 = __ffsll((long long)m) - 1;
                        rs = min(nseg, l0 * 32 + __ffs((int)__builtin_amdgcn_readlane((int)wv, l0)) - 1);
                    }
                }
                const int ps = rs < nseg ? (int)Hseed[rs] : p_hi;
                // first CANDIDATE pixel in [pos, ps)
                int pc = -1;
                for (int wb = (pos >> 5); wb <= ((ps - 1) >> 5) && pos < ps; wb += 64) {
                    const int wq = wb + lane;
                    uint32_t bits = 0u;
                    if (wq <= ((ps - 1) >> 5)) {
                        bits = NR[wq - w_lo] & ~T[wq];
                        if (use_recs) bits &= S.word(wq);
                        if (wq == (pos >> 5)) bits &= 0xFFFFFFFFu << (pos & 31);
                        if (wq == ((ps - 1) >> 5) && (ps & 31)) bits &= 0xFFFFFFFFu >> (32 - (ps & 31));
                    }
                    const unsigned long long m = __ballot(bits != 0u);
                    if (m) {
                        const int l0 = __ffsll((long long)m) - 1;
                        pc = (wb + l0) * 32 + __ffs((int)__builtin_amdgcn_readlane((int)bits, l0)) - 1;
                        break;
                    }
                }
                // the CLEAN records before the event stand: commit them in bulk
                const long long c_e2 = clock64();
                c_scan += c_e2 - c_e1;
                int rk = rs;                                  // first record NOT committed now
                if (pc >= 0) {
                    rk = cur;
                    for (int base = cur; base < rs; base += 64) {
                        const unsigned long long m = __ballot(base + lane >= rs || (int)Hseed[min(base + lane, SPEC_HCAP - 1)] > pc);
                        if (m) { rk = min(rs, base + __ffsll((long long)m) - 1); break; }
                        rk = min(rs, base + 64);
                    }
                }
                if (rk > cur) {
                    const int t_begin = (int)Ht0[cur], t_stop = (int)Ht0[rk - 1] + (int)(Hnt[rk - 1] & 0x7FFFFFFFu);
                    for (int i0 = t_begin; i0 < t_stop; i0 += 256) {          // four independent loads per lane in flight
                        uint32_t e[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { const int i = i0 + u * 64 + lane; e[u] = i < t_stop ? tl[i] : 0u; }
#pragma unroll
                        for (int u = 0; u < 4; u++) if (e[u] & 0x40000000u) { bm_set(T, (int)(e[u] & 0x3FFFFFFFu)); S.set((int)(e[u] & 0x3FFFFFFFu)); }
                    }
                    for (int base = cur; base < rk; base += 64) {
                        const int i = base + lane;
                        const bool hr = i < rk && (Hnt[min(i, SPEC_HCAP - 1)] & 0x80000000u);
                        const unsigned long long rm = __ballot(hr);
                        if (hr) {
                            const int slot = nr + __popcll(rm & ((1ull << lane) - 1ull));
                            if (slot < g.rect_cap) rects[slot] = recs[r_lo + i].rec; else atomicOr(status, 1);
                        }
                        nr += __popcll(rm);
                    }
                    n_commit += rk - cur; n_fast++;
                    cur = rk;
                    CBAR();
                }
                c_bulk += clock64() - c_e2;
                int gseed = -1;                               // pixel to grow on the true flags, if the event calls for it
                if (pc >= 0) {
                    // candidate pixel: every record before it is committed; is it still free?
                    pos = pc + 1;
                    if (bm_get(T, pc)) continue;
                    gseed = pc;
                } else {
                if (rs >= nseg) break;                        // no event left in the segment
                // SUSPECT record rs: the walk's pixel-by-pixel test
                {
                    const int seed = (int)Hseed[rs], t0 = (int)Ht0[rs], nt = (int)(Hnt[rs] & 0x7FFFFFFFu);
                    const bool has_rect = (Hnt[rs] & 0x80000000u) != 0u;
                    const bool true_eff = !bm_get(T, seed);
                    bool valid = true_eff;
                    const long long c_v0 = clock64();
                    n_slow++;
                    for (int i0 = 0; i0 < nt && valid; i0 += 64) {
                        const int i = i0 + lane;
                        bool hit = false;
                        if (i < nt) {
                            const int q = (int)(tl[t0 + i] & 0x3FFFFFFFu), qx = q % W, qy = q / W;
                            for (int dy = -1; dy <= 1; dy++) {
                                const int yy = qy + dy;
                                if (yy < 0 || yy >= H) continue;
                                for (int dx = -1; dx <= 1; dx++) {
                                    const int xx = qx + dx;
                                    if (xx < 0 || xx >= W) continue;
                                    hit |= spec_flag_matters(bm_get(T, yy * W + xx), S.get(yy * W + xx), dx == 0 && dy == 0);
                                }
                            }
                        }
                        if (__ballot(hit)) valid = false;
                    }
                    c_val += clock64() - c_v0;
                    cur = rs + 1; pos = seed + 1;
                    if (valid) {   // every flag the speculative run read was the true one: take its marks and its rectangle
                        for (int i = lane; i < nt; i += 64) { const uint32_t e = tl[t0 + i]; if (e & 0x40000000u) { bm_set(T, (int)(e & 0x3FFFFFFFu)); S.set((int)(e & 0x3FFFFFFFu)); } }
                        if (has_rect) { if (nr < g.rect_cap) { if (lane == 0) rects[nr] = recs[r_lo + rs].rec; } else if (lane == 0) atomicOr(status, 1); nr++; }
                        n_commit++;
                        CBAR();
                        continue;
                    }
                    // the speculative timeline keeps its own marks
                    for (int i = lane; i < nt; i += 64) { const uint32_t e = tl[t0 + i]; if (e & 0x40000000u) { const int q = (int)(e & 0x3FFFFFFFu); S.set(q); if (!bm_get(T, q)) dc_mark(Dc, q, W, ctx); } }
                    CBAR();
                    if (true_eff) gseed = seed;
                    rescan = true;
                }
                }
                if (gseed >= 0) {   // grow on the true flags
                    const float sdeg = __uint_as_float(C.ang[gseed]);
                    const float2 sc0 = C.cs0[gseed];
                    LsdRect rec;
                    int tn = 0, ovf = 0;
                    const long long c_r0 = clock64();
                    const bool okr = spec_seed(C, g, th0, gseed, sdeg, sc0, rec, tl2, tn, 2 * (int)g.s_stride, ovf);
                    c_redo += clock64() - c_r0;
                    if (okr) { if (nr < g.rect_cap) { if (lane == 0) rects[nr] = rec; } else if (lane == 0) atomicOr(status, 1); nr++; }
                    CBAR();
                    for (int i = lane; i < tn; i += 64) { const int q = (int)tl2[i]; if (S.get(q) != bm_get(T, q)) dc_mark(Dc, q, W, ctx); }
                    CBAR();
                    n_redo++;
                    rescan = true;
                }
            }
            if (BUDGET && truncated) break;
            r_lo += nseg; p_lo = p_hi;
        }
        if (BUDGET && truncated) break;
    }
    if (BUDGET && truncated && lane == 0) { atomicOr(status, 8); status[16 + f] = 1; }
    if (stats && f == 0 && lane == 0) stats[8 + 3 * 63 + 2] = (int)(wall_clock64() & 0x7fffffff);   // end of the commit
    if (lane == 0) {
        nrect[f] = min(nr, g.rect_cap);
        if (stats) {
            int *st = stats + 8 * f;
            st[0] = n_commit; st[1] = n_redo; st[2] = n_fast; st[3] = n_slow; st[4] = (int)(c_redo >> 10); st[5] = (int)(c_val >> 10); st[6] = (int)((clock64() - c_t0) >> 10); st[7] = (int)(c_setup >> 10);
            if (f == 0) { stats[201] = (int)(c_bulk >> 10); stats[202] = (int)(c_scan >> 10); stats[203] = (int)(c_rescan >> 10); stats[204] = (int)(c_seg >> 10); }   // (kilo-cycles: bulk commits, event scans, re-classification, segment set-up)
        }
    }
}

__global__ void __launch_bounds__(256) k_lsd_spec_commit(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                        const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                        int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int *__restrict__ stats)
{
    if (SB.s_global) spec_commit_body<true, false>(blockIdx.x, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats);
    else spec_commit_body<false, false>(blockIdx.x, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats);
}
// after `zlast` validation rounds: only the frames that did not converge (or whose logs overflowed) are committed serially, from the logs as they stand
__global__ void __launch_bounds__(256) k_lsd_spec_commit_rest(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                             const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                             int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int *__restrict__ stats, int zlast)
{
    if (SB.s_global) spec_commit_body<true, false>(blockIdx.x, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats, zlast);
    else spec_commit_body<false, false>(blockIdx.x, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats, zlast);
}
__global__ void __launch_bounds__(256) k_lsd_spec_commit_budget(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                        const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                        int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int *__restrict__ stats)
{
    if (SB.s_global) spec_commit_body<true, true>(blockIdx.x, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats);
    else spec_commit_body<false, true>(blockIdx.x, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats);
}

// Both phases in one launch (few frames): workgroups [0, B * nbands) are the band waves, the last B the commit waves.  Workgroups are dispatched in
// index order, so every band wave is resident or finished before a commit wave starts waiting for it; the commit wave then follows the
// bands as they finish (band 0 is the smallest, see k_lsd_spec_bands).
template <bool BUDGET>
__device__ __forceinline__ void spec_fused_body(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                       const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                       int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int *__restrict__ stats, int B)
{
    const int L = blockIdx.x, nb = B * SB.nbands;
    if (L < nb) {
        spec_grow_body<BUDGET>(L % SB.nbands, L / SB.nbands, ang_all, modgrad_all, cs_all, cs0_all, g, SB);
    } else {
        if (SB.s_global) spec_commit_body<true, BUDGET>(L - nb, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats);
        else spec_commit_body<false, BUDGET>(L - nb, ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats);
    }
}

__global__ void __launch_bounds__(256) k_lsd_spec_fused(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                       const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                       int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int *__restrict__ stats, int B)
{
    spec_fused_body<false>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats, B);
}
__global__ void __launch_bounds__(256) k_lsd_spec_fused_budget(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                       const float2 *__restrict__ cs0_all, uint32_t *__restrict__ rxy_all, LsdRect *__restrict__ rects_all,
                                                       int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int *__restrict__ stats, int B)
{
    spec_fused_body<true>(ang_all, modgrad_all, cs_all, cs0_all, rxy_all, rects_all, nrect, status, g, SB, stats, B);
}

// ------------------------------------------------------------------------------------------------
// Parallel validation rounds (round 3; model: oracle/lsd_oracle.c, orc_lsd_band_rounds).  The serial commit wave above spends 4-6 ms per VGA frame AFTER the
// band waves are done (2 ms of it regrowing ~90 regions one after the other).  Instead, EVERY band validates itself, all bands at once:
//   out[b]  = the pixels band b's own records mark;   pre[b] = union of out[b'] for b' < b  (k_lsd_spec_prefix);
//   E_b     = the state of the earlier bands that band b's log is consistent with (initially what its warm-up rows left: SB.halo).
// A round: where pre[b] differs from E_b, band b replays the commit wave's event loop on T = pre[b], S = E_b over its own records and writes a NEW log (the other
// side of the ping-pong buffers): records that stand are copied, the others dropped or regrown on T, skipped seeds that are free grown.  That log IS the serial
// processing of the band's seeds from pre[b]; E_b := pre[b], out[b] := T minus pre[b].  Band 0 never changes, so after round r the bands 0..r are final; the
// rounds stop when no out[] changed (3-6 rounds on the synthetic frames; the host enqueues a fixed number, later launches return at once).  A frame that has not
// converged by then, or whose logs overflowed, is finished by the serial commit wave from the logs as they stand (every log is consistent with its own E_b, which
// is all that wave needs) -- so the result is exact either way.  Rectangles: k_lsd_spec_assemble concatenates the bands' records in order.
// round_state[f] = {changed bands in even rounds, in odd rounds, -, fall back}
// ------------------------------------------------------------------------------------------------
// round_state[f] = {bands whose marks changed in the even rounds, in the odd rounds, converged (sticky), fall back to the serial commit (sticky)}
__device__ __forceinline__ bool spec_rounds_active(const SpecBufs &SB, int f)
{
    const int *rs = SB.round_state + f * 4;
    return !rs[2] && !rs[3];
}
// after `last` rounds: a fixpoint was reached (some round found the one before it unchanged, or the last round changed nothing)
__device__ __forceinline__ bool spec_rounds_converged(const SpecBufs &SB, int f, int last)
{
    const int *rs = SB.round_state + f * 4;
    return !rs[3] && (rs[2] || rs[last & 1] == 0);
}

__global__ void __launch_bounds__(256) k_lsd_spec_prefix(SpecBufs SB, int round)
{
    const int f = blockIdx.y, w = blockIdx.x * 256 + threadIdx.x;
    if (!spec_rounds_active(SB, f)) return;
    if (w >= SB.bm_words) return;
    const size_t base = (size_t)f * SB.nbands * SB.bm_words + w;
    uint32_t acc = 0u;
    for (int b = 0; b < SB.nbands; b++) {
        SB.pre[base + (size_t)b * SB.bm_words] = acc;
        acc |= SB.out[base + (size_t)b * SB.bm_words];
    }
}

template <bool SG>
__device__ __forceinline__ void spec_validate_body(int band, int f, float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                   const float2 *__restrict__ cs0_all, const LsdGeom &g, const SpecBufs &SB, int round)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int W = g.sw, H = g.sh;
    const size_t fb = (size_t)f * SB.nbands + band;
    int *rs = SB.round_state + f * 4;
    LDS_PTR(uint32_t) list = (LDS_PTR(uint32_t))smem;
    LDS_PTR(uint32_t) T = list + ((g.rcap + 1 + 15) & ~15);
    SpecS<SG> S;
    S.l = T + SB.bm_words; S.g = SB.sglob + fb * SB.bm_words;          // (s_global frames: one S bitmap per band here)
    const int ctx = (((W + 7) >> 3) + 31) & ~31, cty = (H + 7) >> 3, cwords = (ctx >> 5) * cty;
    LDS_PTR(uint32_t) Dc = T + (SG ? 1 : 2) * SB.bm_words;
    LDS_PTR(uint32_t) Hseed = Dc + ((cwords + 15) & ~15);
    LDS_PTR(uint32_t) Ht0 = Hseed + SPEC_HCAP;
    LDS_PTR(uint32_t) Hnt = Ht0 + SPEC_HCAP;
    LDS_PTR(uint32_t) Hb0 = Hnt + SPEC_HCAP;
    LDS_PTR(uint32_t) Hb1 = Hb0 + SPEC_HCAP;
    LDS_PTR(uint32_t) Hsus = Hb1 + SPEC_HCAP;
    LDS_PTR(uint32_t) NR = Hsus + SPEC_HCAP / 32;
    const uint32_t *pre = SB.pre + fb * SB.bm_words;
    uint32_t *halo = SB.halo + fb * SB.bm_words;
    // T = what the bands before this one mark now, S = what this band's log assumed they mark; nothing to do where they agree
    // (round 5: ALL FOUR waves of the workgroup load the two bitmaps -- 2 x 24 KB at VGA, 192 dependent-latency loads per lane of one wave: the 28 us every band
    // paid in every round it ran, the whole cost of a round in which nothing is regrown; the other three waves leave behind the barrier)
    // Round 5: the words in which they differ are bracketed on the way; if all of them lie in rows this band's log cannot depend on -- outside its own rows and
    // the dilated boxes of its records (spec_reach) -- the serial processing of its seeds from the new state leaves the same log, every flag it reads being
    // unchanged: E_b := pre[b], nothing else.  (The late rounds of a call are mostly that: one band regrows a region, every band below it walked all of its
    // records -- 30-80 us each -- to find nothing to do.)
    __shared__ int s_dirty[2];
    if (threadIdx.x == 0) { s_dirty[0] = 0x7fffffff; s_dirty[1] = -1; }
    __syncthreads();
    bool differ = false;
    int wlo = 0x7fffffff, whi = -1;
    for (int i0 = lane; i0 < SB.bm_words; i0 += 1024) {
        uint32_t a[4], e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u; a[u] = i < SB.bm_words ? pre[i] : 0u; e[u] = i < SB.bm_words ? halo[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 256 * u;
            if (i < SB.bm_words) {
                T[i] = a[u]; if (SG) __hip_atomic_store(&S.g[i], e[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else S.l[i] = e[u];
                if (a[u] != e[u]) { differ = true; wlo = min(wlo, i); whi = max(whi, i); }
            }
        }
    }
    if (differ) { atomicMin(&s_dirty[0], wlo); atomicMax(&s_dirty[1], whi); }
    if (SG) __threadfence();
    if (!__syncthreads_or(differ ? 1 : 0)) return;
#ifndef PLF_SPEC_NO_REACH_SKIP
    {
        const int dlo = s_dirty[0], dhi = s_dirty[1];
        const int dy0 = (dlo * 32) / W, dy1 = min(H - 1, (dhi * 32 + 31) / W);
        const int *rc = spec_reach(SB, fb);
        if (dy1 < rc[0] || dy0 > rc[1]) {
            for (int i = dlo + (int)threadIdx.x; i <= dhi; i += 256) { const uint32_t a = pre[i]; if (a != halo[i]) halo[i] = a; }
            return;
        }
    }
#endif
    if (threadIdx.x >= 64) return;
    for (int i = lane; i < cwords; i += 64) Dc[i] = 0u;
    CBAR();
    for (int wi = lane; wi < SB.bm_words; wi += 64) {
        uint32_t bits = T[wi] ^ S.word(wi);
        if (!bits) continue;
        if ((W & 31) == 0) { for (int k = 0; k < 4; k++) if ((bits >> (8 * k)) & 0xFFu) dc_mark(Dc, wi * 32 + 8 * k, W, ctx); }
        else { while (bits) { const int a = wi * 32 + __ffs((int)bits) - 1; bits &= bits - 1; dc_mark(Dc, a, W, ctx); } }
    }
    CBAR();
    RegCtx C;
    spec_ctx(C, g, f, ang_all, modgrad_all, cs_all, cs0_all, SB.rxy + fb * g.s_stride, list, T);
    __builtin_amdgcn_s_setprio(3);
    const GrowTh th0 = grow_thresholds(g.prec);
    const int sd = SB.side[fb];
    const uint32_t *tl = (sd ? SB.tl_alt : SB.tl) + fb * SB.tcap;
    const SpecRec *recs = (sd ? SB.recs_alt : SB.recs) + fb * SB.rcap_rec;
    const int *cnt = (sd ? SB.cnt_alt : SB.cnt) + fb * 4;
    uint32_t *tl_n = (sd ? SB.tl : SB.tl_alt) + fb * SB.tcap;            // the new log goes to the other side
    SpecRec *recs_n = (sd ? SB.recs : SB.recs_alt) + fb * SB.rcap_rec;
    int *cnt_n = (sd ? SB.cnt : SB.cnt_alt) + fb * 4;
    uint32_t *tl2 = SB.tl2b + fb * 2 * g.s_stride;
    uint32_t *seedmap = SB.seedmap + (size_t)f * SB.bm_words;
    const uint32_t *defmap = SB.defmap + (size_t)f * SB.bm_words;
    const int y0 = SB.band_y[f * (SB.nbands + 1) + band], y1 = SB.band_y[f * (SB.nbands + 1) + band + 1];
    int nrec_n = 0, tn_n = 0, nrect_n = 0;
    int reach0 = y0, reach1 = max(y1 - 1, y0), lr0 = H, lr1 = -1;   // rows the new log depends on (spec_reach); lr*: per-lane, over the records copied in runs
    bool ovf_n = false;
    const int p_end = y1 * W;
    const int nrec_band = cnt[0];
#ifdef PLF_ROUND_LOG
    const unsigned long long rl_t0 = wall_clock64();
    int rl_seeds = 0, rl_px = 0, rl_same = 0, rl_same_px = 0, rl_old_t0 = -1, rl_old_nt = 0, rl_b_only = 0, rl_b_small = 0;   // (same: regrown records that came out with the accepted list they had)
#endif
    int r_lo = 0, p_lo = y0 * W;
    while (p_lo < p_end && !ovf_n) {
        const int nh = min(SPEC_HCAP, nrec_band - r_lo);
        for (int i = lane; i < nh; i += 64) {
            const int4 *hp = reinterpret_cast<const int4 *>(&recs[r_lo + i]);
            const int4 h0 = hp[0], h1 = hp[1];
            Hseed[i] = (uint32_t)h0.x; Ht0[i] = (uint32_t)h0.y; Hnt[i] = (uint32_t)h0.z | (h0.w ? 0x80000000u : 0u);
            Hb0[i] = (uint32_t)h1.x | ((uint32_t)h1.y << 16); Hb1[i] = (uint32_t)h1.z | ((uint32_t)h1.w << 16);
        }
        int p_hi = min(p_end, ((p_lo >> 5) + SPEC_NRW) << 5);
        if (r_lo + nh < nrec_band) p_hi = min(p_hi, recs[r_lo + nh].seed);
        CBAR();
        int nseg = nh;
        for (int base = 0; base < nh; base += 64) {
            const unsigned long long m = __ballot(base + lane < nh && (int)Hseed[base + lane] >= p_hi);
            if (m) { nseg = base + __ffsll((long long)m) - 1; break; }
        }
        const int w_lo = p_lo >> 5, nw = ((p_hi + 31) >> 5) - w_lo;
        for (int i = lane; i < nw; i += 64) NR[i] = defmap[w_lo + i] & ~seedmap[w_lo + i];
        for (int i = lane; i < SPEC_HCAP / 32; i += 64) Hsus[i] = 0u;
        CBAR();
        bool rescan = true;
        int cur = 0, pos = p_lo;
        while (!ovf_n) {
            if (rescan) {
                for (int i = cur + lane; i < nseg; i += 64) {
                    const uint32_t b0 = Hb0[i], b1 = Hb1[i];
                    const int tx0 = (int)(b0 & 0xFFFFu) >> 3, ty0 = (int)(b0 >> 16) >> 3, tx1 = (int)(b1 & 0xFFFFu) >> 3, ty1 = (int)(b1 >> 16) >> 3;
                    const int w0 = tx0 >> 5, w1 = tx1 >> 5;
                    const uint32_t m0 = 0xFFFFFFFFu << (tx0 & 31), m1 = 0xFFFFFFFFu >> (31 - (tx1 & 31));
                    bool clean = true;
                    for (int ty = ty0; ty <= ty1 && clean; ty++) {
                        LDS_PTR(uint32_t) row = Dc + ty * (ctx >> 5);
                        for (int wq = w0; wq <= w1; wq++) {
                            uint32_t bits = row[wq];
                            if (wq == w0) bits &= m0;
                            if (wq == w1) bits &= m1;
                            if (bits) { clean = false; break; }
                        }
                    }
                    if (!clean) __hip_atomic_fetch_or(&Hsus[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                CBAR();
                rescan = false;
            }
            int rsu = nseg;
            {
                uint32_t wv = 0u;
                if (lane < SPEC_HCAP / 32 && lane >= (cur >> 5)) { wv = Hsus[lane]; if (lane == (cur >> 5)) wv &= 0xFFFFFFFFu << (cur & 31); }
                const unsigned long long m = __ballot(wv != 0u);
                if (m) {
                    const int l0 = __ffsll((long long)m) - 1;
                    rsu = min(nseg, l0 * 32 + __ffs((int)__builtin_amdgcn_readlane((int)wv, l0)) - 1);
                }
            }
            const int ps = rsu < nseg ? (int)Hseed[rsu] : p_hi;
            int pc = -1;
            for (int wb = (pos >> 5); wb <= ((ps - 1) >> 5) && pos < ps; wb += 64) {
                const int wq = wb + lane;
                uint32_t bits = 0u;
                if (wq <= ((ps - 1) >> 5)) {
                    bits = NR[wq - w_lo] & ~T[wq] & S.word(wq);
                    if (wq == (pos >> 5)) bits &= 0xFFFFFFFFu << (pos & 31);
                    if (wq == ((ps - 1) >> 5) && (ps & 31)) bits &= 0xFFFFFFFFu >> (32 - (ps & 31));
                }
                const unsigned long long m = __ballot(bits != 0u);
                if (m) {
                    const int l0 = __ffsll((long long)m) - 1;
                    pc = (wb + l0) * 32 + __ffs((int)__builtin_amdgcn_readlane((int)bits, l0)) - 1;
                    break;
                }
            }
            int rk = rsu;
            if (pc >= 0) {
                rk = cur;
                for (int base = cur; base < rsu; base += 64) {
                    const unsigned long long m = __ballot(base + lane >= rsu || (int)Hseed[min(base + lane, SPEC_HCAP - 1)] > pc);
                    if (m) { rk = min(rsu, base + __ffsll((long long)m) - 1); break; }
                    rk = min(rsu, base + 64);
                }
            }
            if (rk > cur) {   // a run of records that stand: marks into T and S, log entries and records copied to the new log
                const int t_begin = (int)Ht0[cur], t_stop = (int)Ht0[rk - 1] + (int)(Hnt[rk - 1] & 0x7FFFFFFFu);
                if (tn_n + (t_stop - t_begin) > SB.tcap || nrec_n + (rk - cur) > SB.rcap_rec) { ovf_n = true; break; }
                for (int i0 = t_begin; i0 < t_stop; i0 += 256) {
                    uint32_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int i = i0 + u * 64 + lane; e[u] = i < t_stop ? tl[i] : 0u; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * 64 + lane;
                        if (i < t_stop) tl_n[tn_n + (i - t_begin)] = e[u];
                        if (e[u] & 0x40000000u) { bm_set(T, (int)(e[u] & 0x3FFFFFFFu)); S.set((int)(e[u] & 0x3FFFFFFFu)); }
                    }
                }
                for (int base = cur; base < rk; base += 64) {
                    const int i = base + lane;
                    const bool in = i < rk;
                    if (in) {
                        const int4 *src = reinterpret_cast<const int4 *>(&recs[r_lo + i]);
                        int4 *dst = reinterpret_cast<int4 *>(&recs_n[nrec_n + (i - cur)]);
                        int4 v0 = src[0];
                        v0.y += tn_n - t_begin;                 // t0 in the new log
                        dst[0] = v0;
                        lr0 = min(lr0, (int)(Hb0[i] >> 16)); lr1 = max(lr1, (int)(Hb1[i] >> 16));
#pragma unroll
                        for (int q = 1; q < (int)(sizeof(SpecRec) / 16); q++) dst[q] = src[q];
                    }
                    nrect_n += __popcll(__ballot(in && (Hnt[min(i, SPEC_HCAP - 1)] & 0x80000000u)));
                }
                tn_n += t_stop - t_begin; nrec_n += rk - cur;
                cur = rk;
                CBAR();
            }
            int gseed = -1;
#ifdef PLF_ROUND_LOG
            rl_old_t0 = -1;
#endif
            if (pc >= 0) {
                pos = pc + 1;
                if (bm_get(T, pc)) continue;
                gseed = pc;
            } else {
                if (rsu >= nseg) break;
                const int seed = (int)Hseed[rsu], t0 = (int)Ht0[rsu], nt = (int)(Hnt[rsu] & 0x7FFFFFFFu);
                const bool true_eff = !bm_get(T, seed);
                bool valid = true_eff;
                for (int i0 = 0; i0 < nt && valid; i0 += 64) {
                    const int i = i0 + lane;
                    bool hit = false;
                    if (i < nt) {
                        const int q = (int)(tl[t0 + i] & 0x3FFFFFFFu), qx = q % W, qy = q / W;
                        for (int dy = -1; dy <= 1; dy++) {
                            const int yy = qy + dy;
                            if (yy < 0 || yy >= H) continue;
                            for (int dx = -1; dx <= 1; dx++) {
                                const int xx = qx + dx;
                                if (xx < 0 || xx >= W) continue;
                                hit |= spec_flag_matters(bm_get(T, yy * W + xx), S.get(yy * W + xx), dx == 0 && dy == 0);
                            }
                        }
                    }
                    if (__ballot(hit)) valid = false;
                }
                cur = rsu + 1; pos = seed + 1;
#ifdef PLF_ROUND_LOG
                if (!valid && true_eff) {   // why: an accepted pixel truly taken (a), or only pixels the speculation saw taken that are truly free (b)
                    bool any_a = false;
                    for (int i = lane; i < nt; i += 64) any_a |= bm_get(T, (int)(tl[t0 + i] & 0x3FFFFFFFu));
                    if (!__ballot(any_a)) { rl_b_only++; if (nt < g.min_reg_size) rl_b_small++; }
                }
#endif
                if (valid) {
                    if (tn_n + nt > SB.tcap || nrec_n + 1 > SB.rcap_rec) { ovf_n = true; break; }
                    for (int i = lane; i < nt; i += 64) {
                        const uint32_t e = tl[t0 + i];
                        tl_n[tn_n + i] = e;
                        if (e & 0x40000000u) { bm_set(T, (int)(e & 0x3FFFFFFFu)); S.set((int)(e & 0x3FFFFFFFu)); }
                    }
                    if (lane == 0) { SpecRec r = recs[r_lo + rsu]; r.t0 = tn_n; recs_n[nrec_n] = r; }
                    reach0 = min(reach0, (int)(Hb0[rsu] >> 16)); reach1 = max(reach1, (int)(Hb1[rsu] >> 16));
                    if (Hnt[rsu] & 0x80000000u) nrect_n++;
                    tn_n += nt; nrec_n++;
                    CBAR();
                    continue;
                }
                for (int i = lane; i < nt; i += 64) { const uint32_t e = tl[t0 + i]; if (e & 0x40000000u) { const int q = (int)(e & 0x3FFFFFFFu); S.set(q); if (!bm_get(T, q)) dc_mark(Dc, q, W, ctx); } }
                CBAR();
                if (true_eff) gseed = seed;
#ifdef PLF_ROUND_LOG
                rl_old_t0 = t0; rl_old_nt = nt;
#endif
                rescan = true;
            }
            if (gseed >= 0) {   // grow on T; the result is a record of the new log
                const float sdeg = __uint_as_float(C.ang[gseed]);
                const float2 sc0 = C.cs0[gseed];
                LsdRect rec;
                int tn = 0, ovf = 0;
                const bool okr = spec_seed(C, g, th0, gseed, sdeg, sc0, rec, tl2, tn, 2 * (int)g.s_stride, ovf);
                CBAR();
                if (ovf || tn_n + tn > SB.tcap || nrec_n + 1 > SB.rcap_rec) { ovf_n = true; break; }
                int bx0 = W, by0 = H, bx1 = -1, by1 = -1;
                for (int i = lane; i < tn; i += 64) {
                    const uint32_t q = tl2[i];
                    const int qx = (int)(q % (uint32_t)W), qy = (int)(q / (uint32_t)W);
                    bx0 = min(bx0, qx); bx1 = max(bx1, qx); by0 = min(by0, qy); by1 = max(by1, qy);
                    tl_n[tn_n + i] = q | (bm_get(T, (int)q) ? 0x40000000u : 0u);
                    if (S.get((int)q) != bm_get(T, (int)q)) dc_mark(Dc, (int)q, W, ctx);
                }
                bx0 = wave_min_i(bx0); by0 = wave_min_i(by0); bx1 = wave_max_i(bx1); by1 = wave_max_i(by1);
                if (lane == 0) {
                    int4 *dst = reinterpret_cast<int4 *>(&recs_n[nrec_n]);
                    dst[0] = make_int4(gseed, tn_n, tn, okr ? 1 : 0);
                    dst[1] = make_int4(max(bx0 - 1, 0), max(by0 - 1, 0), min(bx1 + 1, W - 1), min(by1 + 1, H - 1));
                    if (okr) recs_n[nrec_n].rec = rec;
                }
                reach0 = min(reach0, max(by0 - 1, 0)); reach1 = max(reach1, min(by1 + 1, H - 1));
                if (okr) nrect_n++;
#ifdef PLF_ROUND_LOG
                rl_seeds++; rl_px += tn;
                if (rl_old_t0 >= 0 && tn == rl_old_nt) {
                    bool diff = false;
                    for (int i = lane; i < tn; i += 64) diff |= tl2[i] != (tl[rl_old_t0 + i] & 0x3FFFFFFFu);
                    if (!__ballot(diff)) { rl_same++; rl_same_px += tn; }
                }
#endif
                tn_n += tn; nrec_n++;
                CBAR();
                rescan = true;
            }
        }
        r_lo += nseg; p_lo = p_hi;
    }
    if (ovf_n) { if (lane == 0) rs[3] = 1; return; }   // the old log, E_b, out[] and the seed map are untouched: the serial commit takes the frame
    // ---- the new log is the band's log now
    __threadfence();
    uint32_t *outb = SB.out + fb * SB.bm_words;
    bool changed = false;
    for (int i = lane; i < SB.bm_words; i += 64) {
        const uint32_t e = pre[i], o = T[i] & ~e;
        if (o != outb[i]) { changed = true; outb[i] = o; }
        halo[i] = e;
    }
    // seed map of the band's rows: the seeds of the new log
    {
        const int a0 = y0 * W, a1 = p_end;
        for (int wq = (a0 >> 5) + lane; wq <= ((a1 - 1) >> 5); wq += 64) {
            uint32_t keep = 0u;
            if (wq == (a0 >> 5)) keep |= (a0 & 31) ? ((1u << (a0 & 31)) - 1u) : 0u;
            if (wq == ((a1 - 1) >> 5) && (a1 & 31)) keep |= ~((1u << (a1 & 31)) - 1u);
            if (keep) atomicAnd(&seedmap[wq], keep); else seedmap[wq] = 0u;
        }
        __threadfence();
        for (int i = lane; i < nrec_n; i += 64) { const int sdp = recs_n[i].seed; atomicOr(&seedmap[sdp >> 5], 1u << (sdp & 31)); }
    }
    reach0 = min(reach0, wave_min_i(lr0)); reach1 = max(reach1, wave_max_i(lr1));
    if (lane == 0) {
        cnt_n[0] = nrec_n; cnt_n[1] = tn_n; cnt_n[2] = 0; cnt_n[3] = 0;
        SB.side[fb] = sd ^ 1;
        SB.nrects[fb] = nrect_n;
        int *rc = spec_reach(SB, fb); rc[0] = reach0; rc[1] = reach1;
    }
    if (__ballot(changed) && lane == 0) atomicAdd(&rs[round & 1], 1);
#ifdef PLF_ROUND_LOG
    if (SB.round_log && lane == 0 && round >= 1 && round <= 16) {
        int *rl = SB.round_log + (fb * 16 + (round - 1)) * 4;
        rl[0] = (int)(wall_clock64() - rl_t0); rl[1] = rl_seeds | (rl_same << 16); rl[2] = min(rl_px, 0xFFFF) | (min(rl_same_px, 0xFFFF) << 16); rl[3] = min(nrec_n - rl_seeds, 0xFFFF) | (min(rl_b_only, 255) << 16) | (min(rl_b_small, 255) << 24);
    }
#endif
}

__global__ void __launch_bounds__(256) k_lsd_spec_validate(float *__restrict__ ang_all, const double *__restrict__ modgrad_all, const double2 *__restrict__ cs_all,
                                                          const float2 *__restrict__ cs0_all, LsdGeom g, SpecBufs SB, int round)
{
    const int band = blockIdx.x, f = blockIdx.y;
    if (band == 0) return;   // band 0 is consistent with the empty set for good
    if (spec_rounds_active(SB, f)) {
        if (SB.s_global) spec_validate_body<true>(band, f, ang_all, modgrad_all, cs_all, cs0_all, g, SB, round);
        else spec_validate_body<false>(band, f, ang_all, modgrad_all, cs_all, cs0_all, g, SB, round);
    }
    // The bookkeeping between two rounds, by the LAST band workgroup of the frame to get here (round 4; it was a launch of its own, 12 per call): if this round
    // changed nothing the frame has converged (sticky, the value is the round); otherwise the next round's counter starts at zero.  The state that decides
    // whether the prefix / validate launches of a round run is still only READ while that round runs -- it changes when all of the frame's bands are through.
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned prev = atomicInc(reinterpret_cast<unsigned *>(SB.round_state + (size_t)SB.frames_cap * 4 + f), (unsigned)SB.nbands - 2u);   // (wraps: bands 1 .. nbands - 1 count)
        if (prev == (unsigned)SB.nbands - 2u) {
            __threadfence();
            int *rs = SB.round_state + f * 4;
            if (!rs[3] && !rs[2]) {
                if (__hip_atomic_load(&rs[round & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) rs[2] = round;
                else rs[(round + 1) & 1] = 0;
            }
        }
    }
}

// everything the speculative schedule wants zeroed at the start of a call, in one launch (six memsets of ~5 us each before; one VGA frame in flight: 4.06 ms)
__global__ void __launch_bounds__(256) k_lsd_spec_clear(SpecBufs SB, int *__restrict__ rowcnt, int rounds_state)
{
    const int f = blockIdx.x, t = threadIdx.x;
    uint32_t *sm = SB.seedmap + (size_t)f * SB.bm_words;
    for (int i = t; i < SB.bm_words; i += 256) sm[i] = 0u;
    for (int i = t; i < 1024; i += 256) rowcnt[f * 1024 + i] = 0;
    for (int i = t; i < SB.nbands; i += 256) { SB.done[f * SB.nbands + i] = 0; SB.side[f * SB.nbands + i] = 0; }
    if (rounds_state && t < 4) SB.round_state[f * 4 + t] = 0;
    if (rounds_state && t == 4) SB.round_state[(size_t)SB.frames_cap * 4 + f] = 0;
}

// rectangles of a converged frame: the bands' records in order
__global__ void __launch_bounds__(64) k_lsd_spec_assemble(LsdRect *__restrict__ rects_all, int *__restrict__ nrect, int *__restrict__ status, LsdGeom g, SpecBufs SB, int last)
{
    const int band = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    if (!spec_rounds_converged(SB, f, last)) return;
    const size_t fb = (size_t)f * SB.nbands + band;
    int off = 0, total = 0;
    for (int b = lane; b < SB.nbands; b += 64) { const int n = SB.nrects[(size_t)f * SB.nbands + b]; total += n; if (b < band) off += n; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); total += __shfl_xor(total, o, 64); }
    LsdRect *rects = rects_all + (size_t)f * g.rect_cap;
    const SpecRec *recs = spec_log_recs(SB, fb);
    const int nrec = spec_log_cnt(SB, fb)[0];
    int k = off;
    for (int base = 0; base < nrec; base += 64) {
        const int i = base + lane;
        const bool hr = i < nrec && recs[i].has_rect != 0;
        const unsigned long long m = __ballot(hr);
        if (hr) {
            const int slot = k + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < g.rect_cap) rects[slot] = recs[i].rec;
        }
        k += __popcll(m);
    }
    if (band == SB.nbands - 1 && lane == 0) {
        nrect[f] = min(total, g.rect_cap);
        if (total > g.rect_cap) atomicOr(status, 1);
    }
}

#ifdef PLF_LSD_TIMING
extern "C" void plf_lsd_timing_dump()
{
    long long t[24];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_lsd_t), sizeof(t));
    printf("[lsd timing, frame 0 accumulated] grow %lld  rect %lld  refine %lld  total %lld cycles | regions %lld points %lld big %lld | iters %lld groups %lld accepts %lld\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]);
    printf("[lsd timing, band wave %d of frame 0 accumulated, cycles] region_grow %lld  log append %lld  region2rect %lld  refine %lld | warm-up phase %lld  own phase %lld\n", PLF_TIMING_BAND, t[10], t[11], t[12], t[13], t[14], t[15]);
    printf("[lsd timing] region_grow of frame 0: accept loops %lld cycles, exposed load wait %lld cycles | band wave: %lld, %lld\n", t[19], t[20], t[21], t[22]);
    printf("[lsd timing, band wave %d] seeds %lld  pixels grown %lld  per-seed record + rescan %lld cycles\n", PLF_TIMING_BAND, t[16], t[17], t[18]);
    printf("[lsd timing, frame 0 of the large-batch kernel] static singles marked without the pipeline %lld\n", t[23]);
}
#endif

// ------------------------------------------------------------------------------------------------
// NFA validation (one wave per rectangle)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double log_gamma_d(double x)
{
    if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
    const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
        a -= log(x + (double)n);
        b += q[n] * pow(x, (double)n);
    }
    return a + log(b);
}

__device__ __forceinline__ bool double_equal_d(double a, double b)
{
    if (a == b) return true;
    const double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    if (abs_max < 2.2250738585072014e-308) abs_max = 2.2250738585072014e-308;
    return (abs_diff / abs_max) <= (100.0 * 2.2204460492503131e-16);
}

// log_gamma(i) for integer i in [0, LGAM_N): filled once per handle by k_lsd_lgamma_table with the very same
// device function, so a lookup is bit-identical to evaluating it in place (3 calls x ~15 transcendentals saved
// per NFA evaluation).
#define LGAM_N 65536
__device__ const double *g_lgam_table = nullptr;

__global__ void k_lsd_lgamma_table(double *tab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < LGAM_N) tab[i] = i > 0 ? log_gamma_d((double)i) : 0.0;
}

__device__ __forceinline__ double log_gamma_int(const double *__restrict__ tab, int i)
{
    return (i > 0 && i < LGAM_N) ? tab[i] : log_gamma_d((double)i);
}

// The rest of the binomial tail cannot change the result any more: the ratio term(i+1) / term(i) = mult_term falls with i, so once it is below 1 every later term is
// smaller than this one, and a term below bin_tail * 2^-54 is less than half an ulp of bin_tail -- every remaining `bin_tail += term` returns bin_tail unchanged
// (round to nearest), and whichever way the loop ends it returns -log10(bin_tail) - LOG_NT of this very bin_tail.  (A rectangle of a few thousand pixels with more
// aligned pixels than n p -- any real edge -- spends nearly all of upstream's n / 2 - k iterations adding such terms.  If bin_tail * 2^-54 underflows the test
// never fires and the loop runs as upstream's.)
#define NFA_DEAD_TAIL(mult, term, bin_tail) ((mult) < 1.0 && (term) < (bin_tail) * 0x1p-54)

__device__ double nfa_d(const double *__restrict__ lgam, double LOG_NT, int n, int k, double p)
{
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - (double)n * log10(p);
    const double p_term = p / (1 - p);
    const double log1term = log_gamma_int(lgam, n + 1) - log_gamma_int(lgam, k + 1) - log_gamma_int(lgam, n - k + 1) +
                            (double)k * log(p) + (double)(n - k) * log(1.0 - p);
    double term = exp(log1term);
    if (double_equal_d(term, 0)) {
        if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - LOG_NT;
        return -LOG_NT;
    }
    double bin_tail = term;
    const double tolerance = 0.1;
    int i = k + 1;
    // While bin_term >= 1 (i.e. n - i + 1 >= i) the reference's loop has no exit test: such iterations are taken four at a time so that the four
    // divisions -- independent of the running product -- overlap; the product / sum chain itself is unchanged, operation for operation.
    while (i + 7 <= n && n - (i + 7) + 1 >= i + 7) {
        double b[8], m[8];
#pragma unroll
        for (int q = 0; q < 8; q++) b[q] = (double)(n - i - q + 1) / (double)(i + q);
#pragma unroll
        for (int q = 0; q < 8; q++) m[q] = b[q] * p_term;
#pragma unroll
        for (int q = 0; q < 8; q++) { term *= m[q]; bin_tail += term; }
        i += 8;
        if (NFA_DEAD_TAIL(m[7], term, bin_tail)) return -log10(bin_tail) - LOG_NT;
    }
    while (i + 3 <= n && n - (i + 3) + 1 >= i + 3) {
        const double b0 = (double)(n - i + 1) / (double)i, b1 = (double)(n - i) / (double)(i + 1), b2 = (double)(n - i - 1) / (double)(i + 2),
                     b3 = (double)(n - i - 2) / (double)(i + 3);
        const double m0 = b0 * p_term, m1 = b1 * p_term, m2 = b2 * p_term, m3 = b3 * p_term;
        term *= m0; bin_tail += term;
        term *= m1; bin_tail += term;
        term *= m2; bin_tail += term;
        term *= m3; bin_tail += term;
        i += 4;
        if (NFA_DEAD_TAIL(m3, term, bin_tail)) return -log10(bin_tail) - LOG_NT;
    }
    for (; i <= n; ++i) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - pow(mult_term, (double)(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * fabs(-log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - LOG_NT;
}

// NFA values of small rectangles, tabulated: rect_improve asks for nfa(n, k, p) tens of millions of times per large batch (9,600 values per VGA frame), nearly
// always with a pixel count n of a few dozen to a few hundred and always with p = 1/8 * 2^-j, j = 0..10 (region2rect's p, halved by the two precision stages).  The
// table holds nfa_d's own result for every n < NFA_TAB_N, k <= n and those eleven p -- filled by k_nfa_table with the very same device function (the lgamma table's
// argument: a lookup is bit-identical to evaluating in place), once per scaled-image size (LOG_NT enters the loop's exit test, so the values depend on it).
#define NFA_TAB_ROW (NFA_TAB_N * (NFA_TAB_N + 1) / 2)   // (n, k <= n) -> n (n + 1) / 2 + k
__global__ void __launch_bounds__(256) k_nfa_table(double *__restrict__ tab, const double *__restrict__ lgam, double log_nt)
{
    const int n = blockIdx.x, j = blockIdx.y;
    double p = 0.125;
    for (int q = 0; q < j; q++) p /= 2;
    for (int k = threadIdx.x; k <= n; k += 256) tab[(size_t)j * NFA_TAB_ROW + n * (n + 1) / 2 + k] = nfa_d(lgam, log_nt, n, k, p);
}

// true (and v) if (n, k, p) is tabulated
__device__ __forceinline__ bool nfa_lookup(const double *__restrict__ tab, int n, int k, double p, double &v)
{
    const long long bits = __double_as_longlong(p);
    const int j = 1023 - 3 - (int)(bits >> 52);   // p = 2^-(3 + j) exactly <=> mantissa 0, sign 0
    if ((bits & 0xFFFFFFFFFFFFFll) != 0 || j < 0 || j >= NFA_TAB_P || n < 0 || n >= NFA_TAB_N || k < 0 || k > n) return false;
    v = tab[(size_t)j * NFA_TAB_ROW + n * (n + 1) / 2 + k];
    return true;
}

struct EdgePt { int x, y, taken; };

// ------------------------------------------------------------------------------------------------
// NFA validation = rect_improve (imgproc/lsd.cpp), restructured for the GPU without changing a decision:
//   * the pixel COUNTING of a rectangle (rect_nfa's scan-line walk) is wave-parallel: upstream's left/right x
//     walk adds integer-valued steps (its slopes are integer divisions), so the span of row y has a closed form
//     and every lane takes a row; the counts for up to 6 angle precisions of the same geometry come from one pass;
//   * the NFA MATH (log-gamma, binomial tail; fp64 transcendentals) is lane-parallel: one lane per rectangle;
//   * rect_improve's 5 search stages stay sequential (stage s+1 starts from the best rectangle of stage s); inside
//     a stage the 5 candidate rectangles do not depend on the comparisons, so they are counted together and the
//     "keep it if better" chain is replayed in order by the lane that owns the rectangle.
// Work lists are global (all frames of the batch) and compacted with atomics; order is irrelevant because every
// result is written to its rectangle's own slot.
// ------------------------------------------------------------------------------------------------
struct NfaEntry { LsdRect r; int frame, nprec, pad0, pad1; };   // nprec: 0 = skip, 1 = r.prec only, 6 = r.prec and r.p/2^k, k=1..5
struct NfaCounts { int total, alg[6], pad; };
struct NfaState { LsdRect rec; double log_nfa; int frame, rect; };

#define NFA_U 4   // gathers in flight per lane in rect_count's column loop
// pixel count of one rectangle by a group of 16 lanes (4 rectangles per wave: most candidate rectangles span
// only a few rows, so a full wave per rectangle would idle)
// NP = 6: the counts for rec.prec and the five halved precisions of the same geometry (stages 0 and 4); NP = 1: rec.prec only (the candidate
// rectangles of stages 1..3 -- most of the pixel visits; the five unused comparisons per pixel were 37 % of the loop body)
__device__ __forceinline__ int div_small(int a, int b);

// scan-line description of a rectangle: everything rect_count's row walk needs (all integers)
struct RectScan { int mnx, y_lo, y_hi, lfy, rty, fl, sl, fr, sr; };

__device__ __forceinline__ void rect_scan_setup(const LsdRect &rec, int H, RectScan &S)
{
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    // the four corners, kept in registers (no indexed array: that would live in scratch memory)
    int px0 = (int)(rec.x1 - dyhw), py0 = (int)(rec.y1 + dxhw);
    int px1 = (int)(rec.x2 - dyhw), py1 = (int)(rec.y2 + dxhw);
    int px2 = (int)(rec.x2 + dyhw), py2 = (int)(rec.y2 - dxhw);
    int px3 = (int)(rec.x1 + dyhw), py3 = (int)(rec.y1 - dxhw);
    // std::sort with AsmallerB_XoverY is a total order on (x, y): a sorting network yields the same sequence
#define PLF_CSWAP(xa, ya, xb, yb) { const bool sw = (xa > xb) || (xa == xb && ya > yb); const int tx = sw ? xb : xa, ty = sw ? yb : ya; \
                                    xb = sw ? xa : xb; yb = sw ? ya : yb; xa = tx; ya = ty; }
    PLF_CSWAP(px0, py0, px1, py1) PLF_CSWAP(px2, py2, px3, py3) PLF_CSWAP(px0, py0, px2, py2) PLF_CSWAP(px1, py1, px3, py3) PLF_CSWAP(px1, py1, px2, py2)
#undef PLF_CSWAP
#define PLF_SELX(i) ((i) == 0 ? px0 : (i) == 1 ? px1 : (i) == 2 ? px2 : px3)
#define PLF_SELY(i) ((i) == 0 ? py0 : (i) == 1 ? py1 : (i) == 2 ? py2 : py3)
    int imin = 0, imax = 0;   // first minimum / first maximum of y in sorted order (strict comparisons)
    {
        int ymin = py0, ymax = py0;
        if (ymin > py1) { imin = 1; ymin = py1; }
        if (ymin > py2) { imin = 2; ymin = py2; }
        if (ymin > py3) { imin = 3; ymin = py3; }
        if (ymax < py1) { imax = 1; ymax = py1; }
        if (ymax < py2) { imax = 2; ymax = py2; }
        if (ymax < py3) { imax = 3; ymax = py3; }
    }
    // only `min` is marked taken before the next picks (upstream); the list is sorted by x, so the leftmost of the
    // rest is the lowest free index, the rightmost is the later of the remaining two unless their x are equal
    const int il = (imin == 0) ? 1 : 0;
    int ra = -1, rb = -1;
    for (int i = 0; i < 4; ++i) if (i != imin && i != il) { if (ra < 0) ra = i; else rb = i; }
    const int ir = (PLF_SELX(ra) < PLF_SELX(rb)) ? rb : ra;
    const int it = (ir == ra) ? rb : ra;
    EdgePt mn, mx, lf, rt, tl;
    mn.x = PLF_SELX(imin); mn.y = PLF_SELY(imin); mx.x = PLF_SELX(imax); mx.y = PLF_SELY(imax);
    lf.x = PLF_SELX(il); lf.y = PLF_SELY(il); rt.x = PLF_SELX(ir); rt.y = PLF_SELY(ir); tl.x = PLF_SELX(it); tl.y = PLF_SELY(it);
#undef PLF_SELX
#undef PLF_SELY
    // upstream: integer divisions, and `tailp->p.x` where p.y was meant
    S.fl = (mn.y != lf.y) ? div_small(mn.x - lf.x, mn.y - lf.y) : 0;
    S.sl = (lf.y != tl.x) ? div_small(lf.x - tl.x, lf.y - tl.x) : 0;
    S.fr = (mn.y != rt.y) ? div_small(mn.x - rt.x, mn.y - rt.y) : 0;
    S.sr = (rt.y != tl.x) ? div_small(rt.x - tl.x, rt.y - tl.x) : 0;
    // rows outside the image are skipped BEFORE the step update (upstream `continue`): the walk starts at y_lo.
    S.mnx = mn.x; S.lfy = lf.y; S.rty = rt.y;
    S.y_lo = max(mn.y, 0); S.y_hi = min(mx.y, H - 1);
}

// span of row y (y_lo <= y <= y_hi).  After visiting row y' the walk adds (y' >= lf.y ? slstep : flstep); all terms are integers, so the span of
// row y is exact in closed form.
__device__ __forceinline__ void rect_span(const RectScan &S, int y, int W, int &xl, int &xr)
{
    // (32-bit: slopes and row counts are below 2^15 for any image the handle accepts -- line_configure checks -- so the products stay below 2^30 and the 24-bit
    // multiplier's result is the full product; upstream's own arithmetic is int)
    const int al = max(0, min(y, S.lfy) - S.y_lo), bl = (y - S.y_lo) - al;
    const int ar = max(0, min(y, S.rty) - S.y_lo), br = (y - S.y_lo) - ar;
    const int left = S.mnx + __mul24(S.fl, al) + __mul24(S.sl, bl), right = S.mnx + __mul24(S.fr, ar) + __mul24(S.sr, br);
    xl = max(left, 0); xr = min(right, W - 1);
}

// truncating integer division (C semantics) of operands below 2^23 in magnitude, b != 0: the float quotient estimate is within 1 of the truth, fixed up exactly
// with the remainder (the compiler's general 32-bit division is ~35 instructions; rect_scan_setup has four)
__device__ __forceinline__ int div_small(int a, int b)
{
    const unsigned ua = (unsigned)abs(a), ub = (unsigned)abs(b);
    unsigned q = (unsigned)((float)ua * __builtin_amdgcn_rcpf((float)ub));
    int r = (int)ua - (int)(q * ub);
    if (r < 0) { q--; r += (int)ub; }
    if (r >= (int)ub) { q++; }
    return ((a ^ b) < 0) ? -(int)q : (int)q;
}

// sum over the 16 lanes of a DPP row (= one rectangle's group), result in every lane: four rotate-and-add steps, one instruction each (ds_bpermute + address
// arithmetic before)
__device__ __forceinline__ int row16_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);   // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);   // row_ror:1
    return v;
}

// |level-line angle - theta| folded as upstream's isAligned does
__device__ __forceinline__ double rect_ntheta(float dw, double theta)
{
    const double a = (double)fabsf(dw) * DEG2RAD_D;   // the sign bit is k_lsd_regions' USED flag
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI_D) {
        n_theta -= M_2__PI_D;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta;
}

template <int NP, int G = 16>
__device__ void rect_count(const float *__restrict__ ang, int W, int H, const LsdRect &rec, NfaCounts &out)
{
    const int lane = plf_lane() & (G - 1);   // G lanes per rectangle: 16 (four rectangles per wave, large batches) or the whole wave (k_nfa_fused)
    RectScan S;
    rect_scan_setup(rec, H, S);
    const int y_lo = S.y_lo, y_hi = S.y_hi;
    double precs[NP];
    precs[0] = rec.prec;
    if (NP > 1) {
        double pp = rec.p;
#pragma unroll
        for (int k = 1; k < NP; k++) { pp /= 2; precs[k] = pp * PI_D; }
    }
    int total = 0, alg[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) alg[k] = 0;
    // 16 lanes = ry_n rows x rx_n interleaved columns: rectangles along the x axis have few, long rows
    int ry_n = G;
    while (ry_n > 1 && ry_n > y_hi - y_lo + 1) ry_n >>= 1;
    const int rx_n = G / ry_n, ry = lane & (ry_n - 1), rx = lane / ry_n;
    for (int y = y_lo + ry; y <= y_hi; y += ry_n) {
        int xl, xr;
        rect_span(S, y, W, xl, xr);
        const float *row = ang + (size_t)y * W;
        // the angle words of a row sit in HBM (thousands of frames in flight: no cache holds them) and the kernel is bound by that latency: NFA_U gathers
        // in flight per lane instead of 1, and 16 waves per SIMD-quad slot (line_host.hip) -- 15.9 -> 7.9 ms per 4096 frames for the five stages
        for (int x = xl + rx; x <= xr; x += NFA_U * rx_n) {
            float dwv[NFA_U];
#pragma unroll
            for (int q = 0; q < NFA_U; q++) dwv[q] = (x + q * rx_n <= xr) ? row[x + q * rx_n] : NOTDEF_F;
#pragma unroll
            for (int q = 0; q < NFA_U; q++) {
                if (x + q * rx_n > xr) continue;
                ++total;
                const float dw = dwv[q];
                if (dw == NOTDEF_F) continue;
                const double n_theta = rect_ntheta(dw, rec.theta);
#pragma unroll
                for (int k = 0; k < NP; k++) if (n_theta <= precs[k]) ++alg[k];
            }
        }

    }
    if (G == 16) {
        total = row16_sum(total);
#pragma unroll
        for (int k = 0; k < NP; k++) alg[k] = row16_sum(alg[k]);
    } else {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {  // butterfly inside the group
            total += __shfl_xor(total, o, 64);
#pragma unroll
            for (int k = 0; k < NP; k++) alg[k] += __shfl_xor(alg[k], o, 64);
        }
    }
    out.total = total;
#pragma unroll
    for (int k = 0; k < 6; k++) out.alg[k] = k < NP ? alg[k < NP ? k : 0] : 0;
}

// The five candidate rectangles of one width stage (1..3) of rect_improve, counted in ONE pass by a group of 16 lanes.  They share theta and the precision and
// cover nearly the same pixels, so (i) lane c < 5 does candidate c's scan-line set-up (corners, ordering, the four integer slopes -- two thirds of the instructions
// of a rect_count call on the small rectangles that make up most of a frame) while the others idle instead of the whole group doing it five times, (ii) every
// pixel of the union of the spans is fetched and its angle folded once, then tested against each candidate's span.  Per candidate the counted set is exactly
// rect_count<1>'s: the pixels of rows y_lo..y_hi inside [xl, xr] of rect_span.  par: 5 x 12 ints of LDS owned by the group.
__device__ void rect_count5(const float *__restrict__ ang, int W, int H, const LsdRect &cand, bool valid, double theta, double prec, int *par, int (&total)[5],
                            int (&alg)[5])
{
    const int lane = plf_lane() & 15;
    RectScan S;
    rect_scan_setup(cand, H, S);
    const bool mine = lane < 5 && valid;
    if (lane < 5) {
        int *P = par + lane * 12;
        P[0] = S.mnx; P[1] = mine ? S.y_lo : 1; P[2] = mine ? S.y_hi : 0; P[3] = S.lfy; P[4] = S.rty; P[5] = S.fl; P[6] = S.sl; P[7] = S.fr; P[8] = S.sr;
    }
    int y_lo = mine ? S.y_lo : (1 << 30), y_hi = mine ? S.y_hi : -(1 << 30);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { y_lo = min(y_lo, __shfl_xor(y_lo, o, 64)); y_hi = max(y_hi, __shfl_xor(y_hi, o, 64)); }
    CBAR();
#pragma unroll
    for (int c = 0; c < 5; c++) total[c] = alg[c] = 0;
    int ry_n = 16;
    while (ry_n > 1 && ry_n > y_hi - y_lo + 1) ry_n >>= 1;
    const int rx_n = 16 / ry_n, ry = lane & (ry_n - 1), rx = lane / ry_n;
    for (int y = y_lo + ry; y <= y_hi; y += ry_n) {
        int xl[5], xr[5], uxl = 1 << 30, uxr = -1;   // (a row none of the candidates has pixels in: an empty loop below, without overflowing uxl + rx)
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const int *P = par + c * 12;
            RectScan Sc;
            Sc.mnx = P[0]; Sc.y_lo = P[1]; Sc.y_hi = P[2]; Sc.lfy = P[3]; Sc.rty = P[4]; Sc.fl = P[5]; Sc.sl = P[6]; Sc.fr = P[7]; Sc.sr = P[8];
            rect_span(Sc, y, W, xl[c], xr[c]);
            if (y < Sc.y_lo || y > Sc.y_hi) { xl[c] = 1; xr[c] = 0; }
            if (xl[c] <= xr[c]) { uxl = min(uxl, xl[c]); uxr = max(uxr, xr[c]); }
        }
        const float *row = ang + (size_t)y * W;
        for (int x = uxl + rx; x <= uxr; x += NFA_U * rx_n) {
            float dwv[NFA_U];
#pragma unroll
            for (int q = 0; q < NFA_U; q++) dwv[q] = (x + q * rx_n <= uxr) ? row[x + q * rx_n] : NOTDEF_F;
#pragma unroll
            for (int q = 0; q < NFA_U; q++) {
                const int xq = x + q * rx_n;
                if (xq > uxr) continue;
                const float dw = dwv[q];
                const bool al = dw != NOTDEF_F && rect_ntheta(dw, theta) <= prec;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const bool in = xq >= xl[c] && xq <= xr[c];
                    total[c] += in ? 1 : 0;
                    alg[c] += (in && al) ? 1 : 0;
                }
            }
        }
    }
    CBAR();
#pragma unroll
    for (int c = 0; c < 5; c++) { total[c] = row16_sum(total[c]); alg[c] = row16_sum(alg[c]); }
}

__device__ __forceinline__ void emit_segment(LsdRect rec, float4 *seg)
{
    rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
    rec.x1 /= 0.8; rec.y1 /= 0.8; rec.x2 /= 0.8; rec.y2 /= 0.8;
    *seg = make_float4((float)rec.x1, (float)rec.y1, (float)rec.x2, (float)rec.y2);
}

// one lane per rectangle of every frame: queue it for the first count
__global__ void __launch_bounds__(256) k_nfa_init(const LsdRect *__restrict__ rects_all, const int *__restrict__ nrect,
                                                  uint8_t *__restrict__ keep_all, NfaEntry *__restrict__ entries, NfaState *__restrict__ states,
                                                  int *__restrict__ counters, int *__restrict__ status, LsdGeom g)
{
    const int f = blockIdx.y, n = nrect[f];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        keep_all[(size_t)f * g.rect_cap + i] = 0;
        const int q = atomicAdd(&counters[0], 1);
        if (q >= g.nfa_pool) { atomicOr(status, 1); continue; }   // the batch holds more rectangles than the pooled stage buffers (k_nfa_clamp trims the count)
        NfaEntry e;
        e.r = rects_all[(size_t)f * g.rect_cap + i]; e.frame = f; e.nprec = 6; e.pad0 = e.pad1 = 0;
        entries[q] = e;
        NfaState st;
        st.rec = e.r; st.log_nfa = -1; st.frame = f; st.rect = i;
        states[q] = st;
    }
}

__global__ void k_nfa_clamp(int *__restrict__ counters, int *__restrict__ status, LsdGeom g)
{
    if (counters[0] > g.nfa_pool) { counters[0] = g.nfa_pool; atomicOr(status, 1); }
}

// persistent waves: entry e -> counts[e];  n = counters[cidx] * mult.  k_nfa_count: stages 0 and 4 (entries carry nprec 6 or 0), k_nfa_count1:
// stages 1..3 (nprec 1 or 0)
template <int NP>
__device__ __forceinline__ void nfa_count_body(const float *__restrict__ ang_all, const NfaEntry *__restrict__ entries, const int *__restrict__ counters,
                                               int cidx, int mult, NfaCounts *__restrict__ counts, const LsdGeom &g)
{
    const int n = counters[cidx] * mult;
    const int grp = threadIdx.x >> 4;
    for (int e0 = blockIdx.x * 4; e0 < n; e0 += gridDim.x * 4) {
        const int e = e0 + grp;
        if (e >= n) continue;     // (no wave-wide barrier below: groups are independent)
        const NfaEntry en = entries[e];
        if (en.nprec == 0) continue;
        NfaCounts c;
        if (en.nprec == NP) rect_count<NP>(ang_all + (size_t)en.frame * g.s_stride, g.sw, g.sh, en.r, c);
        else if (NP == 1) rect_count<6>(ang_all + (size_t)en.frame * g.s_stride, g.sw, g.sh, en.r, c);   // (not produced by k_nfa_math; kept for safety)
        else rect_count<1>(ang_all + (size_t)en.frame * g.s_stride, g.sw, g.sh, en.r, c);
        if ((threadIdx.x & 15) == 0) counts[e] = c;
    }
}
__global__ void __launch_bounds__(64) k_nfa_count(const float *__restrict__ ang_all, const NfaEntry *__restrict__ entries,
                                                  const int *__restrict__ counters, int cidx, int mult, NfaCounts *__restrict__ counts, LsdGeom g)
{
    nfa_count_body<6>(ang_all, entries, counters, cidx, mult, counts, g);
}
__global__ void __launch_bounds__(64) k_nfa_count1(const float *__restrict__ ang_all, const NfaEntry *__restrict__ entries,
                                                   const int *__restrict__ counters, int cidx, int mult, NfaCounts *__restrict__ counts, LsdGeom g)
{
    nfa_count_body<1>(ang_all, entries, counters, cidx, mult, counts, g);
}

// the same with a whole wave per rectangle: what is left for the staged kernels after k_nfa_small are the rectangles of 512 pixels and more (16 lanes would walk
// 32+ pixels each, one dependent gather after the other)
template <int NP>
__device__ __forceinline__ void nfa_count_wave_body(const float *__restrict__ ang_all, const NfaEntry *__restrict__ entries, const int *__restrict__ counters,
                                                    int cidx, int mult, NfaCounts *__restrict__ counts, const LsdGeom &g)
{
    const int n = counters[cidx] * mult;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const NfaEntry en = entries[e];
        if (en.nprec == 0) continue;
        NfaCounts c;
        if (en.nprec == NP) rect_count<NP, 64>(ang_all + (size_t)en.frame * g.s_stride, g.sw, g.sh, en.r, c);
        else if (NP == 1) rect_count<6, 64>(ang_all + (size_t)en.frame * g.s_stride, g.sw, g.sh, en.r, c);
        else rect_count<1, 64>(ang_all + (size_t)en.frame * g.s_stride, g.sw, g.sh, en.r, c);
        if (threadIdx.x == 0) counts[e] = c;
    }
}
__global__ void __launch_bounds__(64) k_nfa_count_w(const float *__restrict__ ang_all, const NfaEntry *__restrict__ entries,
                                                    const int *__restrict__ counters, int cidx, int mult, NfaCounts *__restrict__ counts, LsdGeom g)
{
    PLF_LINE_SETPRIO();
    nfa_count_wave_body<6>(ang_all, entries, counters, cidx, mult, counts, g);
}
__global__ void __launch_bounds__(64) k_nfa_count1_w(const float *__restrict__ ang_all, const NfaEntry *__restrict__ entries,
                                                     const int *__restrict__ counters, int cidx, int mult, NfaCounts *__restrict__ counts, LsdGeom g)
{
    PLF_LINE_SETPRIO();
    nfa_count_wave_body<1>(ang_all, entries, counters, cidx, mult, counts, g);
}

__device__ __forceinline__ void nfa_finish(const NfaState &st, float4 *__restrict__ seg_all, uint8_t *__restrict__ keep_all, const LsdGeom &g)
{
    keep_all[(size_t)st.frame * g.rect_cap + st.rect] = 1;
    emit_segment(st.rec, &seg_all[(size_t)st.frame * g.rect_cap + st.rect]);
}

// NFA value of every counted candidate, one lane per (rectangle, candidate): the fp64-transcendental part is
// spread over as many lanes as there are candidates so that no lane evaluates more than one binomial tail.
// stage 0 / 4: item = rectangle * 6 + k (k-th precision of the same geometry); stages 1..3: item = entry index.
// The length of the binomial-tail loop varies from 1 to thousands of iterations between items (it runs from k + 1
// to about n / 2), so a block first orders its chunk of items by the predicted loop length (LDS counting sort on
// half-octave buckets): the 64 lanes of a wave then run loops of similar length instead of idling behind the
// longest one.  The order only decides which lane evaluates which item.
#define EV_T 256
#define EV_PER 8
#define EV_CHUNK (EV_T * EV_PER)
__device__ __forceinline__ bool nfa_item(int stage, bool multi, int it, const NfaEntry *__restrict__ entries, const NfaCounts *__restrict__ counts,
                                         int &n, int &k, double &p)
{
    if (multi) {
        const int e = it / 6, q = it - e * 6;
        if (entries[e].nprec != 6 || (stage == 4 && q == 0)) return false;
        double pp = entries[e].r.p;
        for (int j = 0; j < q; j++) pp /= 2;
        n = counts[e].total; k = counts[e].alg[q]; p = pp;   // (indexed straight from memory: no by-value struct in scratch)
        return true;
    }
    if (entries[it].nprec == 0) return false;
    n = counts[it].total; k = counts[it].alg[0]; p = entries[it].r.p;
    return true;
}

__global__ void __launch_bounds__(EV_T) k_nfa_eval(int stage, const double *__restrict__ lgam, const double *__restrict__ tab, const NfaCounts *__restrict__ counts,
                                                   const NfaEntry *__restrict__ entries, const int *__restrict__ counters,
                                                   double *__restrict__ vals, LsdGeom g)
{
    PLF_LINE_SETPRIO();
    __shared__ uint16_t order[EV_CHUNK];
    __shared__ int hist[32];
    const bool multi = (stage == 0 || stage == 4);
    const int total = counters[stage] * (multi ? 6 : 5), t = threadIdx.x;
    // items per thread: EV_PER when the grid is full; fewer when there is little work (few frames in flight), so that it spreads over more blocks
    // and no thread evaluates several binomial tails back to back
    const int per = min(EV_PER, max(1, (total + (int)gridDim.x * EV_T - 1) / ((int)gridDim.x * EV_T))), chunk = per * EV_T;
    for (int c0 = blockIdx.x * chunk; c0 < total; c0 += gridDim.x * chunk) {
        if (t < 32) hist[t] = 0;
        __syncthreads();
        int bkt[EV_PER], rnk[EV_PER];
#pragma unroll
        for (int q = 0; q < EV_PER; q++) {
            const int it = c0 + q * EV_T + t;
            bkt[q] = -1; rnk[q] = 0;
            if (q < per && it < total) {
                int n = 0, k = 0;
                double p;
                int b = 0;
                double tv;
                if (nfa_item(stage, multi, it, entries, counts, n, k, p) && n != 0 && k != 0 && n != k && !(tab && nfa_lookup(tab, n, k, p, tv))) {
                    const int L = max(1, (n + 1) / 2 - k);
                    const int lz = 31 - __clz(L);
                    b = min(31, 1 + 2 * lz + (lz > 0 ? ((L >> (lz - 1)) & 1) : 0));
                }
                bkt[q] = b;
                rnk[q] = atomicAdd(&hist[b], 1);
            }
        }
        __syncthreads();
        if (t == 0) {   // longest loops first
            int run = 0;
            for (int b = 31; b >= 0; b--) { const int v = hist[b]; hist[b] = run; run += v; }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EV_PER; q++)
            if (bkt[q] >= 0) order[hist[bkt[q]] + rnk[q]] = (uint16_t)(q * EV_T + t);
        __syncthreads();
        const int cnt = min(chunk, total - c0);
        for (int i = t; i < cnt; i += EV_T) {
            const int it = c0 + order[i];
            int n = 0, k = 0;
            double p = 0.0, v = -1.0e300;
            if (nfa_item(stage, multi, it, entries, counts, n, k, p) && !(tab && nfa_lookup(tab, n, k, p, v))) v = nfa_d(lgam, g.log_nt, n, k, p);
            vals[it] = v;
        }
        __syncthreads();
    }
}

// stage: 0 = first evaluation + "finer precision" loop, 1..3 = the three width loops, 4 = final precision loop.
// Reads the NFA values of the previous eval pass, replays the "keep if better" chain in the reference order,
// finishes meaningful rectangles and queues the next stage's candidate rectangles for the others.
__global__ void __launch_bounds__(64) k_nfa_math(int stage, const double *__restrict__ vals, const NfaEntry *__restrict__ entries,
                                                 const NfaState *__restrict__ st_in, NfaState *__restrict__ st_out,
                                                 NfaEntry *__restrict__ ent_out, int *__restrict__ counters, float4 *__restrict__ seg_all,
                                                 uint8_t *__restrict__ keep_all, LsdGeom g)
{
    PLF_LINE_SETPRIO();
    const int n = counters[stage];
    const double LOG_EPS = 0.0, delta = 0.5, delta_2 = delta / 2.0;
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) {
        NfaState st = st_in[i];
        if (stage == 0 || stage == 4) {
            const NfaEntry en = entries[i];
            if (stage == 0) {
                st.log_nfa = vals[6 * i];
                if (st.log_nfa > LOG_EPS) { nfa_finish(st, seg_all, keep_all, g); continue; }
            }
            if (en.nprec == 6) {
                LsdRect r = st.rec;
                for (int k = 1; k <= 5; ++k) {
                    r.p /= 2;
                    r.prec = r.p * PI_D;
                    const double v = vals[6 * i + k];
                    if (v > st.log_nfa) { st.log_nfa = v; st.rec = r; }
                }
            }
            if (st.log_nfa > LOG_EPS) { nfa_finish(st, seg_all, keep_all, g); continue; }
            if (stage == 4) continue;  // not meaningful
        } else {
            for (int k = 0; k < 5; ++k) {
                const NfaEntry en = entries[5 * i + k];
                if (en.nprec == 0) continue;
                const double v = vals[5 * i + k];
                if (v > st.log_nfa) { st.rec = en.r; st.log_nfa = v; }
            }
            if (st.log_nfa > LOG_EPS) { nfa_finish(st, seg_all, keep_all, g); continue; }
        }
        // queue the next stage
        const int q = atomicAdd(&counters[stage + 1], 1);
        st_out[q] = st;
        LsdRect r = st.rec;
        if (stage <= 2) {
            // stage 0 -> "reduce width", 1 -> "reduce one side", 2 -> "reduce the other side"
            for (int k = 0; k < 5; ++k) {
                NfaEntry e;
                e.frame = st.frame; e.pad0 = e.pad1 = 0; e.nprec = 0;
                if ((r.width - delta) >= 0.5) {
                    if (stage == 1) { r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; }
                    if (stage == 2) { r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; }
                    r.width -= delta;
                    e.nprec = 1;
                }
                e.r = r;
                ent_out[5 * q + k] = e;
            }
        } else {  // stage 3 -> final "finer precision" loop (upstream keeps the width test here too)
            NfaEntry e;
            e.r = r; e.frame = st.frame; e.pad0 = e.pad1 = 0;
            e.nprec = ((r.width - delta) >= 0.5) ? 6 : 0;
            ent_out[q] = e;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rect_improve of one SMALL rectangle by a group of 16 lanes, all five stages in one launch (large batches).  Two thirds of a frame's ~550 rectangles are small
// regions that fail the first test and walk through all five stages (22 pixel counts, 21 NFA values each) only to be rejected; with the staged kernels above that is
// 17 grid-wide launches whose eval / math steps mostly move 112-byte work-list entries around.  Here every NFA value is one load from k_nfa_table's table and the
// "keep it if better" chain stays in registers (the same chain as k_nfa_math, the same rect_count).  A rectangle with a candidate the table does not hold (512 pixels
// or more) is handed, untouched, to the staged kernels through their stage-0 work list, exactly as k_nfa_init would have queued it.
// ------------------------------------------------------------------------------------------------
#ifndef PLF_NFA_SMALL_WPE
#define PLF_NFA_SMALL_WPE 0
#endif
#if PLF_NFA_SMALL_WPE > 0
#define PLF_NFA_SMALL_OCC __attribute__((amdgpu_waves_per_eu(PLF_NFA_SMALL_WPE, PLF_NFA_SMALL_WPE)))
#else
#define PLF_NFA_SMALL_OCC
#endif
// stages LO..HI of one rectangle by its group of 16 lanes (all lanes of the group return the same state)
template <int LO, int HI>
__device__ __forceinline__ void nfa_small_stages(const float *__restrict__ ang, const double *__restrict__ tab, const LsdGeom &g, int *par, int grp, int lane16,
                                                 LsdRect &rec, double &log_nfa, bool &keep, bool &defer, bool &fin)
{
    const double LOG_EPS = 0.0, delta = 0.5, delta_2 = delta / 2.0;
    NfaCounts c;
    for (int stage = LO; stage <= HI && !fin; stage++) {
        if (stage == 0 || stage == 4) {
            if (stage == 4 && !((rec.width - delta) >= 0.5)) break;   // (nprec 0 in k_nfa_math: nothing to evaluate, not meaningful)
            rect_count<6, 16>(ang, g.sw, g.sh, rec, c);
            LsdRect r = rec;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (k > 0) { r.p /= 2; r.prec = r.p * PI_D; }
                if (stage == 4 && k == 0) continue;
                double v;
                if (!nfa_lookup(tab, c.total, c.alg[k], r.p, v)) { defer = fin = true; break; }
                if (stage == 0 && k == 0) {
                    log_nfa = v;
                    if (v > LOG_EPS) { keep = fin = true; break; }
                } else if (v > log_nfa) { log_nfa = v; rec.p = r.p; rec.prec = r.prec; }
            }
        } else {
            // 1: reduce the width, 2: reduce one side, 3: reduce the other side -- five candidates, each derived from the previous one: lane c of the group
            // builds candidate c by the same c + 1 steps (lanes 5..15 idle along with candidate 4), then one pass counts all five
            LsdRect r = rec;
            bool valid = true;
            const int myc = min(lane16, 4);
            for (int k = 0; k <= 4; k++) {
                if (k > myc) break;
                if (!((r.width - delta) >= 0.5)) { valid = false; break; }   // (every later candidate fails the same test)
                if (stage == 2) { r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; }
                if (stage == 3) { r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; }
                r.width -= delta;
            }
            const int nvalid = __popcll((__ballot(valid && lane16 < 5) >> (16 * grp)) & 31ull);   // (monotone: candidates 0 .. nvalid - 1)
            if (nvalid > 0) {
                int total[5], alg[5];
                rect_count5(ang, g.sw, g.sh, r, valid, rec.theta, rec.prec, par, total, alg);
                int best = -1;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    if (k >= nvalid || fin) continue;
                    double v;
                    if (!nfa_lookup(tab, total[k], alg[k], rec.p, v)) { defer = fin = true; continue; }
                    if (v > log_nfa) { log_nfa = v; best = k; }
                }
                if (best >= 0 && !defer) {
                    const int src = (threadIdx.x & 48) + best;
                    rec.x1 = shfl_d(r.x1, src); rec.y1 = shfl_d(r.y1, src); rec.x2 = shfl_d(r.x2, src); rec.y2 = shfl_d(r.y2, src); rec.width = shfl_d(r.width, src);
                }
            }
        }
        if (!fin && log_nfa > LOG_EPS) keep = fin = true;
    }
}

// what a group's first lane does when its rectangle is decided: keep flag and segment, or the hand-over to the staged kernels (the rectangle as found)
__device__ __forceinline__ void nfa_small_finish(int f, int ri, const LsdRect &rec, bool keep, bool defer, const LsdRect *__restrict__ rects_all, uint8_t *__restrict__ keep_all,
                                                 float4 *__restrict__ seg_all, NfaEntry *__restrict__ entries, NfaState *__restrict__ states, int *__restrict__ counters,
                                                 int *__restrict__ status, const LsdGeom &g)
{
    const size_t o = (size_t)f * g.rect_cap + ri;
    keep_all[o] = keep && !defer ? 1 : 0;
    if (defer) {
        const int q = atomicAdd(&counters[0], 1);
        if (q >= g.nfa_pool) { atomicOr(status, 1); return; }
        NfaEntry e;
        e.r = rects_all[o]; e.frame = f; e.nprec = 6; e.pad0 = e.pad1 = 0;
        entries[q] = e;
        NfaState st;
        st.rec = e.r; st.log_nfa = -1; st.frame = f; st.rect = ri;
        states[q] = st;
    } else if (keep) emit_segment(rec, &seg_all[o]);
}

// surv != null (large batches): the rectangles that are not decided by stage 0 -- two thirds -- are queued per FRAME (surv[f * scap ..], fcnt[f]) for
// k_nfa_small2 instead of walking through stages 1-4 here: the four groups of a wave then always have work (a wave of this kernel lasted as long as its longest
// rectangle: 30 % of the group slots idled behind rectangles that stage 0 had already decided).  A full list is no error: the group carries on here.  (One counter
// per frame, not per XCD: 350 k atomics on one address cost more than the stages they were to save.)
__global__ void PLF_NFA_SMALL_OCC __launch_bounds__(64) k_nfa_small(const float *__restrict__ ang_all, const double *__restrict__ tab, const LsdRect *__restrict__ rects_all,
                                                  const int *__restrict__ nrect, uint8_t *__restrict__ keep_all, float4 *__restrict__ seg_all,
                                                  NfaEntry *__restrict__ entries, NfaState *__restrict__ states, int *__restrict__ counters,
                                                  int *__restrict__ status, LsdGeom g, int nframes, NfaState *__restrict__ surv, int *__restrict__ fcnt, int scap)
{
    PLF_LINE_SETPRIO();
    // grid (8 * slots, ceil(B / 8)): workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), so XCD x works through frame 8 * blockIdx.y + x and the
    // angle words of a frame are pulled into ONE L2 (k_orient_brief's order)
    const int f = 8 * (int)blockIdx.y + ((int)blockIdx.x & 7), slot = (int)blockIdx.x >> 3, nslots = (int)gridDim.x >> 3;
    if (f >= nframes) return;
    __shared__ int s_par[4][5 * 12];
    const int n_r = nrect[f], grp = threadIdx.x >> 4, lane16 = threadIdx.x & 15;
    const float *ang = ang_all + (size_t)f * g.s_stride;
    for (int i0 = slot * 4; i0 < n_r; i0 += nslots * 4) {
        const int ri = i0 + grp;
        if (ri >= n_r) continue;   // (groups are independent: rect_count's butterflies stay inside the 16 lanes)
        LsdRect rec = rects_all[(size_t)f * g.rect_cap + ri];
        double log_nfa = -1;
        bool keep = false, defer = false, fin = false;
        nfa_small_stages<0, 0>(ang, tab, g, &s_par[grp][0], grp, lane16, rec, log_nfa, keep, defer, fin);
        if (!fin) {
            int q = -1;
            if (surv) {
                if (lane16 == 0) q = atomicAdd(&fcnt[f], 1);
                q = __shfl(q, threadIdx.x & 48, 64);
            }
            if (q >= 0 && q < scap) {
                if (lane16 == 0) {
                    NfaState st;
                    st.rec = rec; st.log_nfa = log_nfa; st.frame = f; st.rect = ri;
                    surv[(size_t)f * scap + q] = st;
                }
                continue;
            }
            nfa_small_stages<1, 4>(ang, tab, g, &s_par[grp][0], grp, lane16, rec, log_nfa, keep, defer, fin);
        }
        if (lane16 == 0) nfa_small_finish(f, ri, rec, keep, defer, rects_all, keep_all, seg_all, entries, states, counters, status, g);
    }
}

// stages 1-4 of the rectangles k_nfa_small queued, same (frame, slot) grid: XCD x works through the frames 8 * blockIdx.y + x
__global__ void PLF_NFA_SMALL_OCC __launch_bounds__(64) k_nfa_small2(const float *__restrict__ ang_all, const double *__restrict__ tab, const LsdRect *__restrict__ rects_all,
                                                   uint8_t *__restrict__ keep_all, float4 *__restrict__ seg_all, NfaEntry *__restrict__ entries,
                                                   NfaState *__restrict__ states, int *__restrict__ counters, int *__restrict__ status, LsdGeom g, int nframes,
                                                   const NfaState *__restrict__ surv, const int *__restrict__ fcnt, int scap)
{
    PLF_LINE_SETPRIO();
    const int f = 8 * (int)blockIdx.y + ((int)blockIdx.x & 7), slot = (int)blockIdx.x >> 3, nslots = (int)gridDim.x >> 3;
    if (f >= nframes) return;
    __shared__ int s_par[4][5 * 12];
    const int n = min(fcnt[f], scap), grp = threadIdx.x >> 4, lane16 = threadIdx.x & 15;
    const float *ang = ang_all + (size_t)f * g.s_stride;
    for (int i0 = slot * 4; i0 < n; i0 += nslots * 4) {
        const int i = i0 + grp;
        if (i >= n) continue;
        const NfaState st = surv[(size_t)f * scap + i];
        LsdRect rec = st.rec;
        double log_nfa = st.log_nfa;
        bool keep = false, defer = false, fin = false;
        nfa_small_stages<1, 4>(ang, tab, g, &s_par[grp][0], grp, lane16, rec, log_nfa, keep, defer, fin);
        if (lane16 == 0) nfa_small_finish(f, st.rect, rec, keep, defer, rects_all, keep_all, seg_all, entries, states, counters, status, g);
    }
}

// ------------------------------------------------------------------------------------------------
// rect_improve of ONE rectangle by ONE wave, all five stages in one launch (few frames in flight).  The staged kernels above put a grid-wide barrier after
// every count / eval / math step, so a stage lasts as long as its slowest rectangle: with one frame in flight the 17 launches took 1.2-2.4 ms, most of it the
// binomial tails of a few rectangles per stage.  Here a rectangle's chain only waits for itself: count with the whole wave (rect_count<NP, 64>), the up to six
// NFA values of a step in six lanes, the "keep if better" chain replayed exactly as k_nfa_math does.  Same device functions, same operations, same results.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double nfa_bcast(double v, int src) { return shfl_d(v, src); }

// nfa_d in three pieces, so that the long exit-free part of the binomial tail can be shared by the wave.  A rectangle of n pixels needs about n / 2 - k tail
// iterations, each `term *= ((n - i + 1) / i) * p_term; bin_tail += term`; for the large, sparse rectangles that go through all five stages that is 10^4
// iterations per NFA value, and the fp64 division -- 14 of the 17 instructions of an iteration -- does not depend on the running product at all.  So the up to six
// chains of a step (lanes 0..5) run in lock step and ALL 64 lanes compute the divisions of the next 512 iterations of every chain into LDS first; the chain lanes
// then only multiply and add.  Every operation and its order inside a chain are those of nfa_d (the division is the same IEEE operation whichever lane does it).
struct NfaIt { double term, bin_tail, p_term, val; int i, n, iend; bool live; };
#define NFA_COOP_BLK 512

__device__ __forceinline__ NfaIt nfa_head(const double *__restrict__ lgam, double LOG_NT, int n, int k, double p)
{
    NfaIt it;
    it.live = false; it.term = it.bin_tail = it.p_term = 0.0; it.i = it.iend = 0; it.n = n;
    if (n == 0 || k == 0) { it.val = -LOG_NT; return it; }
    if (n == k) { it.val = -LOG_NT - (double)n * log10(p); return it; }
    it.p_term = p / (1 - p);
    const double log1term = log_gamma_int(lgam, n + 1) - log_gamma_int(lgam, k + 1) - log_gamma_int(lgam, n - k + 1) +
                            (double)k * log(p) + (double)(n - k) * log(1.0 - p);
    const double term = exp(log1term);
    if (double_equal_d(term, 0)) {
        it.val = ((double)k > (double)n * p) ? -log1term / 2.30258509299404568402 - LOG_NT : -LOG_NT;
        return it;
    }
    it.live = true; it.term = term; it.bin_tail = term;
    int i = k + 1;
    it.i = i;
    // the iterations nfa_d takes without an exit test (its blocks of 8, then of 4)
    while (i + 7 <= n && n - (i + 7) + 1 >= i + 7) i += 8;
    while (i + 3 <= n && n - (i + 3) + 1 >= i + 3) i += 4;
    it.iend = i;
    return it;
}

// all 64 lanes; `it` of a lane that runs no chain has live == false.  tab: NFA_COOP_BLK doubles per chain lane (lanes 0..5)
__device__ __forceinline__ void nfa_coop(NfaIt &it, LDS_PTR(double) tab, double LOG_NT)
{
    const int lane = plf_lane();
    int done_j = 0;                                                   // iterations of the exit-free part already taken (the same for every chain: lock step)
    int len = it.live ? it.iend - it.i : 0;
    for (;;) {
        int maxlen = len;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));   // (chains live in lanes 0..5)
        maxlen = __builtin_amdgcn_readfirstlane(maxlen);
        if (done_j >= maxlen) break;
        // divisions of the next block of every chain
        for (int c = 0; c < 6; c++) {
            const int len_c = __builtin_amdgcn_readlane(len, c);
            if (len_c <= done_j) continue;
            const int n_c = __builtin_amdgcn_readlane(it.n, c), i_c = __builtin_amdgcn_readlane(it.i, c);   // (it.i: first iteration of the chain's exit-free part)
            const int cnt = min(NFA_COOP_BLK, len_c - done_j);
            for (int j = lane; j < cnt; j += 64) {
                const int i = i_c + done_j + j;
                tab[c * NFA_COOP_BLK + j] = (double)(n_c - i + 1) / (double)i;
            }
        }
        CBAR();
        if (lane < 6 && len > done_j) {
            const int cnt = min(NFA_COOP_BLK, len - done_j);
            LDS_PTR(double) tb = tab + lane * NFA_COOP_BLK;
            double term = it.term, bin_tail = it.bin_tail, lastm = 2.0;
            const double p_term = it.p_term;
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {
                const double m0 = tb[j] * p_term, m1 = tb[j + 1] * p_term, m2 = tb[j + 2] * p_term, m3 = tb[j + 3] * p_term;
                term *= m0; bin_tail += term;
                term *= m1; bin_tail += term;
                term *= m2; bin_tail += term;
                term *= m3; bin_tail += term;
                lastm = m3;
            }
            for (; j < cnt; j++) { const double m0 = tb[j] * p_term; term *= m0; bin_tail += term; lastm = m0; }
            it.term = term; it.bin_tail = bin_tail;
            if (NFA_DEAD_TAIL(lastm, term, bin_tail)) {   // (nfa_d's early exit, checked once per block)
                it.live = false; it.val = -log10(bin_tail) - LOG_NT; len = 0;
            }
        }
        CBAR();
        done_j += NFA_COOP_BLK;
    }
}

__device__ __forceinline__ double nfa_tail(const NfaIt &it, double LOG_NT)
{
    if (!it.live) return it.val;
    const int n = it.n;
    double term = it.term, bin_tail = it.bin_tail;
    const double p_term = it.p_term, tolerance = 0.1;
    for (int i = it.iend; i <= n; ++i) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - pow(mult_term, (double)(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * fabs(-log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - LOG_NT;
}

// wave-uniform condition as a scalar: the compiler cannot see that the NFA values are the same in every lane and would turn `if (c) { rect = r; }` into 24 selects
// (v_cndmask_b32_e32 back to back: 17-40 cycles each, profiles/r03_valu_issue.json) instead of a branch over 24 moves
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// (st: the rectangle as found, log_nfa -1; keep_all of its slot is already 0)
__device__ __forceinline__ NfaIt nfa_head_tab(const double *__restrict__ lgam, const double *__restrict__ nfatab, double LOG_NT, int n, int k, double p)
{
    double v;
    if (nfatab && nfa_lookup(nfatab, n, k, p, v)) {   // (k_nfa_table's value: bit-identical to evaluating here)
        NfaIt it;
        it.live = false; it.term = it.bin_tail = it.p_term = 0.0; it.i = it.iend = 0; it.n = n; it.val = v;
        return it;
    }
    return nfa_head(lgam, LOG_NT, n, k, p);
}

__device__ __forceinline__ void nfa_fused_rect(const float *__restrict__ ang, const double *__restrict__ lgam, const double *__restrict__ nfatab, LDS_PTR(double) tab, NfaState &st,
                                               uint8_t *__restrict__ keep_all, float4 *__restrict__ seg_all, const LsdGeom &g)
{
    const int lane = threadIdx.x;
    const double LOG_EPS = 0.0, delta = 0.5, delta_2 = delta / 2.0;
    {
        bool done = false;
        // ---- stage 0: the rectangle as found, at its precision and the five halved ones
        {
            NfaCounts c;
            rect_count<6, 64>(ang, g.sw, g.sh, st.rec, c);
            double v = -1.0e300;
            {
                NfaIt it; it.live = false; it.n = 0; it.i = it.iend = 0; it.val = v;
                if (lane < 6) {
                    double pp = st.rec.p;
                    for (int j = 0; j < lane; j++) pp /= 2;
                    it = nfa_head_tab(lgam, nfatab, g.log_nt, c.total, c.alg[lane < 6 ? lane : 0], pp);
                }
                nfa_coop(it, tab, g.log_nt);
                if (lane < 6) v = nfa_tail(it, g.log_nt);
            }
            st.log_nfa = nfa_bcast(v, 0);
            if (uni(st.log_nfa > LOG_EPS)) done = true;
            else {
                LsdRect r = st.rec;
                for (int k = 1; k <= 5; ++k) {
                    r.p /= 2;
                    r.prec = r.p * PI_D;
                    const double vk = nfa_bcast(v, k);
                    if (uni(vk > st.log_nfa)) { st.log_nfa = vk; st.rec = r; }
                }
                if (uni(st.log_nfa > LOG_EPS)) done = true;
            }
        }
        // ---- stages 1..3: five progressively narrower / shifted candidates each
        for (int stage = 0; stage <= 2 && !done; stage++) {
            // (the candidates are a progressive sequence: generated once for the counts, once more for the pick -- cheaper than holding five rectangles)
            int tot = 0, al = 0, non = 0;     // lane k keeps the counts of candidate k; non: candidates that exist (the width test fails for good once it fails)
            {
                LsdRect r = st.rec;
                for (int k = 0; k < 5; ++k) {
                    if (!((r.width - delta) >= 0.5)) break;
                    if (stage == 1) { r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; }
                    if (stage == 2) { r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; }
                    r.width -= delta;
                    NfaCounts c;
                    rect_count<1, 64>(ang, g.sw, g.sh, r, c);
                    if (lane == k) { tot = c.total; al = c.alg[0]; }
                    non = k + 1;
                }
            }
            double v = -1.0e300;
            {
                NfaIt it; it.live = false; it.n = 0; it.i = it.iend = 0; it.val = v;
                if (lane < non) it = nfa_head_tab(lgam, nfatab, g.log_nt, tot, al, st.rec.p);
                nfa_coop(it, tab, g.log_nt);
                if (lane < non) v = nfa_tail(it, g.log_nt);
            }
            {
                LsdRect r = st.rec;
                for (int k = 0; k < non; ++k) {
                    if (stage == 1) { r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; }
                    if (stage == 2) { r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; }
                    r.width -= delta;
                    const double vk = nfa_bcast(v, k);
                    if (uni(vk > st.log_nfa)) { st.rec = r; st.log_nfa = vk; }
                }
            }
            if (uni(st.log_nfa > LOG_EPS)) done = true;
        }
        // ---- stage 4: finer precisions of the rectangle the width searches ended with
        if (!done && (st.rec.width - delta) >= 0.5) {
            NfaCounts c;
            rect_count<6, 64>(ang, g.sw, g.sh, st.rec, c);
            double v = -1.0e300;
            {
                NfaIt it; it.live = false; it.n = 0; it.i = it.iend = 0; it.val = v;
                if (lane >= 1 && lane < 6) {
                    double pp = st.rec.p;
                    for (int j = 0; j < lane; j++) pp /= 2;
                    it = nfa_head_tab(lgam, nfatab, g.log_nt, c.total, c.alg[lane < 6 ? lane : 0], pp);
                }
                nfa_coop(it, tab, g.log_nt);
                if (lane >= 1 && lane < 6) v = nfa_tail(it, g.log_nt);
            }
            LsdRect r = st.rec;
            for (int k = 1; k <= 5; ++k) {
                r.p /= 2;
                r.prec = r.p * PI_D;
                const double vk = nfa_bcast(v, k);
                if (uni(vk > st.log_nfa)) { st.log_nfa = vk; st.rec = r; }
            }
            if (uni(st.log_nfa > LOG_EPS)) done = true;
        }
        if (done && lane == 0) nfa_finish(st, seg_all, keep_all, g);
    }
}

__global__ void __launch_bounds__(64) k_nfa_fused(const float *__restrict__ ang_all, const double *__restrict__ lgam, const double *__restrict__ nfatab, const LsdRect *__restrict__ rects_all,
                                                  const int *__restrict__ nrect, uint8_t *__restrict__ keep_all, float4 *__restrict__ seg_all, LsdGeom g)
{
    __shared__ double tab_s[6 * NFA_COOP_BLK];
    LDS_PTR(double) tab = (LDS_PTR(double))tab_s;
    const int f = blockIdx.y, lane = threadIdx.x, n_r = nrect[f];
    const float *ang = ang_all + (size_t)f * g.s_stride;
    for (int ri = blockIdx.x; ri < n_r; ri += gridDim.x) {
        NfaState st;
        st.rec = rects_all[(size_t)f * g.rect_cap + ri]; st.log_nfa = -1; st.frame = f; st.rect = ri;
        if (lane == 0) keep_all[(size_t)f * g.rect_cap + ri] = 0;
        nfa_fused_rect(ang, lgam, nfatab, tab, st, keep_all, seg_all, g);
    }
}
