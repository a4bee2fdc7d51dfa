// orb_kernels.hip -- HIP kernels of the ORB extractor for gfx950 (MI355X).
//
// Pipeline per batch of B frames (every launch covers all frames; most cover all levels):
//   k_pyr_level0 / k_pyr_resize   ORBextractor::ComputePyramid           (include/ORBextractor.h:89, so@0x70430)
//   k_fast_score                  per-pixel FAST-9/16 corner score        (cv::FAST inside so@0x75fa0)
//   k_fast_cells                  per-cell threshold/retry + 3x3 NMS      (ComputeKeyPointsOctTree cell loop)
//   k_octree                      DistributeOctTree / DivideNode          (orb_octree.hip)
//   k_blur7                       GaussianBlur 7x7 sigma 2, 8-bit fixed    (operator(), so@0x77487)
//   k_orient_brief                IC_Angle + steered BRIEF + final layout (so@0x6fb10, so@0x777b5)
//
// Design notes (MI355X): the work is byte/integer stencil, gather and compaction -- HBM/L2 bound,
// no MFMA.  Images are 8-bit planes read with coalesced row accesses and staged through LDS tiles;
// wave64 ballots give the raster-ordered compaction the reference's sequential loops imply.
// All float math is compiled with -ffp-contract=off; the two FMAs the reference binary uses are
// explicit fmaf().
#include "plf_common.h"
#include "orb_geom.h"
#include "orb_pattern.inc"

__constant__ signed char c_pattern[1024];
__constant__ int c_umax[16];

void plf_orb_upload_constants(const int *umax16)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), plf_bit_pattern_31, 1024);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int));
}

// ------------------------------------------------------------------------------------------------
// Pyramid.  Padded plane of level l: (w+38) x (h+38), interior at (19,19), REFLECT_101 border.
// One thread per padded byte; border pixels recompute the value of their mirror source, so a
// level is finished by a single pass (no separate copyMakeBorder pass).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyr_level0(const uint8_t *__restrict__ in, ptrdiff_t in_pitch, ptrdiff_t in_fstride,
                                                    uint8_t *__restrict__ pyr, OrbGeom g)
{
    const OrbLevel &L = g.lv[0];
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y, f = blockIdx.z;
    if (px >= L.ppitch) return;
    const int sx = plf_reflect101(px - PLF_EDGE, L.w), sy = plf_reflect101(py - PLF_EDGE, L.h);
    pyr[(size_t)f * g.pyr_stride + L.plane_off + (size_t)py * L.ppitch + px] = in[(size_t)f * in_fstride + (size_t)sy * in_pitch + sx];
}

// cv::resize INTER_LINEAR 8UC1: coefficient tables (xofs, ialpha, yofs, ibeta) are built on the host
// exactly as OpenCV does (double -> float -> 11-bit fixed point); the kernel evaluates
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
__global__ void __launch_bounds__(256) k_pyr_resize(uint8_t *__restrict__ pyr, OrbGeom g, int l, const int *__restrict__ xofs,
                                                    const short2 *__restrict__ xa, const int *__restrict__ yofs,
                                                    const short2 *__restrict__ yb)
{
    const OrbLevel &D = g.lv[l];
    const OrbLevel &S = g.lv[l - 1];
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y, f = blockIdx.z;
    if (px >= D.ppitch) return;
    const int dx = plf_reflect101(px - PLF_EDGE, D.w), dy = plf_reflect101(py - PLF_EDGE, D.h);
    const uint8_t *src = pyr + (size_t)f * g.pyr_stride + S.plane_off + (size_t)PLF_EDGE * S.ppitch + PLF_EDGE;
    const int sx = xofs[D.tabx_off + dx];
    const short2 a = xa[D.tabx_off + dx];
    const int sy = yofs[D.taby_off + dy];
    const short2 b = yb[D.taby_off + dy];
    const int y0 = min(max(sy, 0), S.h - 1), y1 = min(max(sy + 1, 0), S.h - 1);
    const int sx1 = min(sx + 1, S.w - 1);
    const uint8_t *r0 = src + (size_t)y0 * S.ppitch, *r1 = src + (size_t)y1 * S.ppitch;
    const int s0 = r0[sx] * a.x + r0[sx1] * a.y;
    const int s1 = r1[sx] * a.x + r1[sx1] * a.y;
    const int v = (((b.x * (s0 >> 4)) >> 16) + ((b.y * (s1 >> 4)) >> 16) + 2) >> 2;
    pyr[(size_t)f * g.pyr_stride + D.plane_off + (size_t)py * D.ppitch + px] = (uint8_t)v;
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 corner score map.  score(p) = cornerScore<16>(p) = (max over the 16 arcs of 9 contiguous
// ring pixels of the minimum |I_p - I_x| with a common sign) - 1, clamped at 0.  A pixel is a corner
// at threshold t iff score >= t, and cv::FAST stores exactly this score, so the per-cell
// threshold/retry logic and the NMS can run afterwards on the map (k_fast_cells).
// Tile: 64 x 16 pixels per 256-thread block, staged through LDS with a 3-pixel halo.
// ------------------------------------------------------------------------------------------------
#define TILE_W 64
#define TILE_H 16
#define LT_PITCH 72

__device__ __forceinline__ int find_level_by_tile(const OrbGeom &g, int tile)
{
    int l = 0;
    for (int i = 1; i < g.nlevels; i++)
        if (tile >= g.lv[i].tile_base) l = i;
    return l;
}

__device__ __forceinline__ int fast_score16(const uint8_t *c, int minTh)
{
    // ring offsets (x,y): same circle as cv::FAST; only contiguity matters
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * LT_PITCH + 0];  d[1] = v - c[3 * LT_PITCH + 1];  d[2] = v - c[2 * LT_PITCH + 2];  d[3] = v - c[1 * LT_PITCH + 3];
    d[4] = v - c[3];                 d[5] = v - c[-1 * LT_PITCH + 3]; d[6] = v - c[-2 * LT_PITCH + 2]; d[7] = v - c[-3 * LT_PITCH + 1];
    d[8] = v - c[-3 * LT_PITCH];     d[9] = v - c[-3 * LT_PITCH - 1]; d[10] = v - c[-2 * LT_PITCH - 2]; d[11] = v - c[-1 * LT_PITCH - 3];
    d[12] = v - c[-3];               d[13] = v - c[1 * LT_PITCH - 3]; d[14] = v - c[2 * LT_PITCH - 2]; d[15] = v - c[3 * LT_PITCH - 1];
    // cheap necessary condition at the lowest threshold: every 9-arc contains one pixel of each opposite pair
    const int t = minTh;
    bool br = true, dk = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        br = br && (d[k] > t || d[k + 8] > t);
        dk = dk && (d[k] < -t || d[k + 8] < -t);
    }
    if (!br && !dk) return 0;
    int m3[16], M3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        m3[k] = min(d[k], min(d[(k + 1) & 15], d[(k + 2) & 15]));
        M3[k] = max(d[k], max(d[(k + 1) & 15], d[(k + 2) & 15]));
    }
    int sb = -256, sd = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        sb = max(sb, min(m3[k], min(m3[(k + 3) & 15], m3[(k + 6) & 15])));
        sd = min(sd, max(M3[k], max(M3[(k + 3) & 15], M3[(k + 6) & 15])));
    }
    const int s = max(sb, -sd) - 1;
    return s < 0 ? 0 : s;
}

__global__ void __launch_bounds__(256) k_fast_score(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ score, OrbGeom g)
{
    __shared__ uint8_t tile[(TILE_H + 6) * LT_PITCH];
    const int f = blockIdx.y;
    const int l = find_level_by_tile(g, blockIdx.x);
    const OrbLevel &L = g.lv[l];
    const int tl = blockIdx.x - L.tile_base;
    const int tx = tl % L.tiles_x, ty = tl / L.tiles_x;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;
    const uint8_t *img = pyr + (size_t)f * g.pyr_stride + L.plane_off + (size_t)PLF_EDGE * L.ppitch + PLF_EDGE;
    for (int i = threadIdx.x; i < (TILE_H + 6) * (TILE_W + 6); i += 256) {
        const int r = i / (TILE_W + 6), c = i - r * (TILE_W + 6);
        int gx = x0 - 3 + c, gy = y0 - 3 + r;
        gx = min(gx, L.w + PLF_EDGE - 1);  // stays inside the padded plane
        gy = min(gy, L.h + PLF_EDGE - 1);
        tile[r * LT_PITCH + c] = img[(ptrdiff_t)gy * L.ppitch + gx];
    }
    __syncthreads();
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    uint8_t *sp = score + (size_t)f * g.blur_stride + L.blur_off;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = ry + 4 * k;
        const int gx = x0 + cx, gy = y0 + r;
        // the FAST cells only ever look at x in [19, w-19), y in [19, h-19)
        if (gx >= PLF_EDGE && gy >= PLF_EDGE && gx < L.w - PLF_EDGE && gy < L.h - PLF_EDGE) {
            const int s = fast_score16(&tile[(r + 3) * LT_PITCH + cx + 3], g.minTh);
            sp[(size_t)gy * L.bpitch + gx] = (uint8_t)s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Per-cell detection: one wave per (cell, frame).  Restates, for the cell's sub-image, what two
// cv::FAST(..., nonmax=true) calls would return: pixels of the computed region (sub-image minus a
// 3-px frame) whose score is >= threshold and strictly greater than the scores of the 8 neighbours
// INSIDE the computed region (outside counts as 0); threshold = iniTh, or minTh if that leaves the
// cell empty.  (A neighbour below the threshold counts as 0 in cv::FAST, which cannot change the
// comparison because the centre is >= threshold.)  Output order = raster inside the cell; the cell's
// chunk is allocated from the level's pool with one atomic, and (base,count) is recorded per cell so
// that the octree kernel can gather cells in reference order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fast_cells(const uint8_t *__restrict__ score, const int4 *__restrict__ cells,
                                                   int2 *__restrict__ cellinfo, uint2 *__restrict__ pool,
                                                   int *__restrict__ poolcnt, int *__restrict__ status, OrbGeom g)
{
    const int f = blockIdx.y, cell = blockIdx.x, lane = threadIdx.x;
    int l = 0;
    for (int i = 1; i < g.nlevels; i++) if (cell >= g.lv[i].cell_base) l = i;
    const OrbLevel &L = g.lv[l];
    const int4 rc = cells[cell];  // x0, y0, w, h of the sub-image (level interior coords)
    const int cw = rc.z - 6, ch = rc.w - 6;  // computed region
    const uint8_t *sp = score + (size_t)f * g.blur_stride + L.blur_off;
    const int gx = rc.x + 3 + lane;
    const bool colok = lane < cw;
    unsigned long long my20 = 0, my7 = 0;  // lane r keeps the masks of row r
    int up = 0, mid = 0, dn = 0;
    if (colok && ch > 0) mid = sp[(size_t)(rc.y + 3) * L.bpitch + gx];
    for (int r = 0; r < ch; r++) {
        dn = (colok && r + 1 < ch) ? sp[(size_t)(rc.y + 3 + r + 1) * L.bpitch + gx] : 0;
        int m = max(up, dn);
        // 3-column max of (up, mid, dn) from the left and right neighbour lanes (0 outside the region)
        const int col3 = max(m, mid);
        int lft = __shfl_up(col3, 1, 64), rgt = __shfl_down(col3, 1, 64);
        if (lane == 0) lft = 0;
        if (lane == 63) rgt = 0;
        const int nb = max(m, max(lft, rgt));
        const bool ismax = colok && mid > nb;
        const unsigned long long b20 = __ballot(ismax && mid >= g.iniTh);
        const unsigned long long b7 = __ballot(ismax && mid >= g.minTh);
        if (lane == r) { my20 = b20; my7 = b7; }
        up = mid; mid = dn;
    }
    const int n20 = plf_wave_sum(__popcll(my20));
    const unsigned long long mine = n20 > 0 ? my20 : my7;
    const int cnt = __popcll(mine);
    const int total = plf_wave_sum(cnt);
    const int excl = plf_wave_excl_scan(cnt);
    int base = 0;
    if (lane == 0 && total > 0) base = atomicAdd(&poolcnt[f * g.nlevels + l], total);
    base = __shfl(base, 0, 64);
    if (lane == 0) cellinfo[(size_t)f * g.cells_total + cell] = make_int2(base, total);
    if (total == 0) return;
    if (base + total > (int)L.pool_cap) {  // cannot happen (pool sized for the densest possible NMS output)
        if (lane == 0) atomicOr(status, 1);
        return;
    }
    uint2 *out = pool + (size_t)f * g.pool_stride + L.pool_off + base + excl;
    unsigned long long mm = mine;
    const int gy = rc.y + 3 + lane;
    int k = 0;
    while (mm) {
        const int c = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int x = rc.x + 3 + c;
        const int resp = sp[(size_t)gy * L.bpitch + x];
        // coordinates relative to (minBorderX, minBorderY) as DistributeOctTree expects
        out[k++] = make_uint2((uint32_t)(x - PLF_MINB) | ((uint32_t)(gy - PLF_MINB) << 16), (uint32_t)resp);
    }
}

// ------------------------------------------------------------------------------------------------
// GaussianBlur(7x7, sigma 2) on 8U: OpenCV 3.3 separable fixed-point path, taps round(k*256) per axis
// (sum 257, not renormalised), int32 row pass, column pass rounded like the SSE2 column filter
// (exact sum/65536 to nearest-even) for x < (w & ~3) and (sum + 32768) >> 16 for the last w % 4
// columns.  The padded pyramid plane already holds the REFLECT_101 border the blur needs.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_blur7(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, OrbGeom g, int4 taps)
{
    __shared__ uint8_t raw[(TILE_H + 6) * LT_PITCH];
    __shared__ int hrow[(TILE_H + 6) * TILE_W];
    const int f = blockIdx.y;
    const int l = find_level_by_tile(g, blockIdx.x);
    const OrbLevel &L = g.lv[l];
    const int tl = blockIdx.x - L.tile_base;
    const int tx = tl % L.tiles_x, ty = tl / L.tiles_x;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;
    const uint8_t *img = pyr + (size_t)f * g.pyr_stride + L.plane_off + (size_t)PLF_EDGE * L.ppitch + PLF_EDGE;
    for (int i = threadIdx.x; i < (TILE_H + 6) * (TILE_W + 6); i += 256) {
        const int r = i / (TILE_W + 6), c = i - r * (TILE_W + 6);
        int gx = x0 - 3 + c, gy = y0 - 3 + r;
        gx = min(gx, L.w + PLF_EDGE - 1);
        gy = min(gy, L.h + PLF_EDGE - 1);
        raw[r * LT_PITCH + c] = img[(ptrdiff_t)gy * L.ppitch + gx];
    }
    __syncthreads();
    const int k0 = taps.x, k1 = taps.y, k2 = taps.z, k3 = taps.w;
    for (int i = threadIdx.x; i < (TILE_H + 6) * TILE_W; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t *p = &raw[r * LT_PITCH + c];
        hrow[i] = k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3];
    }
    __syncthreads();
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int wvec = L.w & ~3;
    uint8_t *bp = blur + (size_t)f * g.blur_stride + L.blur_off;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = ry + 4 * k;
        const int gx = x0 + cx, gy = y0 + r;
        if (gx < L.w && gy < L.h) {
            const int *q = &hrow[r * TILE_W + cx];
            const int s = k0 * (q[0] + q[6 * TILE_W]) + k1 * (q[TILE_W] + q[5 * TILE_W]) + k2 * (q[2 * TILE_W] + q[4 * TILE_W]) + k3 * q[3 * TILE_W];
            int v;
            if (gx < wvec) {
                v = s >> 16;
                const int rem = s & 0xFFFF;
                if (rem > 0x8000 || (rem == 0x8000 && (v & 1))) v++;
            } else {
                v = (s + 32768) >> 16;
            }
            bp[(size_t)gy * L.bpitch + gx] = (uint8_t)min(v, 255);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor: one wave per selected keypoint.
//   IC_Angle: integer moments over the radius-15 disc of the UN-blurred level, lanes = patch rows,
//             angle = cv::fastAtan2(float(m01), float(m10)).
//   steered BRIEF on the blurred level: lane j evaluates comparisons 4j..4j+3; sample coordinates
//             row = cvRound(fmaf(px, b, py*a)), col = cvRound(fmaf(px, a, -(py*b))) (the reference
//             binary contracts exactly these two FMAs), (b, a) = glibc sincosf(angle * 0.01745329238f),
//             re-evaluated operation by operation in double (plf_sincosf_glibc).
// Output layout: level-major; the offset of level l is the sum of the counts of the levels below.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_orient_brief(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                     const uint2 *__restrict__ sel, const int *__restrict__ selcnt,
                                                     plf_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
                                                     int *__restrict__ n_out, int capacity, int *__restrict__ status, OrbGeom g)
{
    const int idx = blockIdx.x, l = blockIdx.y, f = blockIdx.z, lane = threadIdx.x;
    const int *cnt = selcnt + f * g.nlevels;
    int offset = 0, total = 0;
    for (int i = 0; i < g.nlevels; i++) {
        const int c = cnt[i];
        if (i < l) offset += c;
        total += c;
    }
    if (idx == 0 && l == 0 && lane == 0) {
        n_out[f] = min(total, capacity);
        if (total > capacity) atomicOr(status, 2);
    }
    if (idx >= cnt[l]) return;
    const int o = offset + idx;
    if (o >= capacity) return;
    const OrbLevel &L = g.lv[l];
    const uint2 s = sel[(size_t)f * g.sel_stride + L.sel_off + idx];
    const int x = (int)(s.x & 0xFFFF) + PLF_MINB, y = (int)(s.x >> 16) + PLF_MINB;  // level coordinates (integers)
    const uint8_t *img = pyr + (size_t)f * g.pyr_stride + L.plane_off + (size_t)PLF_EDGE * L.ppitch + PLF_EDGE;
    const uint8_t *center = img + (ptrdiff_t)y * L.ppitch + x;
    int m10 = 0, m01 = 0;
    if (lane < PLF_PATCH) {
        const int v = lane - PLF_HALF_PATCH;
        const int d = c_umax[v < 0 ? -v : v];
        const uint8_t *row = center + (ptrdiff_t)v * L.ppitch;
        int su = 0, si = 0;
        for (int u = -d; u <= d; ++u) {
            const int I = row[u];
            su += u * I;
            si += I;
        }
        m10 = su;
        m01 = v * si;
    }
    m10 = plf_wave_sum(m10);
    m01 = plf_wave_sum(m01);
    const float angle = plf_fast_atan2((float)m01, (float)m10);
    // steered BRIEF
    const float arad = angle * 0.01745329238f;
    float a, b;
    plf_sincosf_glibc(arad, &b, &a);  // a = cos, b = sin, as glibc's sincosf returns them (so@0x77803)
    const uint8_t *bc = blur + (size_t)f * g.blur_stride + L.blur_off + (size_t)y * L.bpitch + x;
    uint32_t bits = 0;
    const signed char *pat = c_pattern + lane * 16;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float px0 = pat[4 * k], py0 = pat[4 * k + 1], px1 = pat[4 * k + 2], py1 = pat[4 * k + 3];
        const int r0 = __float2int_rn(fmaf(px0, b, py0 * a)), c0 = __float2int_rn(fmaf(px0, a, -(py0 * b)));
        const int r1 = __float2int_rn(fmaf(px1, b, py1 * a)), c1 = __float2int_rn(fmaf(px1, a, -(py1 * b)));
        const int t0 = bc[(ptrdiff_t)r0 * L.bpitch + c0], t1 = bc[(ptrdiff_t)r1 * L.bpitch + c1];
        bits |= (uint32_t)(t0 < t1) << k;
    }
    // lane j holds bits 4j..4j+3 -> nibble (j&1) of byte j>>1; assemble dwords in lanes 0,8,16,..
    uint32_t v = bits << (4 * (lane & 7));
    v |= __shfl_xor(v, 1, 64);
    v |= __shfl_xor(v, 2, 64);
    v |= __shfl_xor(v, 4, 64);
    if ((lane & 7) == 0) reinterpret_cast<uint32_t *>(desc + ((size_t)f * capacity + o) * 32)[lane >> 3] = v;
    if (lane == 0) {
        plf_keypoint kp;
        kp.x = (float)x; kp.y = (float)y;
        if (l != 0) { kp.x = kp.x * L.scale; kp.y = kp.y * L.scale; }
        kp.size = (float)L.size_i;
        kp.angle = angle;
        kp.response = (float)s.y;
        kp.octave = l;
        kp.class_id = -1;
        kps[(size_t)f * capacity + o] = kp;
    }
}
