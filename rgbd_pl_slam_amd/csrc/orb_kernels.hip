// orb_kernels.hip -- HIP kernels of the ORB extractor for gfx950 (MI355X).
//
// Pipeline per batch of B frames (every launch covers all frames; most cover all levels):
//   k_pyr_level0 / k_pyr_resize   ORBextractor::ComputePyramid           (include/ORBextractor.h:89, so@0x70430)
//   k_score_blur                  per-pixel FAST-9/16 corner score        (cv::FAST inside so@0x75fa0) fused with
//                                 GaussianBlur 7x7 sigma 2, 8-bit fixed    (operator(), so@0x77487)
//   k_fast_cells                  per-cell threshold/retry + 3x3 NMS      (ComputeKeyPointsOctTree cell loop)
//   k_octree                      DistributeOctTree / DivideNode          (orb_octree.hip)
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

typedef uint32_t __attribute__((aligned(1))) plf_u32u;   // dword access at byte alignment (legal on gfx950 global memory)
typedef unsigned long long __attribute__((aligned(1))) plf_u64u;
struct __attribute__((aligned(4))) plf_int4u { int x, y, z, w; };

__constant__ signed char c_pattern[1024];
__constant__ int c_umax[16];

void plf_orb_upload_constants(const int *umax16)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), plf_bit_pattern_31, 1024);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int));
}

// ------------------------------------------------------------------------------------------------
// Pyramid.  Padded plane of level l: (w+38) x (h+38), interior at (19,19), REFLECT_101 border.
// One thread per 4 padded bytes of a row; border pixels recompute the value of their mirror source, so a
// level is finished by a single pass (no separate copyMakeBorder pass).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyr_level0(const uint8_t *__restrict__ in, ptrdiff_t in_pitch, ptrdiff_t in_fstride,
                                                    uint8_t *__restrict__ pyr, OrbGeom g)
{
    const OrbLevel &L = g.lv[0];
    const int gpr = (L.ppitch + 3) >> 2, id = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    const int py = id / gpr, px0 = (id - py * gpr) * 4;
    if (py >= L.h + 2 * PLF_EDGE) return;
    const uint8_t *row = in + (size_t)f * in_fstride + (size_t)plf_reflect101(py - PLF_EDGE, L.h) * in_pitch;
    uint8_t *dst = pyr + (size_t)f * g.pyr_stride + L.plane_off + (size_t)py * L.ppitch + px0;
    const int x = px0 - PLF_EDGE;
    if (x >= 0 && x + 3 < L.w) { *(plf_u32u *)dst = *(const plf_u32u *)(row + x); return; }   // interior: straight copy
    for (int j = 0; j < 4 && px0 + j < L.ppitch; j++) dst[j] = row[plf_reflect101(x + j, L.w)];
}

// cv::resize INTER_LINEAR 8UC1: coefficient tables (xofs, ialpha, yofs, ibeta) are built on the host
// exactly as OpenCV does (double -> float -> 11-bit fixed point); the kernel evaluates
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// A thread produces 4 consecutive bytes of one padded row (row terms shared, one dword store).
__global__ void __launch_bounds__(256) k_pyr_resize(uint8_t *__restrict__ pyr, OrbGeom g, int l, const int *__restrict__ xofs,
                                                    const short2 *__restrict__ xa, const int *__restrict__ yofs,
                                                    const short2 *__restrict__ yb)
{
    const OrbLevel &D = g.lv[l];
    const OrbLevel &S = g.lv[l - 1];
    // (row, 4-byte group) pairs are numbered linearly so that every workgroup is full whatever the level width
    const int gpr = (D.ppitch + 3) >> 2, id = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    const int py = id / gpr, px0 = (id - py * gpr) * 4;
    if (py >= D.h + 2 * PLF_EDGE) return;
    const int dy = plf_reflect101(py - PLF_EDGE, D.h);
    const uint8_t *src = pyr + (size_t)f * g.pyr_stride + S.plane_off + (size_t)PLF_EDGE * S.ppitch + PLF_EDGE;
    const int sy = yofs[D.taby_off + dy];
    const short2 b = yb[D.taby_off + dy];
    const int y0 = min(max(sy, 0), S.h - 1), y1 = min(max(sy + 1, 0), S.h - 1);
    const uint8_t *r0 = src + (size_t)y0 * S.ppitch, *r1 = src + (size_t)y1 * S.ppitch;
    uint32_t out = 0;
    const int x = px0 - PLF_EDGE;
    bool done = false;
    if (x >= 0 && x + 3 < D.w) {
        // interior group: 4 consecutive table entries (two 16-byte loads) and the <= 8 source bytes per row they
        // address (two 8-byte loads) instead of 24 scalar gathers
        const plf_int4u so = *(const plf_int4u *)(xofs + D.tabx_off + x);
        const int base = so.x;
        if (so.w + 1 - base <= 7) {
            const plf_int4u ar = *(const plf_int4u *)(xa + D.tabx_off + x);   // 4 x short2
            const unsigned long long w0 = *(const plf_u64u *)(r0 + base), w1 = *(const plf_u64u *)(r1 + base);
            const int sxs[4] = {so.x, so.y, so.z, so.w}, as[4] = {ar.x, ar.y, ar.z, ar.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int o0 = 8 * (sxs[j] - base), o1 = 8 * (min(sxs[j] + 1, S.w - 1) - base);
                const int ax = (short)(as[j] & 0xFFFF), ay = as[j] >> 16;
                const int s0 = (int)((w0 >> o0) & 0xFF) * ax + (int)((w0 >> o1) & 0xFF) * ay;
                const int s1 = (int)((w1 >> o0) & 0xFF) * ax + (int)((w1 >> o1) & 0xFF) * ay;
                const int v = (((b.x * (s0 >> 4)) >> 16) + ((b.y * (s1 >> 4)) >> 16) + 2) >> 2;
                out |= (uint32_t)(v & 0xFF) << (8 * j);
            }
            done = true;
        }
    }
    if (!done) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dx = plf_reflect101(min(px0 + j, D.ppitch - 1) - PLF_EDGE, D.w);
            const int sx = xofs[D.tabx_off + dx];
            const short2 a = xa[D.tabx_off + dx];
            const int sx1 = min(sx + 1, S.w - 1);
            const int s0 = r0[sx] * a.x + r0[sx1] * a.y;
            const int s1 = r1[sx] * a.x + r1[sx1] * a.y;
            const int v = (((b.x * (s0 >> 4)) >> 16) + ((b.y * (s1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xFF) << (8 * j);
        }
    }
    uint8_t *dst = pyr + (size_t)f * g.pyr_stride + D.plane_off + (size_t)py * D.ppitch + px0;
    if (px0 + 3 < D.ppitch) *(plf_u32u *)dst = out;
    else for (int j = 0; j < 4 && px0 + j < D.ppitch; j++) dst[j] = (uint8_t)(out >> (8 * j));
}

// ------------------------------------------------------------------------------------------------
// Fused FAST score + GaussianBlur 7x7 (the two full-resolution passes over every pyramid level).
// FAST-9/16 score(p) = cornerScore<16>(p) = (max over the 16 arcs of 9 contiguous ring pixels of the minimum
// |I_p - I_x| with a common sign) - 1, clamped at 0.  A pixel is a corner at threshold t iff score >= t, and
// cv::FAST stores exactly this score, so the per-cell threshold/retry logic and the NMS can run afterwards on the
// map (k_fast_cells).
// GaussianBlur(7x7, sigma 2) on 8U: OpenCV 3.3 separable fixed-point path, taps round(k*256) per axis (sum 257, not
// renormalised), exact int32 sums, rounded once like the SSE2 column filter (sum/65536 to nearest-even) for
// x < (w & ~3) and (sum + 32768) >> 16 for the last w % 4 columns.  The padded pyramid plane already holds the
// REFLECT_101 border the blur needs.
// A lane owns a strip of 4 pixels x SB_RS rows and walks it top to bottom with a
// 7-row sliding window held in registers: per input row 3 dword loads (12 bytes = the 4 pixels and their 3-pixel
// halo), no LDS, no barrier.  From the window it produces, for the centre row, the 4 blur bytes (exact integer
// sum of taps, both passes) and the 4 FAST scores.
// ------------------------------------------------------------------------------------------------
#define SB_RS 16

__device__ __forceinline__ int fast_score_ring(const int d[16], int t)
{
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

// FAST score of TWO horizontally adjacent pixels at once in packed int16 lanes (v_pk_sub/min/max_i16): the ring
// differences are in [-255, 255].  Same arithmetic as fast_score_ring: the opposite-pair precheck at the lowest
// threshold gates each pixel separately, the min3/max3 sliding windows give the score.
typedef short plf_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ plf_s2 pk_min(plf_s2 a, plf_s2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ plf_s2 pk_max(plf_s2 a, plf_s2 b) { return __builtin_elementwise_max(a, b); }
// bytes i and i+1 (0 <= i <= 10) of the 12-byte row segment R as two zero-extended int16 (one v_perm_b32)
#define SB_PAIR(R, i) __builtin_bit_cast(plf_s2, __builtin_amdgcn_perm((R)[((i) >> 2) < 2 ? ((i) >> 2) + 1 : 2], (R)[(i) >> 2], \
                                                                       (uint32_t)((i) & 3) | 0x0C000C00u | ((uint32_t)(((i) & 3) + 1) << 16)))

__device__ __forceinline__ void fast_score_pair(const uint32_t (*raw)[3], int j, int t, int &s_lo, int &s_hi)
{
    const int c = 4 + j;
    const plf_s2 v = SB_PAIR(raw[3], c);
    plf_s2 d[16];
    d[0] = v - SB_PAIR(raw[6], c);      d[1] = v - SB_PAIR(raw[6], c + 1);  d[2] = v - SB_PAIR(raw[5], c + 2);  d[3] = v - SB_PAIR(raw[4], c + 3);
    d[4] = v - SB_PAIR(raw[3], c + 3);  d[5] = v - SB_PAIR(raw[2], c + 3);  d[6] = v - SB_PAIR(raw[1], c + 2);  d[7] = v - SB_PAIR(raw[0], c + 1);
    d[8] = v - SB_PAIR(raw[0], c);      d[9] = v - SB_PAIR(raw[0], c - 1);  d[10] = v - SB_PAIR(raw[1], c - 2); d[11] = v - SB_PAIR(raw[2], c - 3);
    d[12] = v - SB_PAIR(raw[3], c - 3); d[13] = v - SB_PAIR(raw[4], c - 3); d[14] = v - SB_PAIR(raw[5], c - 2); d[15] = v - SB_PAIR(raw[6], c - 1);
    // every 9-arc contains one pixel of each opposite pair: bright needs min_k max(d[k], d[k+8]) > t, dark max_k min(...) < -t
    plf_s2 bmin = pk_max(d[0], d[8]), dmax = pk_min(d[0], d[8]);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        bmin = pk_min(bmin, pk_max(d[k], d[k + 8]));
        dmax = pk_max(dmax, pk_min(d[k], d[k + 8]));
    }
    const bool p_lo = bmin.x > t || dmax.x < -t, p_hi = bmin.y > t || dmax.y < -t;
    s_lo = 0; s_hi = 0;
    if (!(p_lo || p_hi)) return;
    plf_s2 m3[16], M3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        m3[k] = pk_min(d[k], pk_min(d[(k + 1) & 15], d[(k + 2) & 15]));
        M3[k] = pk_max(d[k], pk_max(d[(k + 1) & 15], d[(k + 2) & 15]));
    }
    plf_s2 sb = {-256, -256}, sd = {256, 256};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        sb = pk_max(sb, pk_min(m3[k], pk_min(m3[(k + 3) & 15], m3[(k + 6) & 15])));
        sd = pk_min(sd, pk_max(M3[k], pk_max(M3[(k + 3) & 15], M3[(k + 6) & 15])));
    }
    const plf_s2 zero = {0, 0}, one = {1, 1};
    const plf_s2 sc = pk_max(pk_max(sb, zero - sd) - one, zero);
    if (p_lo) s_lo = sc.x;
    if (p_hi) s_hi = sc.y;
}

// byte i (0..11) of the 12-byte row segment held in three dwords
#define SB_BYTE(R, i) ((int)(((R)[(i) >> 2] >> (8 * ((i) & 3))) & 0xFFu))

__global__ void __launch_bounds__(64) k_score_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ score, uint8_t *__restrict__ blur,
                                                   OrbGeom g, int4 taps)
{
    // strips (4 px x SB_RS rows) of all levels are numbered linearly, level-major, row-major inside a level; a wave
    // takes 64 consecutive strips of one level, so only the last wave of a level has idle lanes
    const int f = blockIdx.y, lane = threadIdx.x;
    int l = 0, base = 0, sx_n = 1, ns = 0;
    for (int i = 0; i < g.nlevels; i++) {
        sx_n = (g.lv[i].w + 3) >> 2;
        ns = sx_n * ((g.lv[i].h + SB_RS - 1) / SB_RS);
        const int nw = (ns + 63) >> 6;
        l = i;
        if ((int)blockIdx.x < base + nw) break;
        base += nw;
    }
    const OrbLevel &L = g.lv[l];
    const int sid = ((int)blockIdx.x - base) * 64 + lane;
    if (sid >= ns) return;
    const int x = (sid % sx_n) * 4, y0 = (sid / sx_n) * SB_RS;
    if (x >= L.w) return;
    const uint8_t *colp = pyr + (size_t)f * g.pyr_stride + L.plane_off + (size_t)PLF_EDGE * L.ppitch + PLF_EDGE + (x - 4);
    uint8_t *bp = blur + (size_t)f * g.blur_stride + L.blur_off;
    uint8_t *sp = score + (size_t)f * g.blur_stride + L.blur_off;
    const int k0 = taps.x, k1 = taps.y, k2 = taps.z, k3 = taps.w;
    const bool even_round = x < (L.w & ~3), full = x + 3 < L.w;
    uint32_t raw[7][3];
    int hs[7][4];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        raw[i][0] = raw[i][1] = raw[i][2] = 0u;
        hs[i][0] = hs[i][1] = hs[i][2] = hs[i][3] = 0;
    }
    for (int r = 0; r < SB_RS + 6; r++) {
        const int gy = min(y0 - 3 + r, L.h + PLF_EDGE - 1);   // stays inside the padded plane
        const uint8_t *rp = colp + (ptrdiff_t)gy * L.ppitch;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            raw[i][0] = raw[i + 1][0]; raw[i][1] = raw[i + 1][1]; raw[i][2] = raw[i + 1][2];
            hs[i][0] = hs[i + 1][0]; hs[i][1] = hs[i + 1][1]; hs[i][2] = hs[i + 1][2]; hs[i][3] = hs[i + 1][3];
        }
        raw[6][0] = *(const plf_u32u *)rp; raw[6][1] = *(const plf_u32u *)(rp + 4); raw[6][2] = *(const plf_u32u *)(rp + 8);
#pragma unroll
        for (int j = 0; j < 4; j++)
            hs[6][j] = k0 * (SB_BYTE(raw[6], j + 1) + SB_BYTE(raw[6], j + 7)) + k1 * (SB_BYTE(raw[6], j + 2) + SB_BYTE(raw[6], j + 6)) +
                       k2 * (SB_BYTE(raw[6], j + 3) + SB_BYTE(raw[6], j + 5)) + k3 * SB_BYTE(raw[6], j + 4);
        const int oy = y0 + r - 6;
        if (r < 6 || oy >= L.h) continue;
        // ---- blur of row oy
        uint32_t bw = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sm = k0 * (hs[0][j] + hs[6][j]) + k1 * (hs[1][j] + hs[5][j]) + k2 * (hs[2][j] + hs[4][j]) + k3 * hs[3][j];
            int v;
            if (even_round) {
                v = sm >> 16;
                const int rem = sm & 0xFFFF;
                if (rem > 0x8000 || (rem == 0x8000 && (v & 1))) v++;
            } else {
                v = (sm + 32768) >> 16;
            }
            bw |= (uint32_t)min(v, 255) << (8 * j);
        }
        uint8_t *bo = bp + (size_t)oy * L.bpitch + x;
        if (full) *(uint32_t *)bo = bw;
        else for (int j = 0; j < 4 && x + j < L.w; j++) bo[j] = (uint8_t)(bw >> (8 * j));
        // ---- FAST score of row oy (the cells only ever look at x in [19, w-19), y in [19, h-19))
        if (oy < PLF_EDGE || oy >= L.h - PLF_EDGE || x + 3 < PLF_EDGE || x >= L.w - PLF_EDGE) continue;
        uint32_t sw = 0;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            int s0, s1;
            fast_score_pair(raw, j, g.minTh, s0, s1);
            if (x + j < PLF_EDGE || x + j >= L.w - PLF_EDGE) s0 = 0;
            if (x + j + 1 < PLF_EDGE || x + j + 1 >= L.w - PLF_EDGE) s1 = 0;
            sw |= ((uint32_t)s0 | ((uint32_t)s1 << 8)) << (8 * j);
        }
        uint8_t *so = sp + (size_t)oy * L.bpitch + x;
        if (full) *(uint32_t *)so = sw;
        else for (int j = 0; j < 4 && x + j < L.w; j++) so[j] = (uint8_t)(sw >> (8 * j));
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
    // one wave per OUTPUT slot o of the frame (level-major order); its level follows from the per-level counts
    const int o = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const int *cnt = selcnt + f * g.nlevels;
    int offset = 0, total = 0, l = -1;
    for (int i = 0; i < g.nlevels; i++) {
        const int c = cnt[i];
        if (l < 0 && o < total + c) { l = i; offset = total; }
        total += c;
    }
    if (o == 0 && lane == 0) {
        n_out[f] = min(total, capacity);
        if (total > capacity) atomicOr(status, 2);
    }
    if (l < 0 || o >= capacity) return;
    const int idx = o - offset;
    const OrbLevel &L = g.lv[l];
    const uint2 s = sel[(size_t)f * g.sel_stride + L.sel_off + idx];
    const int x = (int)(s.x & 0xFFFF) + PLF_MINB, y = (int)(s.x >> 16) + PLF_MINB;  // level coordinates (integers)
    const uint8_t *img = pyr + (size_t)f * g.pyr_stride + L.plane_off + (size_t)PLF_EDGE * L.ppitch + PLF_EDGE;
    const uint8_t *center = img + (ptrdiff_t)y * L.ppitch + x;
    int m10 = 0, m01 = 0;
    if (lane < PLF_PATCH) {
        const int v = lane - PLF_HALF_PATCH;
        const int d = c_umax[v < 0 ? -v : v];
        // the row segment u = -16..15 as 8 dwords (the padded plane has 19 border pixels) and two byte dot products per
        // dword: sum (u + 16) * I and sum I over |u| <= d, so m10 = sum u * I = first - 16 * second (exact integers)
        const uint8_t *row = center + (ptrdiff_t)v * L.ppitch - 16;
        uint32_t sw = 0, si = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t px4 = *(const plf_u32u *)(row + 4 * q);
            uint32_t wgt = 0, one = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int u = 4 * q + e - 16;
                const bool in = (u < 0 ? -u : u) <= d;
                wgt |= (in ? (uint32_t)(u + 16) : 0u) << (8 * e);
                one |= (in ? 1u : 0u) << (8 * e);
            }
            sw = __builtin_amdgcn_udot4(px4, wgt, sw, false);
            si = __builtin_amdgcn_udot4(px4, one, si, false);
        }
        m10 = (int)sw - 16 * (int)si;
        m01 = v * (int)si;
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
