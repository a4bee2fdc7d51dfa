// orb_front.hip -- the fused ORB front pass for gfx950: ONE tiled kernel per pyramid level does what the reference spreads over
//   ORBextractor::ComputePyramid          (include/ORBextractor.h:89, so@0x70430: cv::resize INTER_LINEAR from the previous level + copyMakeBorder)
//   the 815 cv::FAST cell calls of ComputeKeyPointsOctTree (:90, so@0x75fa0: score, per-cell threshold / retry, 3x3 NMS)
//   GaussianBlur(7x7, sigma 2) of operator() (so@0x77487)
// A 256-thread workgroup owns a tile of 2 x 2 FAST cells of one level of one frame.  The level pixels of the tile (+ 3-pixel halo) are produced
// ONCE into LDS -- resized from the previous level's plane (itself staged through LDS with coalesced loads) or copied from the input for level 0 --
// and everything else is computed from that LDS tile before it leaves the CU:
//   * the tile's part of the padded pyramid plane (mvImagePyramid layout, REFLECT_101 border included) -- the only copy the later stages need
//     (IC_Angle reads it, the next level is resized from it);
//   * the 7x7 blur (8-bit fixed-point separable path, exact integer sums, OpenCV 3.3's column rounding), row sums as uint16 in LDS;
//   * the FAST-9/16 score, in two phases: a cheap necessary test on every pixel (two ADJACENT compass points of the ring must both be brighter
//     or both darker by the minimum threshold: every 9-arc contains such a pair), survivors compacted with wave ballots into an LDS list, and the
//     full cornerScore only for them, densely packed over the lanes;
//   * per cell (one wave each): threshold iniThFAST, retry with minThFAST when the cell stays empty, strict 3x3 non-maximum suppression inside
//     the cell's computed region, raster-ordered emission into the level's candidate pool.
// The score never reaches HBM, the pyramid is written once and read once (by the next level), the blurred plane is written once.
// Cell geometry: cell (cx, cy) of a level has the sub-image x0 = 16 + cx * wCell, width min(x0 + wCell + 6, w - 16) - x0; cv::FAST computes
// the sub-image minus a 3-pixel frame, so the computed regions of neighbouring cells abut: [19 + cx * wCell, 19 + (cx + 1) * wCell).
#include "plf_common.h"
#include "orb_geom.h"

typedef uint32_t __attribute__((aligned(1))) plf_u32u;

// plf_reflect101 for indices at most one image size outside [0, n): two selects instead of the general loop (which the compiler keeps as a data-dependent loop
// in every place it is inlined: the tile phases carried five of them); anything further out -- halo wider than a tiny level -- still takes the loop
__device__ __forceinline__ int of_reflect101(int p, int n)
{
    int q = p < 0 ? -p : p;
    q = q >= n ? 2 * (n - 1) - q : q;
    if (__builtin_expect((unsigned)q >= (unsigned)n, 0)) return plf_reflect101(p, n);
    return q;
}

// cornerScore<16> of cv::FAST (largest threshold for which the pixel is still a corner) minus 1, clamped at 0; d[k] = I_p - I_ring[k]
// (scalar form: the definition; the kernel runs the packed form orb_fast_score_pk below)
__device__ __forceinline__ int orb_fast_score(const int d[16], int t)
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

struct __attribute__((aligned(8))) OrbColTab { uint32_t sel, coef; };   // one tile column of the resize: v_perm selector of its two source bytes inside
                                                                         // the group's 8-byte window (bytes 0 and 2; 1 and 3 = zero), coefficients as short2
struct __attribute__((aligned(8))) OrbRowTab { short off, nxt, c0, c1; };   // one tile row: the two source rows inside the staged tile, coefficients
typedef short plf_s2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ plf_s2v of_min(plf_s2v a, plf_s2v b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ plf_s2v of_max(plf_s2v a, plf_s2v b) { return __builtin_elementwise_max(a, b); }
// cornerScore<16> as above on packed int16 pairs: P[k] = (d[k], d[k + 8]), k = 0..7 (|d| <= 255, t <= 255: exact in 16 bits).  With E[j] = P[j] for j < 8 and
// the half-swapped P[j - 8] for j >= 8, lane x of an expression over E[k], E[k + 1], ... is the scalar expression at ring index k and lane y the one at
// k + 8 -- half the min / max instructions of the scalar form.
__device__ __forceinline__ plf_s2v of_swap(plf_s2v a)
{
    const uint32_t u = __builtin_bit_cast(uint32_t, a);
    return __builtin_bit_cast(plf_s2v, __builtin_amdgcn_alignbit(u, u, 16));
}
__device__ __forceinline__ int orb_fast_score_pk(const plf_s2v P[8], int t)
{
    plf_s2v E[10];
#pragma unroll
    for (int k = 0; k < 8; k++) E[k] = P[k];
    E[8] = of_swap(P[0]); E[9] = of_swap(P[1]);
    // (No early exit on the scalar form's necessary condition "of every opposite pair one pixel is brighter than t / darker than -t": in a wave it saves nothing
    // unless all 64 survivors fail it, and it cost 30 packed operations + 6 swaps per survivor.  Without it a pixel that is no corner at t gets its exact score,
    // which is below t -- no 9-arc above t means max(sb, -sd) <= t -- instead of 0: the non-maximum suppression skips both alike (sc < minTh) and a corner's
    // score exceeds either; the score tile never leaves the CU.)
    (void)t;
    plf_s2v m3[14], M3[14];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m3[k] = of_min(E[k], of_min(E[k + 1], E[k + 2]));
        M3[k] = of_max(E[k], of_max(E[k + 1], E[k + 2]));
    }
#pragma unroll
    for (int k = 8; k < 14; k++) { m3[k] = of_swap(m3[k - 8]); M3[k] = of_swap(M3[k - 8]); }
    plf_s2v sb = of_min(m3[0], of_min(m3[3], m3[6])), sd = of_max(M3[0], of_max(M3[3], M3[6]));
#pragma unroll
    for (int k = 1; k < 8; k++) {
        sb = of_max(sb, of_min(m3[k], of_min(m3[k + 3], m3[k + 6])));
        sd = of_min(sd, of_max(M3[k], of_max(M3[k + 3], M3[k + 6])));
    }
    const int sbs = max((int)sb.x, (int)sb.y), sds = min((int)sd.x, (int)sd.y);
    const int s_ = max(sbs, -sds) - 1;
    return s_ < 0 ? 0 : s_;
}

// bytes k and k + 1 (0 <= k <= 10) of the 12 bytes (A, B, C) as two zero-extended int16 (one v_perm_b32); bytes k, k + 1 (k <= 2) of one dword
#define OF_PAIR(A, B, C, k) __builtin_bit_cast(plf_s2v, __builtin_amdgcn_perm((k) < 4 ? (B) : (C), (k) < 4 ? (A) : (k) < 8 ? (B) : (C), \
                                                                              (uint32_t)((k) & 3) | 0x0C000C00u | ((uint32_t)(((k) & 3) + 1) << 16)))
#define OF_PAIR1(A, k) __builtin_bit_cast(plf_s2v, __builtin_amdgcn_perm(0u, (A), (uint32_t)(k) | 0x0C000C00u | ((uint32_t)((k) + 1) << 16)))
// byte k (0..11) of the 12 bytes held in three dwords
#define OF_BYTE(A, B, C, k) ((int)((((k) < 4 ? (A) : (k) < 8 ? (B) : (C)) >> (8 * ((k) & 3))) & 0xFFu))

// Thread layout of the pixel phases: 8 rows x 32 groups of 4 tile columns per pass (tid >> 5, tid & 31): no integer division, LDS accessed as
// aligned dwords.  Tile column c <-> level x = ex0 + c with ex0 = 4 * floor(xs / 4) - 4, so groups of 4 columns are 4-aligned in the level image
// too (aligned stores to the blurred plane).
#define OF_NT PLF_ORB_LEVEL_THREADS   // threads per tile
// Register budget: 8 waves per SIMD = at most 64 VGPRs (the compiler lands on 51 without spilling a vector register; 73 uncapped).  Not for this
// kernel's own occupancy (LDS allows 6 workgroups per CU) but for co-residency: four region-growing waves hold 416 of a SIMD's 512 VGPRs for 80 ms, and
// in the 96 that are left a 56-register ORB wave fits TOGETHER with a matcher wave (40-48), an 80-register one alone: +2.5 % for the pipeline.
#ifndef PLF_ORB_LEVEL_WPE
#define PLF_ORB_LEVEL_WPE 8
#endif
#define OF_OCC __attribute__((amdgpu_waves_per_eu(PLF_ORB_LEVEL_WPE, PLF_ORB_LEVEL_WPE)))
#ifndef PLF_ORB_PRIO
#define PLF_ORB_PRIO 2
#endif
__global__ void OF_OCC __launch_bounds__(OF_NT) k_orb_level(const uint8_t *__restrict__ in, ptrdiff_t in_pitch, ptrdiff_t in_fstride, uint8_t *__restrict__ pyr,
                                                   uint8_t *__restrict__ blur, int l, const int *__restrict__ xofs, const short2 *__restrict__ xa,
                                                   const int *__restrict__ yofs, const short2 *__restrict__ yb, const int4 *__restrict__ cells,
                                                   int2 *__restrict__ cellinfo, uint2 *__restrict__ pool, int *__restrict__ poolcnt,
                                                   int *__restrict__ status, OrbGeom g, int4 taps)
{
    // issue priority above the other throughput kernels (matchers, k_lsd_pre, NFA stages: 0), below the region chain (3): the tile kernel is the longest
    // of the co-runners and latency-bound per tile; 0 -> 2: 134.5 -> 131.8 ms per 4096-frame step (3: the same; above the region waves: 132.8)
    __builtin_amdgcn_s_setprio(PLF_ORB_PRIO);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_nlist;
    __shared__ unsigned long long s_mask[4][64];   // per cell of the tile, per row of its computed region: the NMS maxima of the pass that emits the cell (a cell that
                                                     // needs the second pass has none from the first)
    const OrbLevel &L = g.lv[l];
    const int tid = threadIdx.x, f = blockIdx.y;
    const int trow = tid >> 5, tc4 = (tid & 31) * 4;
    constexpr int NR = OF_NT / 32;   // tile rows per pass of the pixel phases
    // (An XCD-aware order -- giving each of the 8 XCDs a contiguous run of the frame's tiles so that neighbours share halo reads and merge their
    // 60-byte row segments in one L2 -- was measured: 3 % SLOWER solo, 30.8 vs 29.8 ms per 4096 frames; the plain raster order stays.)
    const int tile = (int)blockIdx.x;
    const int tx = tile % L.tcx, ty = tile / L.tcx;
    const int W = L.w, H = L.h;
    // ---- tile geometry
    const int cx0 = 2 * tx, cx1 = min(cx0 + 2, L.ncx), cy0 = 2 * ty, cy1 = min(cy0 + 2, L.ncy);
    const int rx0 = PLF_EDGE + cx0 * L.wCell, rx1 = cx1 == L.ncx ? L.rex : PLF_EDGE + cx1 * L.wCell;   // bounding box of the cells' computed regions
    const int ry0 = PLF_EDGE + cy0 * L.hCell, ry1 = cy1 == L.ncy ? L.rey : PLF_EDGE + cy1 * L.hCell;
    // where the second cell column / row of the tile starts -- clamped: the LAST cell's sub-image is cut at the level border (maxBorder), which can leave
    // the cell before it a shorter computed region than wCell / hCell and the last one none at all
    const int xm = cx1 - cx0 == 2 ? min(rx0 + L.wCell, rx1) : rx1, ym = cy1 - cy0 == 2 ? min(ry0 + L.hCell, ry1) : ry1;
    const int xs = tx == 0 ? 0 : rx0, xe = tx == L.tcx - 1 ? W : rx1;                                   // owned part of the level image
    const int ys = ty == 0 ? 0 : ry0, ye = ty == L.tcy - 1 ? H : ry1;
    const int ex0 = (xs & ~3) - 4, EW = ((xe - 1) & ~3) + 8 - ex0;                                     // tile columns: owned + halo, 4-aligned
    const int xs4 = xs & ~3, xe4 = tx == L.tcx - 1 ? xe : xe & ~3;                                     // the columns this tile WRITES (plane, blur): inner boundaries 4-aligned
    const int ey0 = ys - 3, EH = ye - ys + 6;                                                          // tile rows: owned + 3-row halo
    const int PW = g.lds_pw;
    uint8_t *P = smem;
    uint8_t *SRC = smem + g.lds_off_a;
    uint16_t *LIST = reinterpret_cast<uint16_t *>(smem + g.lds_off_a);   // FAST survivors (the staged source is dead by then)
    uint8_t *S = smem + g.lds_off_s;
    OrbColTab *XT = reinterpret_cast<OrbColTab *>(smem + g.lds_off_tab);
    OrbRowTab *YT = reinterpret_cast<OrbRowTab *>(XT + PW);
    short *GM = reinterpret_cast<short *>(YT + g.lds_eh);   // per column group: first source byte of its window (negative: -1 - first, window wider than 8)
    if (tid == 0) s_nlist = 0;
    for (int i = tid; i < 4 * 64; i += OF_NT) (&s_mask[0][0])[i] = 0ull;
    // ---- 1. the level pixels of the tile
    if (l == 0) {
        const uint8_t *img = in + (size_t)f * in_fstride;
        // (a thread owns a column group and walks down the rows, as in the plane write below: the column test and the mirrored columns are settled once)
        // Round 6: the threads are dealt to (column group, row slot) by the tile's own group count -- 240 of 256 lanes busy on a VGA tile instead of the 20 of every
        // 32 that the fixed 32-groups-per-row layout used -- and the rows inside the image step a pointer; only rows mirrored at the image border take the
        // REFLECT_101 path (10.6 vector lane-instructions per pixel for this COPY before, profiles/r06_orb_phase_insts.txt)
        {
            const int ng0 = EW >> 2, nslots = OF_NT / ng0;
            const uint32_t rcp0 = 0xFFFFFFFFu / (uint32_t)ng0 + 1u;
            const int slot = ng0 > 1 ? (int)__umulhi((uint32_t)tid, rcp0) : tid, cg = tid - slot * ng0;
            if (slot < nslots) {
                const int c4 = cg * 4, x = ex0 + c4;
                const bool whole = x >= 0 && x + 3 < W;
                int mx[4] = {x, x + 1, x + 2, x + 3};
                if (!whole) {
#pragma unroll
                    for (int j = 0; j < 4; j++) mx[j] = of_reflect101(x + j, W);
                }
                const int in0 = max(0, -ey0), in1 = min(EH, H - ey0);   // tile rows [in0, in1) lie inside the image
                int ey = slot;
                if (ey < in0) ey += (in0 - ey + nslots - 1) / nslots * nslots;
                const uint8_t *row = img + (size_t)(ey0 + ey) * in_pitch;
                uint8_t *dst = P + ey * PW + c4;
                const size_t rstep = (size_t)nslots * in_pitch;
                if (whole) {
                    row += x;
                    for (; ey < in1; ey += nslots, row += rstep, dst += nslots * PW) *reinterpret_cast<uint32_t *>(dst) = *(const plf_u32u *)row;
                } else {
                    for (; ey < in1; ey += nslots, row += rstep, dst += nslots * PW)
                        *reinterpret_cast<uint32_t *>(dst) = (uint32_t)row[mx[0]] | ((uint32_t)row[mx[1]] << 8) | ((uint32_t)row[mx[2]] << 16) | ((uint32_t)row[mx[3]] << 24);
                }
                if (in0 > 0 || in1 < EH) {
                    for (int e2 = slot; e2 < EH; e2 += nslots) {
                        if (e2 >= in0 && e2 < in1) continue;
                        const uint8_t *r2 = img + (size_t)of_reflect101(ey0 + e2, H) * in_pitch;
                        *reinterpret_cast<uint32_t *>(P + e2 * PW + c4) = (uint32_t)r2[mx[0]] | ((uint32_t)r2[mx[1]] << 8) | ((uint32_t)r2[mx[2]] << 16) | ((uint32_t)r2[mx[3]] << 24);
                    }
                }
            }
        }
    } else {
        const OrbLevel &SL = g.lv[l - 1];
        const int SPW = g.lds_spw;
        // level coordinates the tile needs (mirrored halo coordinates fall inside this range), and the source rectangle behind them
        const int lx_lo = max(ex0, 0), lx_hi = min(ex0 + EW - 1, W - 1), ly_lo = max(ey0, 0);
        const int sx_lo = xofs[L.tabx_off + lx_lo] & ~3, sx_hi = min(xofs[L.tabx_off + lx_hi] + 1, SL.w - 1);
        const int sy_lo = min(max(yofs[L.taby_off + ly_lo], 0), SL.h - 1);   // row origin of the row table below
        const int SWt = sx_hi - sx_lo + 1;
        const uint8_t *src = pyr + (size_t)f * g.pyr_stride + SL.plane_off + (size_t)PLF_EDGE * SL.ppitch + PLF_EDGE;
        const int ngrp = EW >> 2;
        for (int i = tid; i < ngrp + EH; i += OF_NT) {
            if (i < ngrp) {   // one column group: its 4 table entries relative to the group's first source byte
                int off[4], nxt[4];
                uint32_t cf[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int X = of_reflect101(ex0 + 4 * i + j, W);
                    const int sx = xofs[L.tabx_off + X];
                    const short2 a = xa[L.tabx_off + X];
                    off[j] = sx - sx_lo; nxt[j] = min(sx + 1, SL.w - 1) - sx_lo;
                    cf[j] = (uint32_t)(uint16_t)a.x | ((uint32_t)(uint16_t)a.y << 16);
                }
                const int mn = min(min(off[0], off[1]), min(off[2], off[3])), mx = max(max(nxt[0], nxt[1]), max(nxt[2], nxt[3]));
                const bool wide = mx - mn > 7;
                GM[i] = (short)(wide ? -1 - mn : mn);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    OrbColTab t;
                    t.coef = cf[j];
                    t.sel = wide ? ((uint32_t)off[j] | ((uint32_t)nxt[j] << 16)) : ((uint32_t)(off[j] - mn) | 0x0C000C00u | ((uint32_t)(nxt[j] - mn) << 16));
                    XT[4 * i + j] = t;
                }
            } else {
                const int Y = of_reflect101(ey0 + (i - ngrp), H);
                const int sy = yofs[L.taby_off + Y];
                const short2 b = yb[L.taby_off + Y];
                OrbRowTab t;
                t.off = (short)(min(max(sy, 0), SL.h - 1) - sy_lo); t.nxt = (short)(min(max(sy + 1, 0), SL.h - 1) - sy_lo); t.c0 = b.x; t.c1 = b.y;
                YT[i - ngrp] = t;
            }
        }
        // cv::resize INTER_LINEAR 8UC1 (OpenCV 3.3): dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, 11-bit coefficients.
        // 4 outputs per thread: per source row one 8-byte window (three aligned LDS dwords, v_alignbyte), per output v_perm picks its two
        // bytes as int16 lanes and v_dot2 multiplies them with the (a0, a1) pair.
        // The source rows are staged in g.lds_parts passes (the rows of the tile split evenly; orb_part_rows is the host's arithmetic): the staging buffer is
        // what decides how many tiles a CU holds, and the kernel's run time is inversely proportional to that number.
        const uint32_t rcp_g = 0xFFFFFFFFu / (uint32_t)ngrp + 1u;   // i / ngrp = umulhi(i, rcp_g) for i < 65536
        for (int part = 0; part < g.lds_parts; part++) {
        int ps_lo, ps_hi;
        orb_part_rows(ey0, EH, g.lds_parts, part, H, SL.h, yofs + L.taby_off, &ps_lo, &ps_hi);
        const int e0 = part * EH / g.lds_parts, e1 = (part + 1) * EH / g.lds_parts, SHt = ps_hi - ps_lo + 1;
        if (part > 0) __syncthreads();   // the previous part's rows have been consumed
        for (int r = trow; r < SHt; r += NR)
            for (int c4 = tc4; c4 < SWt + 8; c4 += 128)   // (+8: the 12-byte windows below may read past the last needed byte; the padded plane has them)
                *reinterpret_cast<uint32_t *>(SRC + r * SPW + c4) = *(const plf_u32u *)(src + (size_t)(ps_lo + r) * SL.ppitch + sx_lo + c4);
        __syncthreads();                 // (first part: the tables above as well)
        // Round 6: a thread owns one column group for a BAND of consecutive tile rows and walks down them.  What depends on the column group only -- its four table
        // entries, the window offset -- is loaded once, and consecutive output rows share a source row (scale 1.2: the lower source row of output row y is the upper one
        // of row y + 1 five times out of six): its horizontal pass (one v_perm + one v_dot2 per output, already >> 4) stays in registers.  The arithmetic is
        // unchanged; (c * x) >> 16 of the vertical pass is one v_mul_hi_u32 with the coefficient held as c << 16 (c <= 2048, x < 2^15: no overflow).
        // Before: every (row, group) item reloaded the tables and ran both horizontal passes -- 100 vector instructions per 4 outputs, 64 M of the kernel's 306 M per
        // 1024-frame launch (profiles/r06_orb_phase_insts.txt).
        {
            const int nrows = e1 - e0, tpg = OF_NT / ngrp, bh = (nrows + tpg - 1) / tpg;
            const int band = ngrp > 1 ? (int)__umulhi((uint32_t)tid, rcp_g) : tid, cg = tid - band * ngrp;
            const int rA = e0 + band * bh, rB = min(rA + bh, e1);
            if (band < tpg && rA < rB) {
                const int c4 = cg * 4;
                OrbColTab t[4];
                *reinterpret_cast<uint4 *>(&t[0]) = *reinterpret_cast<const uint4 *>(&XT[c4]);
                *reinterpret_cast<uint4 *>(&t[2]) = *reinterpret_cast<const uint4 *>(&XT[c4 + 2]);
                const int gm = GM[cg];
                uint8_t *pout = P + rA * PW + c4;
                const int srow0 = sy_lo - ps_lo;
                if (gm >= 0) {
                    const int base = gm & ~3, sh = gm & 3;
                    const uint8_t *sb = SRC + base;
                    int prev = -0x7fffffff;
                    uint32_t hp[4] = {0u, 0u, 0u, 0u};
                    auto hpass = [&](int srow, uint32_t hx[4]) {
                        const uint8_t *r = sb + srow * SPW;
                        const uint32_t a0 = *reinterpret_cast<const uint32_t *>(r), a1 = *reinterpret_cast<const uint32_t *>(r + 4), a2 = *reinterpret_cast<const uint32_t *>(r + 8);
                        const uint32_t alo = __builtin_amdgcn_alignbyte(a1, a0, sh), ahi = __builtin_amdgcn_alignbyte(a2, a1, sh);
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            hx[j] = (uint32_t)(__builtin_amdgcn_sdot2(__builtin_bit_cast(plf_s2v, __builtin_amdgcn_perm(ahi, alo, t[j].sel)), __builtin_bit_cast(plf_s2v, t[j].coef), 0, false) >> 4);
                    };
                    for (int ey = rA; ey < rB; ey++, pout += PW) {
                        const OrbRowTab ty_ = YT[ey];
                        const int s0 = ty_.off + srow0, s1 = ty_.nxt + srow0;
                        uint32_t h0[4], h1[4];
                        if (s0 != prev) hpass(s0, h0);
                        else {
#pragma unroll
                            for (int j = 0; j < 4; j++) h0[j] = hp[j];
                        }
                        hpass(s1, h1);
                        const uint32_t cs0 = (uint32_t)(uint16_t)ty_.c0 << 16, cs1 = (uint32_t)(uint16_t)ty_.c1 << 16;
                        uint32_t out = 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            out |= ((__umulhi(cs0, h0[j]) + __umulhi(cs1, h1[j]) + 2u) >> 2) << (8 * j);   // (<= 255: the coefficients of a pair sum to 2048)
                            hp[j] = h1[j];
                        }
                        prev = s1;
                        *reinterpret_cast<uint32_t *>(pout) = out;
                    }
                } else {
                    for (int ey = rA; ey < rB; ey++, pout += PW) {
                        const OrbRowTab ty_ = YT[ey];
                        const uint8_t *r0 = SRC + (ty_.off + srow0) * SPW, *r1 = SRC + (ty_.nxt + srow0) * SPW;
                        uint32_t out = 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int o0 = (int)(t[j].sel & 0xFFFF), o1 = (int)(t[j].sel >> 16), c0 = (short)(t[j].coef & 0xFFFF), c1 = (short)(t[j].coef >> 16);
                            const int sa = r0[o0] * c0 + r0[o1] * c1;
                            const int sb_ = r1[o0] * c0 + r1[o1] * c1;
                            out |= (uint32_t)(((((ty_.c0 * (sa >> 4)) >> 16) + ((ty_.c1 * (sb_ >> 4)) >> 16) + 2) >> 2) & 0xFF) << (8 * j);
                        }
                        *reinterpret_cast<uint32_t *>(pout) = out;
                    }
                }
            }
        }
        }
    }
    __syncthreads();
#if defined(OF_STOP) && OF_STOP <= 1
    return;
#endif
    // ---- 2. this tile's part of the padded plane (interior + REFLECT_101 border)
    {
        // columns written by this tile: its owned range with the inner boundaries rounded DOWN to multiples of 4 (the tile holds those level
        // pixels in its halo, the left neighbour stops there as well), so that every group of an inner tile is a whole dword
        const int pxs = tx == 0 ? 0 : xs4 + PLF_EDGE, pxe = tx == L.tcx - 1 ? L.ppitch : xe4 + PLF_EDGE;
        const int pys = ty == 0 ? 0 : ys + PLF_EDGE, pye = ty == L.tcy - 1 ? H + 2 * PLF_EDGE : ye + PLF_EDGE;
        uint8_t *plane = pyr + (size_t)f * g.pyr_stride + L.plane_off;
        const int ppitch = L.ppitch;
        // groups of 4 level columns x4 = 4-aligned, from the one holding plane column pxs to the one holding pxe - 1
        const int xg0 = (pxs - PLF_EDGE) & ~3;
        // A thread owns one group of 4 columns and walks down the rows: what depends on the column only -- whether the group is a plain dword or
        // a partial / mirrored one, the 4 source columns and their validity -- is settled once, a row costs one LDS read, one store and the two
        // pointer steps.  (With the rows outside, every thread redid the column tests and the 64-bit row arithmetic per dword: 45 instructions per
        // stored dword; this phase issued as many VALU instructions as the resize or the blur -- SQ_INSTS_VALU of cut builds, 71 M of 416 M per
        // launch.)
        // (measured alternatives for the stores themselves: 16-byte stores at byte alignment 29.8 -> 41.9 ms per 4096 frames; a plane layout that
        // makes these dword stores aligned -- pitch rounded to 64, one pad byte in front of every row -- changes nothing: 29.8 ms; writing the mirrored
        // border columns as byte-swapped dwords instead of single bytes: 30.4 ms)
        // (round 6: threads dealt to (column group, row slot) by the tile's own group count, as in the level-0 copy above)
        const int ngp = (pxe - PLF_EDGE - xg0 + 3) >> 2, NRp = OF_NT / ngp;
        const uint32_t rcpp = 0xFFFFFFFFu / (uint32_t)ngp + 1u;
        const int pslot = ngp > 1 ? (int)__umulhi((uint32_t)tid, rcpp) : tid;
        for (int x4 = xg0 + 4 * (tid - pslot * ngp); pslot < NRp && x4 + PLF_EDGE < pxe; x4 += 4 * ngp) {
            const bool full = x4 >= 0 && x4 + 3 < W && x4 + PLF_EDGE >= pxs && x4 + 3 + PLF_EDGE < pxe;
            int sc[4] = {x4 - ex0, x4 + 1 - ex0, x4 + 2 - ex0, x4 + 3 - ex0};
            bool ok[4] = {true, true, true, true};
            if (!full) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    ok[j] = x4 + j + PLF_EDGE >= pxs && x4 + j + PLF_EDGE < pxe;
                    sc[j] = of_reflect101(x4 + j, W) - ex0;
                }
            }
            uint8_t *dcol = plane + PLF_EDGE + x4;
            // Round 6: the rows inside the level (all rows of a tile that is not in the first / last tile row) need no mirror test: one LDS read, one store and two
            // pointer steps per row; only the border rows above / below the level go through the REFLECT_101 index (a data-dependent loop in machine code, which
            // the row loop used to carry for every row: 43 vector instructions per stored dword, profiles/r06_orb_phase_insts.txt)
            const int pin0 = max(pys, PLF_EDGE), pin1 = min(pye, H + PLF_EDGE);   // plane rows [pin0, pin1) hold level rows [pin0 - 19, pin1 - 19)
            {
                int py = pys + pslot;
                if (py < pin0) py += (pin0 - py + NRp - 1) / NRp * NRp;   // first row of this thread inside the level
                const uint8_t *prow = P + (py - PLF_EDGE - ey0) * PW;
                uint8_t *d = dcol + (size_t)py * ppitch;
                const size_t dstep = (size_t)NRp * ppitch;
                if (full) {
                    prow += x4 - ex0;
                    for (; py < pin1; py += NRp, prow += NRp * PW, d += dstep) *(plf_u32u *)d = *reinterpret_cast<const uint32_t *>(prow);
                } else {
                    for (; py < pin1; py += NRp, prow += NRp * PW, d += dstep) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (ok[j]) d[j] = prow[sc[j]];
                    }
                }
            }
            if (pys < pin0 || pye > pin1) {   // (first / last tile row only)
                for (int py = pys + pslot; py < pye; py += NRp) {
                    if (py >= pin0 && py < pin1) continue;
                    const int ly = py - PLF_EDGE;
                    const uint8_t *prow = P + (of_reflect101(ly, H) - ey0) * PW;
                    uint8_t *d = dcol + (size_t)py * ppitch;
                    if (full) *(plf_u32u *)d = *reinterpret_cast<const uint32_t *>(prow + (x4 - ex0));
                    else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (ok[j]) d[j] = prow[sc[j]];
                    }
                }
            }
        }
    }
#if defined(OF_STOP) && OF_STOP <= 2
    return;
#endif
    // ---- 3. GaussianBlur 7x7.  A thread owns one group of 4 columns for a segment of 8 output rows and walks down the 14 tile rows behind
    // them: row sums by two byte dot products per pixel (taps 18 34 49 55 fit a byte; v_alignbyte lines the 4-byte windows up), the 7 live
    // rows of sums stay in registers (fully unrolled: no window shifting), exact column sums, rounded as OpenCV 3.3's column filter does.
    // Round 6: the column sums run in fp32.  On gfx950 v_add_f32 / v_fma_f32 issue at twice the rate of the integer multiply-adds, shifts, min and bit-field
    // forms (profiles/r03_valu_issue.json: 2.2 against 4.2 cycles per SIMD), and the arithmetic is EXACT: a row sum is an integer <= 255 * 257, the taps are
    // scaled by 2^-16 (a power of two), so every partial sum is an integer multiple of 2^-16 that needs at most 24 bits while the final sum stays below 2^24 * 2^-16
    // = 256 -- and a sum of 256 or more is clamped to 255 whatever its last bits are (fp32 rounding cannot carry a value across 256 = 2^8 downwards).  The vector
    // rule of SymmColumnVec_32s8u, sum / 65536 rounded half to even, is what (t + 2^23) - 2^23 computes in the default rounding mode; v_cvt_pk_u8_f32 converts
    // the integer-valued float, saturates and places the byte.  The w % 4 tail columns of the image (half-up rule) keep the integer path.
    {
        const uint32_t K0123 = (uint32_t)taps.x | ((uint32_t)taps.y << 8) | ((uint32_t)taps.z << 16) | ((uint32_t)taps.w << 24);
        const uint32_t K210 = (uint32_t)taps.z | ((uint32_t)taps.y << 8) | ((uint32_t)taps.x << 16);
        const int k0 = taps.x, k1 = taps.y, k2 = taps.z, k3 = taps.w;
        const float kf0 = (float)k0 * (1.f / 65536.f), kf1 = (float)k1 * (1.f / 65536.f), kf2 = (float)k2 * (1.f / 65536.f), kf3 = (float)k3 * (1.f / 65536.f);
        float magic = 8388608.f;          // 2^23; held in a VGPR: as a 32-bit literal it would halve the issue rate of the two additions that use it
        asm volatile("" : "+v"(magic));
        const int OH = ye - ys, ngb = (EW >> 2) - 2, nseg = (OH + 7) >> 3, wvec = W & ~3;
        uint8_t *bp = blur + (size_t)f * g.blur_stride + L.blur_off;
        const int bpitch = L.bpitch;   // (a local: read through the reference it is re-loaded from the kernel arguments behind every store)
        for (int it = tid; it < ngb * nseg; it += OF_NT) {
            const int seg = it / ngb, c4 = 4 + 4 * (it - seg * ngb), oy0 = seg * 8, x4 = ex0 + c4;
#ifndef OF_BLUR_INT
            if (x4 + 3 < wvec) {
                float hf[14][4];
#pragma unroll
                for (int r = 0; r < 14; r++) {
                    const int ey = min(oy0 + r, EH - 1);   // (rows past the tile only feed outputs that are not stored)
                    const uint32_t *p = reinterpret_cast<const uint32_t *>(P + ey * PW + c4 - 4);
                    const uint32_t A = p[0], B = p[1], C = p[2];   // level x - 4 .. x + 7 of the group's first pixel x
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t lo4 = j == 3 ? B : __builtin_amdgcn_alignbyte(B, A, j + 1);   // bytes x + j - 3 .. x + j
                        const uint32_t hi4 = j == 3 ? C : __builtin_amdgcn_alignbyte(C, B, j + 1);   // bytes x + j + 1 .. x + j + 4 (the last has tap 0)
                        hf[r][j] = (float)__builtin_amdgcn_udot4(hi4, K210, __builtin_amdgcn_udot4(lo4, K0123, 0u, false), false);
                    }
                    if (r < 6) continue;
                    const int oy = oy0 + r - 6;
                    if (oy >= OH) continue;
                    uint32_t bw = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float t = kf0 * (hf[r - 6][j] + hf[r][j]);
                        t = __builtin_fmaf(kf1, hf[r - 5][j] + hf[r - 1][j], t);
                        t = __builtin_fmaf(kf2, hf[r - 4][j] + hf[r - 2][j], t);
                        t = __builtin_fmaf(kf3, hf[r - 3][j], t);
                        t = (t + magic) - magic;   // sum / 65536, half to even (t < 2^9)
                        bw = __builtin_amdgcn_cvt_pk_u8_f32(t, j, bw);   // (saturates at 255)
                    }
                    uint8_t *bo = bp + (size_t)(ys + oy) * bpitch + x4;
                    if (x4 >= xs4 && x4 + 3 < xe4) *reinterpret_cast<uint32_t *>(bo) = bw;
                    else
                        for (int j = 0; j < 4; j++)
                            if (x4 + j >= xs4 && x4 + j < xe4) bo[j] = (uint8_t)(bw >> (8 * j));
                }
                continue;
            }
#endif
            // rounding rule per column, settled once per item: 0 = the vector loop's half-to-even, 1 = the scalar tail's half-up (below: branch-free)
            int tail[4];
#pragma unroll
            for (int j = 0; j < 4; j++) tail[j] = x4 + j < wvec ? 0 : 1;
            int hs[14][4];
#pragma unroll
            for (int r = 0; r < 14; r++) {
                const int ey = min(oy0 + r, EH - 1);   // (rows past the tile only feed outputs that are not stored)
                const uint32_t *p = reinterpret_cast<const uint32_t *>(P + ey * PW + c4 - 4);
                const uint32_t A = p[0], B = p[1], C = p[2];   // level x - 4 .. x + 7 of the group's first pixel x
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t lo4 = j == 3 ? B : __builtin_amdgcn_alignbyte(B, A, j + 1);   // bytes x + j - 3 .. x + j
                    const uint32_t hi4 = j == 3 ? C : __builtin_amdgcn_alignbyte(C, B, j + 1);   // bytes x + j + 1 .. x + j + 4 (the last has tap 0)
                    hs[r][j] = (int)__builtin_amdgcn_udot4(hi4, K210, __builtin_amdgcn_udot4(lo4, K0123, 0u, false), false);
                }
                if (r < 6) continue;
                const int oy = oy0 + r - 6;
                if (oy >= OH) continue;
                uint32_t bw = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (24-bit multiplies: a row sum is at most 255 * 256, a pair of them 17 bits, a tap 6 -- v_mad_u32_u24 chains instead of v_mul_lo_u32 / v_mad_u64_u32)
                    const int sm = (int)(__umul24(k0, hs[r - 6][j] + hs[r][j]) + __umul24(k1, hs[r - 5][j] + hs[r - 1][j]) + __umul24(k2, hs[r - 4][j] + hs[r - 2][j]) +
                                         __umul24(k3, hs[r - 3][j]));
                    // SymmColumnVec_32s8u: sum / 65536 rounded half to even for x < (w & ~3); its scalar tail ((sum + 32768) >> 16) for the last w % 4 columns
                    // (sm + 0x7FFF + bit 16 of sm) >> 16 in the vector columns, (sm + 0x8000) >> 16 in the tail: one expression, no EXEC region per pixel
                    const int v = (sm + 0x7FFF + (((sm >> 16) & 1) | tail[j])) >> 16;
                    bw |= (uint32_t)min(v, 255) << (8 * j);
                }
                uint8_t *bo = bp + (size_t)(ys + oy) * bpitch + x4;
                if (x4 >= xs4 && x4 + 3 < xe4) *reinterpret_cast<uint32_t *>(bo) = bw;
                else
                    for (int j = 0; j < 4; j++)
                        if (x4 + j >= xs4 && x4 + j < xe4) bo[j] = (uint8_t)(bw >> (8 * j));
            }
        }
    }
#if defined(OF_STOP) && OF_STOP <= 3
    return;
#endif
    // ---- 4-6. FAST-9/16 of the cells: pre-test, score, 3x3 non-maximum suppression, emission -- the reference's TWO calls per cell (so@0x763d4, so@0x76753) as two
    // passes over the tile (round 6).  Pass A runs everything at iniThFAST for all cells: the pre-test passes 18 % of the pixels instead of 28 % at minThFAST (44 instead
    // of 80 % on natural-image-like frames, tools/experiments/README.md), and every later stage -- the exact score of the survivors (the largest phase), the neighbourhood
    // test -- scales with that.  A cell with a maximum at iniThFAST is emitted at once, as the reference's first call does.  Pass B, only for the cells that stayed
    // empty (12 % on the polygon scenes) and only if the tile has any: the same three stages at minThFAST restricted to those cells.  Results are the same bits: a score
    // is exact whatever threshold admitted the pixel (orb_fast_score_pk), a maximum >= iniThFAST beats every neighbour below iniThFAST whether that neighbour carries
    // its exact score or 0, and non-maximum suppression never looks across a cell border.
    //   pre-test: a thread owns one group of 4 columns for a band of rows and walks down them; a bright 9-arc needs two ADJACENT compass points of the ring with
    //     d > t, a dark one two with d < -t; two pixels per instruction in packed int16 lanes; the four flags come off the sign bits (bits 0, 16: first pixel pair; 1,
    //     17: second), survivors take their LIST slots with one LDS atomic per group (LIST's order is free);
    //   score: cornerScore<16> of every survivor, densely packed over the lanes (S by position);
    //   suppression: each wave walks its chunks of LIST, queues the survivors whose score reaches the pass's threshold (128-entry queue of its own in the dead table
    //     area) and tests 64 queued corners at a time, branch-free (the eight scores are read unconditionally -- the bytes around the score tile are valid LDS -- and
    //     masked with the four "inside the cell" flags); maxima are recorded as one bit per (cell, row, column);
    //   emission: one wave per cell, lane = row; prefix count by a DPP scan, raster order.
    const int RH = ry1 - ry0;
    const int SP = g.lds_sp, cS0 = (rx0 - ex0) & ~3;   // score tile: pitch and first tile column (the computed regions only)
    __shared__ int s_need;
    for (int i = tid; i < ((SP * RH + 3) >> 2); i += OF_NT) reinterpret_cast<uint32_t *>(S)[i] = 0u;
    if (tid == 0) s_need = 0;
    __syncthreads();   // (every thread is done with the staged source: LIST aliases it)
    const int tmin = g.minTh, tini = g.iniTh;
    const int wv = tid >> 6, lane = tid & 63;
    const uint32_t lds_nlist = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)&s_nlist;
    // pre-test geometry of this thread: column group, band of rows, the pixels of the group inside the computed regions split by cell column
    const int cA = (rx0 - ex0) & ~3, cB = (rx1 - 1 - ex0) & ~3, ngr = ((cB - cA) >> 2) + 1;
    const uint32_t rcp_r = 0xFFFFFFFFu / (uint32_t)ngr + 1u;
    const int ptpg = OF_NT / ngr, pbh = (RH + ptpg - 1) / ptpg;
    const int pband = ngr > 1 ? (int)__umulhi((uint32_t)tid, rcp_r) : tid, pc4 = cA + (tid - pband * ngr) * 4;
    const int prA = pband * pbh, prB = pband < ptpg ? min(prA + pbh, RH) : prA;
    uint32_t cmL, cmR;   // flag positions (0, 16, 1, 17 = pixel 0, 1, 2, 3) of the group's pixels in the left / right cell column
    {
        const int lo = rx0 - (ex0 + pc4), hi = rx1 - (ex0 + pc4), mid = xm - (ex0 + pc4);   // first pixel of the right cell column, relative to the group
        const uint32_t in4 = (hi >= 4 ? 15u : (1u << max(hi, 0)) - 1u) & ~((1u << min(max(lo, 0), 4)) - 1u);
        const uint32_t l4 = in4 & ((1u << min(max(mid, 0), 4)) - 1u), r4 = in4 & ~l4;
        cmL = (l4 & 1u) | ((l4 & 2u) << 15) | ((l4 & 4u) >> 1) | ((l4 & 8u) << 14);
        cmR = (r4 & 1u) | ((r4 & 2u) << 15) | ((r4 & 4u) >> 1) | ((r4 & 8u) << 14);
    }
    // LIST holds half the pixels of the largest computed region (g.lds_list_cap entries, the host's arithmetic): enough for any HALF of a tile's rows.  A group whose
    // slots would end past the capacity writes nothing -- the count still grows, and a pass that finds more survivors than the list holds redoes the tile in two row
    // halves (fast_pass below; dense noise only).  The list is what decided the LDS of a tile: with all pixels (10 KB at VGA) 5 tiles fit a CU, with half of them 6 --
    // ORB extractor 6.69 -> 6.36 ms per 1024 frames (tools/orb_ab_libs.sh r6occ6).
    const int list_cap = g.lds_list_cap;
    auto pretest = [&](int t, uint32_t cells_on, int rlo, int rhi) {   // rows [rlo, rhi) of the region
        const plf_s2v tt = {(short)t, (short)t};
        const int ymr = min(max(ym - ry0, 0), RH);   // first row of the lower cell row, relative to the region
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const uint32_t cm = ((cells_on >> (2 * half)) & 1u ? cmL : 0u) | ((cells_on >> (2 * half + 1)) & 1u ? cmR : 0u);
            const int r0 = max(half ? max(prA, ymr) : prA, rlo), r1 = min(half ? prB : min(prB, ymr), rhi);
            if (cm == 0u || r0 >= r1) continue;
            const uint8_t *prow = P + (ry0 + r0 - ey0) * PW + pc4;
            for (int ry = r0; ry < r1; ry++, prow += PW) {
                const uint32_t Lw = *reinterpret_cast<const uint32_t *>(prow - 4), C = *reinterpret_cast<const uint32_t *>(prow),
                               Rw = *reinterpret_cast<const uint32_t *>(prow + 4), N = *reinterpret_cast<const uint32_t *>(prow + 3 * PW),
                               Sd = *reinterpret_cast<const uint32_t *>(prow - 3 * PW);
                uint32_t sg[2];
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    // d = ring - centre for the compass points 0 (row + 3), 4 (x + 3), 8 (row - 3), 12 (x - 3)
                    const plf_s2v v = OF_PAIR(Lw, C, Rw, 4 + j);
                    const plf_s2v dn = OF_PAIR1(N, j) - v, de = OF_PAIR(Lw, C, Rw, 7 + j) - v, ds = OF_PAIR1(Sd, j) - v, dw = OF_PAIR(Lw, C, Rw, 1 + j) - v;
                    // max over the four ADJACENT pairs of min(pair) = min(max(dn, ds), max(de, dw)): min distributes over max, and every point of {n, s} is
                    // adjacent to every point of {e, w} on the 4-cycle (dk: dually)
                    const plf_s2v br = of_min(of_max(dn, ds), of_max(de, dw));
                    const plf_s2v dk = of_max(of_min(dn, ds), of_min(de, dw));
                    sg[j >> 1] = __builtin_bit_cast(uint32_t, of_min(tt - br, tt + dk));   // sign bit set <=> br > t or dk < -t: possible corner
                }
                const uint32_t poss = (((sg[0] >> 15) & 0x10001u) | ((sg[1] >> 14) & 0x20002u)) & cm;
                if (poss) {
                    // (inline asm: the compiler's atomic optimizer turns a divergent atomicAdd into a scalar loop over the active lanes -- ~9 instructions per lane)
                    int base;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(lds_nlist), "v"(__popc(poss)) : "memory");
                    const uint32_t e0 = (uint32_t)(pc4 | (ry << 8));
                    uint16_t *Lp = LIST + base;
                    if (base + 4 > list_cap) continue;   // (no room for a whole group: nothing is written, the pass is redone in halves)
                    if (poss & 1u) Lp[0] = (uint16_t)e0;
                    if (poss & 2u) Lp[poss & 1u] = (uint16_t)(e0 + 2);
                    if (poss & 0x10000u) Lp[__popc(poss & 3u)] = (uint16_t)(e0 + 1);
                    if (poss & 0x20000u) Lp[__popc(poss & 0x10003u)] = (uint16_t)(e0 + 3);
                }
            }
        }
    };
    auto score = [&](int nl) {
        for (int k = tid; k < nl; k += OF_NT) {
            const int q = LIST[k], c = q & 255, ry = q >> 8;
            const uint8_t *p = P + (ry0 + ry - ey0) * PW + c;
            const int v = p[0];
            // ring pixel k and k + 8 packed as (low, high) int16, subtracted from (v, v) by one v_pk_sub_i16
            const plf_s2v vv = {(short)v, (short)v};
#define OF_RP_(a, b) (vv - __builtin_bit_cast(plf_s2v, (uint32_t)(a) | ((uint32_t)(b) << 16)))
            plf_s2v Pk[8];
            Pk[0] = OF_RP_(p[3 * PW], p[-3 * PW]);          Pk[1] = OF_RP_(p[3 * PW + 1], p[-3 * PW - 1]);
            Pk[2] = OF_RP_(p[2 * PW + 2], p[-2 * PW - 2]);  Pk[3] = OF_RP_(p[PW + 3], p[-PW - 3]);
            Pk[4] = OF_RP_(p[3], p[-3]);                    Pk[5] = OF_RP_(p[-PW + 3], p[PW - 3]);
            Pk[6] = OF_RP_(p[-2 * PW + 2], p[2 * PW - 2]);  Pk[7] = OF_RP_(p[-3 * PW + 1], p[3 * PW - 1]);
#undef OF_RP_
            S[ry * SP + c - cS0] = (uint8_t)orb_fast_score_pk(Pk, tmin);
        }
    };
    uint16_t *Q = reinterpret_cast<uint16_t *>(XT) + wv * 128;
    auto nms = [&](int nl, int tc) {   // maxima with a score >= tc -> s_mask[cell][row]
        auto nms_one = [&](int q) {
            const int c = q & 255, ry = q >> 8;
            const uint8_t *sp = S + ry * SP + c - cS0;
            const int sc = sp[0];
            const int x = ex0 + c, y = ry0 + ry;
            const bool ccol = x >= xm, crow = y >= ym;
            const int xl = ccol ? xm : rx0, xr = ccol ? rx1 : xm, yt = crow ? ym : ry0, yb_ = crow ? ry1 : ym;   // the cell's computed region
            const uint32_t ml = x != xl ? ~0u : 0u, mr = x + 1 != xr ? ~0u : 0u, mu = y != yt ? ~0u : 0u, md = y + 1 != yb_ ? ~0u : 0u;
            const uint32_t a = sp[-1] & ml, b = sp[1] & mr;
            const uint32_t u0 = sp[-SP - 1] & ml, u1 = sp[-SP], u2 = sp[-SP + 1] & mr;
            const uint32_t d0 = sp[SP - 1] & ml, d1 = sp[SP], d2 = sp[SP + 1] & mr;
            const uint32_t up = max(max(u0, u1), u2) & mu, dn = max(max(d0, d1), d2) & md;
            const uint32_t nb = max(max(a, b), max(up, dn));
            if ((uint32_t)sc > nb) atomicOr(&s_mask[(ccol ? 1 : 0) + (crow ? 2 : 0)][y - yt], 1ull << (x - xl));
        };
        int qn = 0;   // (wave-uniform)
        for (int k0 = wv * 64; k0 < nl; k0 += OF_NT) {
            const int k = k0 + lane;
            int q = 0;
            bool corner = false;
            if (k < nl) {
                q = LIST[k];
                corner = (int)S[(q >> 8) * SP + (q & 255) - cS0] >= tc;
            }
            const unsigned long long m = __ballot(corner);
            if (m == 0ull) continue;
            if (corner) Q[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)q;
            qn += __popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (qn >= 64) {
                qn -= 64;
                nms_one(Q[qn + lane]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (lane < qn) nms_one(Q[lane]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // this wave's cell
    const int ecx = cx0 + (wv & 1), ecy = cy0 + (wv >> 1);
    const bool ecell = wv < 4 && ecx < cx1 && ecy < cy1;
    const int cell = L.cell_base + ecy * L.ncx + ecx;
    int4 rc = make_int4(0, 0, 6, 6);
    if (ecell) rc = cells[cell];                     // x0, y0, w, h of the sub-image (level interior coordinates)
    auto emit = [&](unsigned long long mine) {       // raster-ordered emission of the cell's maxima (lane r holds row r's bits)
        const int cnt = __popcll(mine);
        int incl = cnt;   // inclusive prefix sum over the wave: four row shifts, two row broadcasts
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);   // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);   // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);   // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);   // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
        const int total = __builtin_amdgcn_readlane(incl, 63), excl = incl - cnt;
        int base = 0;
        if (lane == 0 && total > 0) base = atomicAdd(&poolcnt[f * g.nlevels + l], total);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane == 0) cellinfo[(size_t)f * g.cells_total + cell] = make_int2(base, total);
        if (total == 0) return;
        if (base + total > (int)L.pool_cap) {  // cannot happen (pool sized for the densest possible NMS output)
            if (lane == 0) atomicOr(status, 1);
            return;
        }
        const uint8_t *sp = S + (rc.y + 3 - ry0) * SP + (rc.x + 3 - ex0 - cS0);
        uint2 *out = pool + (size_t)f * g.pool_stride + L.pool_off + base + excl;
        unsigned long long mm = mine;
        const int gy = rc.y + 3 + lane;
        int k = 0;
        while (mm) {
            const int c = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int x = rc.x + 3 + c;
            const int resp = sp[lane * SP + c];
            // coordinates relative to (minBorderX, minBorderY) as DistributeOctTree expects
            out[k++] = make_uint2((uint32_t)(x - PLF_MINB) | ((uint32_t)(gy - PLF_MINB) << 16), (uint32_t)resp);
        }
    };
    const int ch = rc.w - 6;                         // rows of the cell's computed region
    // one pass = pre-test, score, suppression at threshold t for the cells `on`.  More survivors than LIST holds (dense noise): the rows in two halves -- each fits by
    // construction -- scored one after the other; the suppression needs every score, so it runs on the second half's list and then on the first half's, found again.
    auto fast_pass = [&](int t, uint32_t on, int stop) -> bool {
        pretest(t, on, 0, RH);
        __syncthreads();
        int nl = s_nlist;
        if (stop == 4) return true;
        if (nl + 4 <= list_cap) {   // (a group is refused when fewer than 4 slots are left: nl + 4 <= cap means none was)
            score(nl);
            __syncthreads();
            if (stop == 5) return true;
            nms(nl, t);
            __syncthreads();
            return stop == 6;
        }
        const int hr = (RH + 1) >> 1;
        for (int step = 0; step < 3; step++) {   // rows [0, hr): score; rows [hr, RH): score + suppression; rows [0, hr) again: suppression
            __syncthreads();   // (every thread has read the count)
            if (tid == 0) s_nlist = 0;
            __syncthreads();
            pretest(t, on, step == 1 ? hr : 0, step == 1 ? RH : hr);
            __syncthreads();
            nl = s_nlist;
            if (step < 2) { score(nl); __syncthreads(); }
            if (step > 0) { nms(nl, t); __syncthreads(); }
        }
        return stop == 5 || stop == 6;
    };
#ifdef OF_STOP
    constexpr int of_stop = OF_STOP;
#else
    constexpr int of_stop = 0;
#endif
    // ---- pass A: iniThFAST, all cells
    if (fast_pass(tini, 0xFu, of_stop)) return;
    if (ecell) {
        const unsigned long long my20 = lane < ch ? s_mask[wv][lane] : 0ull;
        if (__ballot(my20 != 0ull) != 0ull) emit(my20);
        else if (lane == 0) atomicOr(&s_need, 1 << wv);
    }
    if (tid == 0) s_nlist = 0;   // (every thread read the count before the last barrier)
    __syncthreads();
    const uint32_t need = (uint32_t)s_need;
    if (need == 0u) return;
    // ---- pass B: minThFAST, the cells the first pass left empty
    fast_pass(tmin, need, 0);
    if (ecell && ((need >> wv) & 1u)) emit(lane < ch ? s_mask[wv][lane] : 0ull);
}
