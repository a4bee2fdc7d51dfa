// line_kernels.hip -- KeyLine construction, top-N selection, Sobel and the LBD band descriptor on gfx950.
//   k_lsd_finalize  LSDDetector::detectImpl KeyLine fill (opencv_contrib 3.3 line_descriptor/LSDDetector.cpp) +
//                   the fork's "sort by response, keep N, renumber class_id" (include/auxiliar.h:67-72) +
//                   normalised line equations (Eigen cross product in the fork's ExtractLineSegment)
//   k_sobel3        cv::Sobel(img, CV_16S, 1|0, 0|1, 3), BORDER_REFLECT_101 (BinaryDescriptor::computeSobel)
//   k_blur5_sobel3  the same two Sobel calls on BinaryDescriptor::computeGaussianPyramid's octave 0 (GaussianBlur 5x5, sigma 1), fused
//   k_lbd           BinaryDescriptor::computeLBD + binaryConversion (line_descriptor/binary_descriptor.cpp)
// Float arithmetic follows the upstream statement order exactly (no FMA contraction); sequential float sums are
// kept sequential per row / per band and spread over lanes only across rows / bands.
#include "plf_common.h"
#include "lsd_geom.h"

__global__ void __launch_bounds__(256) k_lsd_finalize(const float4 *__restrict__ seg_all, const uint8_t *__restrict__ keep_all,
                                                      const int *__restrict__ nrect, float4 *__restrict__ segs_out, int *__restrict__ nseg_out,
                                                      plf_keyline *__restrict__ kl_tmp_all, plf_keyline *__restrict__ lines,
                                                      double *__restrict__ lineeq, int *__restrict__ n_out, int capacity,
                                                      int *__restrict__ status, unsigned long long *__restrict__ sort_scratch, LsdGeom g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: g.sort_lds keys (the compaction flags alias them: dead before the first key is written, barrier below) + 257 scan words.  A frame with
    // more rectangles / segments than that (textures of thousands of tiny regions) uses its row of the global scratch instead.
    const int f = blockIdx.x, T = blockDim.x, t = threadIdx.x;
    const int nr = nrect[f];
    unsigned long long *grow = sort_scratch ? sort_scratch + (size_t)f * g.sort_cap : (unsigned long long *)smem;
    unsigned long long *skey = (unsigned long long *)smem;
    int *flag = nr <= 2 * g.sort_lds ? (int *)smem : (int *)grow;
    int *scan_tmp = (int *)((unsigned long long *)smem + g.sort_lds);
    const float4 *seg = seg_all + (size_t)f * g.rect_cap;
    const uint8_t *keep = keep_all + (size_t)f * g.rect_cap;
    for (int i = t; i < nr; i += T) flag[i] = keep[i] ? 1 : 0;
    __syncthreads();
    const int ns = plf_block_excl_scan(flag, nr, scan_tmp);
    float4 *so = segs_out + (size_t)f * g.rect_cap;
    plf_keyline *klt = kl_tmp_all + (size_t)f * g.rect_cap;
    const int W = g.w, H = g.h;
    for (int i = t; i < nr; i += T) {
        if (!keep[i]) continue;
        const int k = flag[i];
        const float4 s = seg[i];
        so[k] = s;
        float e0 = s.x, e1 = s.y, e2 = s.z, e3 = s.w;
        if (e0 < 0) e0 = 0;
        if (e0 >= W) e0 = (float)W - 1.0f;
        if (e2 < 0) e2 = 0;
        if (e2 >= W) e2 = (float)W - 1.0f;
        if (e1 < 0) e1 = 0;
        if (e1 >= H) e1 = (float)H - 1.0f;
        if (e3 < 0) e3 = 0;
        if (e3 >= H) e3 = (float)H - 1.0f;
        plf_keyline kl;
        kl.startPointX = e0 * 1.0f; kl.startPointY = e1 * 1.0f; kl.endPointX = e2 * 1.0f; kl.endPointY = e3 * 1.0f;
        kl.sPointInOctaveX = e0; kl.sPointInOctaveY = e1; kl.ePointInOctaveX = e2; kl.ePointInOctaveY = e3;
        kl.lineLength = (float)sqrt((double)(e0 - e2) * (double)(e0 - e2) + (double)(e1 - e3) * (double)(e1 - e3));
        const int x0 = __float2int_rn(e0), y0 = __float2int_rn(e1), x1 = __float2int_rn(e2), y1 = __float2int_rn(e3);
        const int dx = abs(x1 - x0), dy = abs(y1 - y0);
        kl.numOfPixels = (dx > dy ? dx : dy) + 1;
        kl.angle = (float)atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));
        kl.class_id = k;
        kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.response = kl.lineLength / (float)(W > H ? W : H);
        kl.pt_x = (kl.endPointX + kl.startPointX) / 2;
        kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
        klt[k] = kl;
    }
    if (t == 0) nseg_out[f] = ns;
    __syncthreads();
    int nout = ns;
    const bool sorted = ns > g.nkeep;
    if (sorted) {
        // stable "response descending" order: key = (response bits, ~index), sorted descending
        int P2 = 1;
        while (P2 < ns) P2 <<= 1;
        if (P2 > g.sort_lds) skey = grow;
        for (int i = t; i < P2; i += T)
            skey[i] = i < ns ? (((unsigned long long)__float_as_uint(klt[i].response) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
        __syncthreads();
        for (int k2 = 2; k2 <= P2; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = t; i < P2; i += T) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = skey[i], b = skey[ixj];
                        const bool desc = (i & k2) == 0;
                        if (desc ? (a < b) : (a > b)) { skey[i] = b; skey[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        nout = g.nkeep;
    }
    if (nout > capacity) { nout = capacity; if (t == 0) atomicOr(status, 2); }
    for (int i = t; i < nout; i += T) {
        const int src = sorted ? (int)(0xFFFFFFFFu - (unsigned)(skey[i] & 0xFFFFFFFFull)) : i;
        plf_keyline kl = klt[src];
        if (sorted) kl.class_id = i;
        lines[(size_t)f * capacity + i] = kl;
        const double sx = kl.startPointX, sy = kl.startPointY, ex = kl.endPointX, ey = kl.endPointY;
        const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
        const double nrm = sqrt(l0 * l0 + l1 * l1);
        double *eq = lineeq + ((size_t)f * capacity + i) * 3;
        eq[0] = l0 / nrm; eq[1] = l1 / nrm; eq[2] = l2 / nrm;
    }
    if (t == 0) n_out[f] = nout;
}

typedef unsigned long long __attribute__((aligned(1))) plf_u64u;
typedef uint32_t __attribute__((aligned(1))) plf_u32u;            // dword access at byte alignment   // 8-byte access at byte alignment (legal on gfx950 global memory)
struct __attribute__((aligned(4))) plf_short8 { short2 a, b, c, d; };

// A thread produces 4 consecutive pixels: 3 x 8 source bytes (one unaligned 8-byte load per row) instead of 32 byte
// gathers; the image border (REFLECT_101) takes the scalar path.
__global__ void __launch_bounds__(256) k_sobel3(const uint8_t *__restrict__ in, ptrdiff_t pitch, ptrdiff_t fstride, short2 *__restrict__ grad,
                                                LsdGeom g)
{
    const int gpr = (g.w + 3) >> 2, id = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    const int y = id / gpr, x0 = (id - y * gpr) * 4;
    if (y >= g.h) return;
    const uint8_t *img = in + (size_t)f * fstride;
    const uint8_t *r0 = img + (size_t)plf_reflect101(y - 1, g.h) * pitch, *r1 = img + (size_t)y * pitch,
                  *r2 = img + (size_t)plf_reflect101(y + 1, g.h) * pitch;
    short2 *out = grad + (size_t)f * g.full_stride + (size_t)y * g.w + x0;
    if (x0 >= 1 && x0 + 7 <= g.w) {   // bytes x0-1 .. x0+6 exist: pixels x0 .. x0+3 need x0-1 .. x0+4
        const unsigned long long a0 = *(const plf_u64u *)(r0 + x0 - 1), a1 = *(const plf_u64u *)(r1 + x0 - 1), a2 = *(const plf_u64u *)(r2 + x0 - 1);
        short2 o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#define B_(A, k) ((int)(((A) >> (8 * (k))) & 0xFF))
            const int gx = (B_(a0, j + 2) + 2 * B_(a1, j + 2) + B_(a2, j + 2)) - (B_(a0, j) + 2 * B_(a1, j) + B_(a2, j));
            const int gy = (B_(a2, j) + 2 * B_(a2, j + 1) + B_(a2, j + 2)) - (B_(a0, j) + 2 * B_(a0, j + 1) + B_(a0, j + 2));
#undef B_
            o[j] = make_short2((short)gx, (short)gy);
        }
        plf_short8 v; v.a = o[0]; v.b = o[1]; v.c = o[2]; v.d = o[3];
        *(plf_short8 *)out = v;
        return;
    }
    for (int j = 0; j < 4 && x0 + j < g.w; j++) {
        const int x = x0 + j, xm = plf_reflect101(x - 1, g.w), xp = plf_reflect101(x + 1, g.w);
        const int gx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
        const int gy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
        out[j] = make_short2((short)gx, (short)gy);
    }
}

// BinaryDescriptor::computeGaussianPyramid (opencv_contrib 3.3): octave 0 = cv::GaussianBlur(image.clone(), Size(5, 5), 1), then the two
// cv::Sobel calls read THAT image (plf_line_params.lbd_sobel_input = PLF_LBD_BLURRED).  Fused: a 256-thread block produces a 64 x BS_TH tile of
// (dx, dy); the 5-tap row sums (BS_TH + 6 rows x 66), the blurred bytes (BS_TH + 2 rows x 66) and nothing else live in LDS -- the blurred image never reaches HBM.
// 8U GaussianBlur = 8-bit fixed-point separable filter: taps k5 (14 63 103 63 14), row pass exact int32, column pass sum / 65536 rounded as
// OpenCV 3.3's SymmColumnVec_32s8u does (half-to-even) for x < (w & ~3) and as its scalar tail ((s + 32768) >> 16) for the last w % 4
// columns -- the same rule as the 7 x 7 blur of the ORB path (orb_kernels.hip); REFLECT_101 at the image edge for the blur AND for the Sobel.
// Every phase works on groups of 4 pixels: the raw bytes of the tile are staged in LDS with dword loads (396 per tile; round 1 of this kernel issued
// 7260 single-byte global loads per tile and ran at 1 TB/s), the 5-tap row sums are one v_dot4_u32_u8 + one multiply-add per pixel on byte windows
// cut out with v_alignbyte, the column pass reads int4 row-sum vectors, the Sobel reads two dwords per row for 4 pixels.  The staged bytes are already
// mirrored (REFLECT_101) at the image border, so the row pass has no border case.
// (BS_TW, BS_TH: lsd_geom.h)
#define BS_RAWW 80   // staged bytes per row: image x0 - 4 .. x0 + 75
#define BS_RSW 68    // row sums per row:    image x0 - 1 .. x0 + 66 (66 used)
#define BS_BLW 72    // blurred bytes per row: image x0 - 1 .. (68 written, 66 used)
// INNER: the tile's rows y0 - 3 .. y0 + BS_TH + 2 all lie inside the image (every tile but the first and last row of tiles): no row is reflected, no row test
template <bool INNER>
__device__ __forceinline__ void blur5_sobel3_tile(const uint8_t *__restrict__ img, ptrdiff_t pitch, short2 *__restrict__ gout, int W, int H, int x0, int y0, int t, int4 k5,
                                                  uint8_t (*raw)[BS_RAWW], int (*rows)[BS_RSW], uint8_t (*blur)[BS_BLW])
{
    // ---- 0. raw bytes, mirrored in x
    for (int i = t; i < (BS_TH + 6) * (BS_RAWW / 4); i += 256) {
        const int r = i / (BS_RAWW / 4), i4 = i - r * (BS_RAWW / 4);
        const int Y = y0 - 3 + r, X4 = x0 - 4 + 4 * i4;
        if (!INNER && (Y < 0 || Y >= H)) continue;   // only reached through reflection, which lands inside the image
        const uint8_t *S = img + (size_t)Y * pitch;
        uint32_t v;
        if (X4 >= 0 && X4 + 3 < W) v = *(const plf_u32u *)(S + X4);
        else {
            v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) v |= (uint32_t)S[plf_reflect101(X4 + j, W)] << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(&raw[r][4 * i4]) = v;
    }
    __syncthreads();
    // ---- 1. row sums s(X) = k0 (S[X-2] + S[X+2]) + k1 (S[X-1] + S[X+1]) + k2 S[X], 4 per item: X = x0 - 1 + c has its taps at staged columns c + 1 .. c + 5
    {
        const uint32_t K = (uint32_t)k5.x | ((uint32_t)k5.y << 8) | ((uint32_t)k5.z << 16) | ((uint32_t)k5.y << 24);
        for (int i = t; i < (BS_TH + 6) * (BS_RSW / 4); i += 256) {
            const int r = i / (BS_RSW / 4), g4 = i - r * (BS_RSW / 4);
            const int Y = y0 - 3 + r;
            if (!INNER && (Y < 0 || Y >= H)) continue;
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(&raw[r][4 * g4]);
            const uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2];
            int4 o;
            o.x = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), K, (uint32_t)k5.x * ((w1 >> 8) & 0xFFu), false);
            o.y = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), K, (uint32_t)k5.x * ((w1 >> 16) & 0xFFu), false);
            o.z = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), K, (uint32_t)k5.x * (w1 >> 24), false);
            o.w = (int)__builtin_amdgcn_udot4(w1, K, (uint32_t)k5.x * (w2 & 0xFFu), false);
            *reinterpret_cast<int4 *>(&rows[r][4 * g4]) = o;
        }
    }
    __syncthreads();
    // ---- 2. column pass + OpenCV's rounding, 4 per item
    {
        const int wvec = W & ~3;
        for (int i = t; i < (BS_TH + 2) * (BS_RSW / 4); i += 256) {
            const int r = i / (BS_RSW / 4), g4 = i - r * (BS_RSW / 4);
            const int Y = y0 - 1 + r;
            if (!INNER && (Y < 0 || Y >= H)) continue;
#define ROW_(yy) (*reinterpret_cast<const int4 *>(&rows[(INNER ? (yy) : plf_reflect101((yy), H)) - (y0 - 3)][4 * g4]))
            const int4 a = ROW_(Y - 2), b = ROW_(Y - 1), c = ROW_(Y), d = ROW_(Y + 1), e = ROW_(Y + 2);
#undef ROW_
#define COL_(m) ((int)(__umul24(k5.x, a.m + e.m) + __umul24(k5.y, b.m + d.m) + __umul24(k5.z, c.m)))   // (row sums <= 255 * 257: 24-bit multiplies are exact)
            const int sv[4] = {COL_(x), COL_(y), COL_(z), COL_(w)};
#undef COL_
            uint32_t out = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int X = x0 - 1 + 4 * g4 + j, s_ = sv[j];
                // half to even = (s + 0x7FFF + bit 16 of s) >> 16 (as in the 7x7 blur of the ORB path); the scalar tail rounds half up
                const int v = (s_ + 0x7FFF + (((s_ >> 16) & 1) | (X < wvec ? 0 : 1))) >> 16;
                out |= (uint32_t)(v > 255 ? 255 : v) << (8 * j);
            }
            *reinterpret_cast<uint32_t *>(&blur[r][4 * g4]) = out;
        }
    }
    __syncthreads();
    // ---- 3. the two 3 x 3 Sobel derivatives of the blurred image, 4 pixels per item (BS_TH rows x 16 groups)
    for (int it = t; it < BS_TH * 16; it += 256) {
        const int ry = it >> 4, cx = (it & 15) * 4;
        const int x = x0 + cx, y = y0 + ry;
        if (x >= W || y >= H) continue;
        const int ym = INNER ? ry : plf_reflect101(y - 1, H) - (y0 - 1), yc = ry + 1, yp = INNER ? ry + 2 : plf_reflect101(y + 1, H) - (y0 - 1);
        short2 *out = gout + (size_t)y * W + x;
        if (x >= 1 && x + 4 < W) {   // blurred columns x - 1 .. x + 4 exist: blur[][cx .. cx + 5]
            // Packed 16-bit lanes (round 4; the scalar form extracted 36 bytes and did ~100 instructions per 4 pixels, a third of the kernel): the bytes b0..b5 of
            // a row as pairs P01 P23 P45 (and Q12 Q34 for the rows' own sums), column sums C = top + 2 mid + bottom per pair, gx[j] = C[j+2] - C[j];
            // row sums T = P + 2 Q + P', gy = bottom - top.  All values fit 11 bits + sign.
            typedef short s2v __attribute__((ext_vector_type(2)));
#define PERM_(hi, lo, sel) __builtin_bit_cast(s2v, __builtin_amdgcn_perm((hi), (lo), (sel)))
            const uint32_t t0 = *reinterpret_cast<const uint32_t *>(&blur[ym][cx]), t1 = *reinterpret_cast<const uint32_t *>(&blur[ym][cx + 4]);
            const uint32_t m0 = *reinterpret_cast<const uint32_t *>(&blur[yc][cx]), m1 = *reinterpret_cast<const uint32_t *>(&blur[yc][cx + 4]);
            const uint32_t b0 = *reinterpret_cast<const uint32_t *>(&blur[yp][cx]), b1 = *reinterpret_cast<const uint32_t *>(&blur[yp][cx + 4]);
            const s2v two = {2, 2};
            const s2v tP01 = PERM_(0u, t0, 0x0C010C00u), tP23 = PERM_(0u, t0, 0x0C030C02u), tP45 = PERM_(0u, t1, 0x0C010C00u);
            const s2v tQ12 = PERM_(0u, t0, 0x0C020C01u), tQ34 = PERM_(t1, t0, 0x0C040C03u);
            const s2v mP01 = PERM_(0u, m0, 0x0C010C00u), mP23 = PERM_(0u, m0, 0x0C030C02u), mP45 = PERM_(0u, m1, 0x0C010C00u);
            const s2v bP01 = PERM_(0u, b0, 0x0C010C00u), bP23 = PERM_(0u, b0, 0x0C030C02u), bP45 = PERM_(0u, b1, 0x0C010C00u);
            const s2v bQ12 = PERM_(0u, b0, 0x0C020C01u), bQ34 = PERM_(b1, b0, 0x0C040C03u);
            const s2v C01 = mP01 * two + tP01 + bP01, C23 = mP23 * two + tP23 + bP23, C45 = mP45 * two + tP45 + bP45;
            const s2v gx01 = C23 - C01, gx23 = C45 - C23;
            const s2v T01 = tQ12 * two + tP01 + tP23, T23 = tQ34 * two + tP23 + tP45;
            const s2v B01 = bQ12 * two + bP01 + bP23, B23 = bQ34 * two + bP23 + bP45;
            const s2v gy01 = B01 - T01, gy23 = B23 - T23;
            const uint32_t X01 = __builtin_bit_cast(uint32_t, gx01), Y01 = __builtin_bit_cast(uint32_t, gy01);
            const uint32_t X23 = __builtin_bit_cast(uint32_t, gx23), Y23 = __builtin_bit_cast(uint32_t, gy23);
            uint4 o4;   // (gx, gy) of pixels 0..3 as short2 each
            o4.x = __builtin_amdgcn_perm(Y01, X01, 0x05040100u); o4.y = __builtin_amdgcn_perm(Y01, X01, 0x07060302u);
            o4.z = __builtin_amdgcn_perm(Y23, X23, 0x05040100u); o4.w = __builtin_amdgcn_perm(Y23, X23, 0x07060302u);
#undef PERM_
            plf_short8 v;
            v.a = __builtin_bit_cast(short2, o4.x); v.b = __builtin_bit_cast(short2, o4.y); v.c = __builtin_bit_cast(short2, o4.z); v.d = __builtin_bit_cast(short2, o4.w);
            *(plf_short8 *)out = v;
            continue;
        }
        for (int j = 0; j < 4 && x + j < W; j++) {
            const int xx = x + j;
            const int xm = plf_reflect101(xx - 1, W) - (x0 - 1), xc = cx + j + 1, xp = plf_reflect101(xx + 1, W) - (x0 - 1);
            const int gx = (blur[ym][xp] + 2 * blur[yc][xp] + blur[yp][xp]) - (blur[ym][xm] + 2 * blur[yc][xm] + blur[yp][xm]);
            const int gy = (blur[yp][xm] + 2 * blur[yp][xc] + blur[yp][xp]) - (blur[ym][xm] + 2 * blur[ym][xc] + blur[ym][xp]);
            out[j] = make_short2((short)gx, (short)gy);
        }
    }
}

__global__ void __launch_bounds__(256) k_blur5_sobel3(const uint8_t *__restrict__ in, ptrdiff_t pitch, ptrdiff_t fstride, short2 *__restrict__ grad,
                                                      LsdGeom g, int4 k5 /* k[0], k[1], k[2] */)
{
    __shared__ __attribute__((aligned(16))) uint8_t raw[BS_TH + 6][BS_RAWW];   // image (x0 - 4 + i, y0 - 3 + r)
    __shared__ __attribute__((aligned(16))) int rows[BS_TH + 6][BS_RSW];       // row sums at image (x0 - 1 + c, y0 - 3 + r)
    __shared__ __attribute__((aligned(16))) uint8_t blur[BS_TH + 2][BS_BLW];   // blurred bytes at image (x0 - 1 + c, y0 - 1 + r)
    const int x0 = blockIdx.x * BS_TW, y0 = blockIdx.y * BS_TH, f = blockIdx.z, t = threadIdx.x;
    const uint8_t *img = in + (size_t)f * fstride;
    short2 *gout = grad + (size_t)f * g.full_stride;
    if (y0 >= 3 && y0 + BS_TH + 3 <= g.h) blur5_sobel3_tile<true>(img, pitch, gout, g.w, g.h, x0, y0, t, k5, raw, rows, blur);
    else blur5_sobel3_tile<false>(img, pitch, gout, g.w, g.h, x0, y0, t, k5, raw, rows, blur);
}

__constant__ int c_lbd_comb[64] = {0, 1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 1, 2, 1, 3, 1, 4, 1, 5, 1, 6, 2, 3, 2, 4, 2, 5, 2, 6, 2, 7,
                                   2, 8, 3, 4, 3, 5, 3, 6, 3, 7, 3, 8, 4, 5, 4, 6, 4, 7, 4, 8, 5, 6, 5, 7, 5, 8, 6, 7, 6, 8, 7, 8};

// BinaryDescriptor::computeLBD for one line: ONE WAVE per (line, frame).  Lanes 0..62 = the rows of the 63-row line support region (sequential
// float sums along the row, in the reference's order; the gradient gathers of 4 consecutive steps are issued together -- the coordinates are a float
// recurrence that does not depend on the loaded data -- so a row waits for one memory round trip per 4 pixels instead of per pixel); then the 72
// (band, statistic) sums, two passes of the wave; the descriptor statistics (mean / stddev per band), each lane its own entry; the three norm sums are
// ORDERED float sums over 36 / 36 / 72 entries: every lane runs the same chain on values broadcast with v_readlane (no private array: the 72-float
// scratch copy of round 1 cost 140 VGPRs); finally 32 lanes pack the bytes.  cf: the Gaussian weights in device memory (indexed per lane).
#ifdef PLF_LBD_WPE   // experiment switch (tools/variant_build.sh)
#define PLF_LBD_OCC __attribute__((amdgpu_waves_per_eu(PLF_LBD_WPE, PLF_LBD_WPE)))
#else
#define PLF_LBD_OCC
#endif
__device__ __forceinline__ float lbd_bcast(float lo, float hi, int j)   // entry j (0..71) of a 72-vector held as lane j of lo (j < 64) / lane j - 64 of hi
{
    return __int_as_float(j < 64 ? __builtin_amdgcn_readlane(__float_as_int(lo), j) : __builtin_amdgcn_readlane(__float_as_int(hi), j - 64));
}
// (short)(int)roundf(v) clamped to [0, hi] as computeLBD does.  roundf rounds halves away from zero; for v >= 0 that is trunc(v) + (v - trunc(v) >= 0.5) -- the
// difference is exact in float -- and every negative v ends at 0 after the clamp either way (trunc(v) <= 0, no increment).  Coordinates are far inside the
// range of short (line_configure bounds the image size), so the wrap of the cast never acts.
__device__ __forceinline__ short lbd_round_clamp(float v, short hi)
{
    const float tr = truncf(v);
    int r = (short)((int)tr + ((v - tr >= 0.5f) ? 1 : 0));   // (the reference's cast, kept: one v_bfe_i32)
    r = r < 0 ? 0 : r;
    return (short)(r > (int)hi ? (int)hi : r);
}
__global__ void PLF_LBD_OCC __launch_bounds__(64) k_lbd(const short2 *__restrict__ grad_all, const plf_keyline *__restrict__ lines,
                                                        const int *__restrict__ n_out, uint8_t *__restrict__ desc, int capacity, LsdGeom g,
                                                        const LbdCoefs *__restrict__ cf)
{
    __shared__ float rowsum[8][64];
    __shared__ float dv[72];
    const int li = blockIdx.x, f = blockIdx.y, t = threadIdx.x;
    if (li >= n_out[f]) return;
    const plf_keyline kl = lines[(size_t)f * capacity + li];
    const short2 *grad = grad_all + (size_t)f * g.full_stride;
    const int realWidth = g.w;
    const short imageWidth = (short)(g.w - 1), imageHeight = (short)(g.h - 1);
    const short halfHeight = 31;
    const short lengthOfLSP = (short)kl.numOfPixels;
    const short halfWidth = (short)((lengthOfLSP - 1) / 2);
    const float lineMiddlePointX = (float)(0.5 * (double)(kl.sPointInOctaveX + kl.ePointInOctaveX));
    const float lineMiddlePointY = (float)(0.5 * (double)(kl.sPointInOctaveY + kl.ePointInOctaveY));
    const float dL0 = (float)cos((double)kl.angle), dL1 = (float)sin((double)kl.angle);
    const float dO0 = -dL1, dO1 = dL0;
    if (t < 63) {
        float sCorX = -dL0 * (float)halfWidth + dL1 * (float)halfHeight + lineMiddlePointX;
        float sCorY = -dL1 * (float)halfWidth - dL0 * (float)halfHeight + lineMiddlePointY;
        for (int h = 0; h < t; h++) { sCorX -= dL1; sCorY += dL0; }
        float pgdL = 0, ngdL = 0, pgdO = 0, ngdO = 0;
        const int len = lengthOfLSP;
        for (int w0 = 0; w0 < len; w0 += 4) {
            short2 d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const short xCor = lbd_round_clamp(sCorX, imageWidth), yCor = lbd_round_clamp(sCorY, imageHeight);
                d[q] = grad[(int)yCor * realWidth + (int)xCor];   // (steps past the end of the row read a clamped, valid address and are not accumulated)
                sCorX += dL0;
                sCorY += dL1;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (w0 + q < len) {
                    const float gDL = (float)d[q].x * dL0 + (float)d[q].y * dL1;
                    const float gDO = (float)d[q].x * dO0 + (float)d[q].y * dO1;
                    // (if (g > 0) p += g; else n -= g;  as two unconditional additions: the sums start at +0 and only ever receive non-negative terms, so adding
                    // +0 leaves them unchanged bit for bit, and max(-g, 0) is -g exactly when the reference subtracts a negative g)
                    pgdL += fmaxf(gDL, 0.f); ngdL += fmaxf(-gDL, 0.f);
                    pgdO += fmaxf(gDO, 0.f); ngdO += fmaxf(-gDO, 0.f);
                }
            }
        }
        const float cg = cf->gG[t];
        pgdL = cg * pgdL; ngdL = cg * ngdL; pgdO = cg * pgdO; ngdO = cg * ngdO;
        rowsum[0][t] = pgdL; rowsum[1][t] = ngdL; rowsum[2][t] = pgdL * pgdL; rowsum[3][t] = ngdL * ngdL;
        rowsum[4][t] = pgdO; rowsum[5][t] = ngdO; rowsum[6][t] = pgdO * pgdO; rowsum[7][t] = ngdO * ngdO;
    }
    __syncthreads();
    for (int e = t; e < 72; e += 64) {
        // band sums: band b receives, in row order, the rows of bands b-1, b, b+1 with the local Gaussian weights
        const int b = e >> 3, st = e & 7;
        const bool sq = (st & 2) != 0;  // statistics 2,3,6,7 are the squared sums
        float acc = 0;
        const int h0 = max(0, 7 * (b - 1)), h1 = min(62, 7 * (b + 2) - 1);
        for (int hID = h0; hID <= h1; hID++) {
            const int rb = hID / 7;
            const int ci = (rb == b) ? (hID % 7 + 7) : (rb == b + 1 ? hID % 7 + 14 : hID % 7);
            const float c = cf->gL[ci];
            const float v = rowsum[st][hID];
            acc += sq ? (c * c * v) : (c * v);
        }
        dv[e] = acc;  // staged as [band][stat: pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2]
    }
    __syncthreads();
    // descriptor entry j = 8 b + k: k < 4 the band means (of pgdL, ngdL, pgdO, ngdO), k >= 4 their standard deviations.  des_lo = entry t, des_hi = entry 64 + t
    const float invN2 = (float)(1.0 / (7 * 2.0)), invN3 = (float)(1.0 / (7 * 3.0));
    float des_lo = 0.f, des_hi = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int j = t + 64 * pass;
        if (j < 72) {
            const int b = j >> 3, k = j & 7, m = k & 3;
            const float invN = (b == 0 || b == 8) ? invN2 : invN3;
            const int sm = (m < 2) ? m : m + 2;         // slot of the sum: 0, 1, 4, 5
            const float temp = dv[b * 8 + sm] * invN;
            float v = temp;
            if (k >= 4) v = sqrtf(dv[b * 8 + sm + 2] * invN - temp * temp);
            if (pass == 0) des_lo = v; else des_hi = v;
        }
    }
    // tempM / tempS: ordered sums of squares over the 36 means / 36 deviations (band by band, k ascending)
    float tempM = 0, tempS = 0;
#pragma unroll
    for (int b = 0; b < 9; b++) {
#pragma unroll
        for (int k = 0; k < 4; k++) { const float d = lbd_bcast(des_lo, des_hi, 8 * b + k); tempM += d * d; }
#pragma unroll
        for (int k = 4; k < 8; k++) { const float d = lbd_bcast(des_lo, des_hi, 8 * b + k); tempS += d * d; }
    }
    tempM = 1 / sqrtf(tempM);
    tempS = 1 / sqrtf(tempS);
    {
        const bool mean_lo = (t & 7) < 4;   // (entry 64 + t has the same k as entry t)
        des_lo = des_lo * (mean_lo ? tempM : tempS);
        des_hi = des_hi * (mean_lo ? tempM : tempS);
        if ((double)des_lo > 0.4) des_lo = (float)0.4;
        if ((double)des_hi > 0.4) des_hi = (float)0.4;
    }
    float temp = 0;
#pragma unroll
    for (int i = 0; i < 72; i++) { const float d = lbd_bcast(des_lo, des_hi, i); temp += d * d; }
    temp = 1 / sqrtf(temp);
    __syncthreads();   // every lane has read dv
    dv[t] = des_lo * temp;
    if (t < 8) dv[64 + t] = des_hi * temp;
    __syncthreads();
    if (t < 32) {
        const float *f1 = &dv[8 * c_lbd_comb[2 * t]], *f2 = &dv[8 * c_lbd_comb[2 * t + 1]];
        unsigned r = 0;
        for (int i = 0; i < 8; i++)
            if (f1[i] > f2[i]) r += (unsigned)(8 * (8 - i - 1));  // upstream accumulates 8*(7-i), not 1<<(7-i)
        desc[((size_t)f * capacity + li) * 32 + t] = (uint8_t)r;
    }
}
