// orb_kernels.hip -- HIP kernels of the ORB extractor for gfx950 (MI355X).
//
// Pipeline per batch of B frames (every launch covers all frames):
//   k_orb_level x nlevels         ORBextractor::ComputePyramid (include/ORBextractor.h:89, so@0x70430) + the cv::FAST cell calls of
//                                 ComputeKeyPointsOctTree (so@0x75fa0: score, threshold / retry, NMS) + GaussianBlur 7x7 (so@0x77487),
//                                 fused per tile of 2 x 2 cells (orb_front.hip)
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

__constant__ signed char c_pattern[1024];   // (as bytes: ONE 16-byte load per lane.  As floats -- four loads -- the key point wave is slower: the kernel is bound by the CU's vector-memory pipe)
// IC_Angle: the columns u = -16..15 of patch row v that lie inside the disc, |u| <= umax[|v|], as one bit mask per |v| (bit u + 16)
__constant__ uint32_t c_icmask[16];

void plf_orb_upload_constants(const int *umax16)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), plf_bit_pattern_31, 1024);
    uint32_t msk[16];
    for (int v = 0; v < 16; v++) {
        msk[v] = 0;
        for (int u = -16; u < 16; u++)
            if ((u < 0 ? -u : u) <= umax16[v]) msk[v] |= 1u << (u + 16);
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_icmask), msk, sizeof(msk));
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
                                                     int *__restrict__ n_out, int capacity, int *__restrict__ status, OrbGeom g, int nframes)
{
    // one wave per OUTPUT slot o of the frame (level-major order); its level follows from the per-level counts
    // XCD-aware order: workgroups are handed to the 8 XCDs round-robin in dispatch order (x fastest), so with (slot, frame) = (blockIdx.x, blockIdx.y)
    // the key points of ONE frame are spread over all 8 L2s and every L2 pulls most of that frame's planes from HBM (FETCH_SIZE 2.9x the
    // algorithmic bytes).  Remapped: 8 consecutive workgroups take the same slot of 8 different frames, i.e. XCD x works through frame 8 G + x.
    // (round 4: the grid is (8 * slots, ceil(B / 8)) -- blockIdx.y = the group of 8 frames, blockIdx.x = 8 * slot + frame of the group -- so that this order costs a
    // shift and a mask instead of three integer divisions per wave, a fifth of the instructions of a key point)
    const int lane = threadIdx.x;
    const int f = 8 * (int)blockIdx.y + ((int)blockIdx.x & 7), o = (int)blockIdx.x >> 3;
    if (f >= nframes) return;
    const int *cnt = selcnt + f * g.nlevels;
    // (the per-level counts in ONE vector load, lane i = level i, walked with v_readlane: the scalar loop it replaces waited for eight dependent scalar loads)
    const int cl = lane < g.nlevels ? cnt[lane] : 0;
    int offset = 0, total = 0, l = -1;
    for (int i = 0; i < g.nlevels; i++) {
        const int c = __builtin_amdgcn_readlane(cl, i);
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
        // the row segment u = -16..15 as 8 dwords (the padded plane has 19 border pixels) and two byte dot products per
        // dword: sum (u + 16) * I and sum I over |u| <= umax[|v|], so m10 = sum u * I = first - 16 * second (exact integers)
        const uint8_t *row = center + (ptrdiff_t)v * L.ppitch - 16;
        // per dword q: the row mask's nibble spread to a byte mask (n * 0x00204081 puts bit k of n at bit 8 k), the pixels outside the disc zeroed, then the
        // two dot products against constants.  (Round 1-4 built weight words per lane from compile-time u and a run-time umax: ~130 of a key point's ~450
        // vector instructions; a table of the words in memory cost four more 16-byte loads per lane and made the kernel SLOWER, 8.0 -> 9.5 ms.)
        const uint32_t M = c_icmask[v < 0 ? -v : v];
        uint32_t sw = 0, si = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t px4 = *(const plf_u32u *)(row + 4 * q);
            const uint32_t bm = ((((M >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
            const uint32_t pm = px4 & bm;
            sw = __builtin_amdgcn_udot4(pm, 0x03020100u + 0x04040404u * (uint32_t)q, sw, false);
            si = __builtin_amdgcn_udot4(pm, 0x01010101u, si, false);
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
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // lanes i ^ 1 (quad_perm [1, 0, 3, 2])
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // lanes i ^ 2: the quad is complete
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror: a lane of the other quad of the group of 8
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
