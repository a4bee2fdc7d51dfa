// match_kernels.hip -- Hamming matchers of the tracking front-end on gfx950.
//   k_build_grid            Frame::AssignFeaturesToGrid (so@0xf9120): 64x48 cell CSR, insertion order kept
//   k_mp_candidates/_rounds ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)   include/ORBmatcher.h:61, so@0x79f10
//                           with Frame::GetFeaturesInArea (include/Frame.h:113, so@0xfbc60)
//   k_match_lastframe       ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  include/ORBmatcher.h:78, so@0x80d00
//   k_knn2                  cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) (cv::batchDistance tie rules)
//   k_lines_lastframe       LSDmatcher::SearchByProjection(Frame&, const Frame&) + Frame::lineDescriptorMAD
//   k_match_project_lines   LSDmatcher::SearchByProjection(Frame&, vector<MapLine*>&, th) + Frame::GetLinesInArea
//   k_match_bow             ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)     include/ORBmatcher.h:104, so@0x80150
//   k_project_kf            search half of ORBmatcher::Fuse (both overloads) and SearchBySim3    include/ORBmatcher.h:116-122
//   k_hamming_matrix        DescriptorDistance over all pairs
//
// The reference loops are GREEDY: map point m skips key points claimed by map points before it, so its result
// depends on theirs.  The kernels keep that order exactly with a conflict-free round scheme: in every round each
// unfinished item posts its index (atomicMin) on all of its still-free candidates; an item that owns every one of
// its candidates has no unfinished predecessor it could interact with, so it is decided now with the current
// claims -- exactly what the sequential loop would see.  Items decided in one round never share a candidate, the
// lowest unfinished item always qualifies, and the result equals the sequential loop for any schedule.
// No MFMA: 256-bit XOR + popcount per candidate; the work is gather / compare bound.
#include "plf_common.h"

#define GRID_COLS 64
#define GRID_ROWS 48
#define GRID_CELLS (GRID_COLS * GRID_ROWS)
#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30

struct FrameDev {
    int n;
    const int *n_dev;
    const plf_keypoint *keys;
    const float *uright;
    const uint8_t *desc;
    float min_x, min_y, max_x, max_y, inv_w, inv_h;
    const float *scale_factors;
    int nlevels;
    const int *cell_start;  // GRID_CELLS + 1
    const int *cell_idx;    // n
    const float4 *cell_kp;  // n: (x, y, octave, index) of the key points in cell order (cell_idx order)
};

__global__ void __launch_bounds__(256) k_build_grid(const FrameDev *__restrict__ frames, int *__restrict__ cell_start_all,
                                                    int *__restrict__ cell_idx_all, int *__restrict__ cell_of_all, int kp_stride)
{
    __shared__ int cnt[GRID_CELLS + 1];
    __shared__ int scan_tmp[260];
    const int f = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    FrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int *cell_of = cell_of_all + (size_t)f * kp_stride;
    int *cs = cell_start_all + (size_t)f * (GRID_CELLS + 1);
    int *ci = cell_idx_all + (size_t)f * kp_stride;
    for (int c = t; c <= GRID_CELLS; c += T) cnt[c] = 0;
    __syncthreads();
    for (int i = t; i < F.n; i += T) {
        const plf_keypoint kp = F.keys[i];
        const int px = (int)roundf((kp.x - F.min_x) * F.inv_w), py = (int)roundf((kp.y - F.min_y) * F.inv_h);
        int c = -1;
        if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) { c = px * GRID_ROWS + py; atomicAdd(&cnt[c], 1); }
        cell_of[i] = c;
    }
    __syncthreads();
    plf_block_excl_scan(cnt, GRID_CELLS + 1, scan_tmp);
    for (int c = t; c <= GRID_CELLS; c += T) cs[c] = cnt[c];
    __syncthreads();
    // stable placement (insertion order inside a cell = key-point order).  Round 5: ONE wave walks the key points in chunks of 64 -- the rank of a key point among
    // the earlier ones of its cell = the cell's cursor (cnt[], advanced chunk by chunk) + the lanes below it in the chunk that hold the same cell (64 v_readlane
    // compares).  Until then every key point counted ALL earlier key points of the frame with the same cell, a loop of dependent global loads: O(n^2) -- 0.12 ms
    // per call for the 1000 key points of one VGA frame, 0.53 ms for 2000, twice per frame (map-point and last-frame search): as long as the ORB extraction itself.
    if (t < 64) {
        for (int base = 0; base < F.n; base += 64) {
            const int i = base + t;
            const int c = i < F.n ? cell_of[i] : -1;
            int r = 0, total = 0;
#pragma unroll 8
            for (int l = 0; l < 64; l++) {
                const int cl = __builtin_amdgcn_readlane(c, l);
                const bool same = cl == c;
                r += (same && l < t) ? 1 : 0; total += same ? 1 : 0;
            }
            int pos = 0;
            if (c >= 0) {
                pos = cnt[c] + r;
                ci[pos] = i;
                const plf_keypoint kp = F.keys[i];
                const_cast<float4 *>(F.cell_kp)[pos] = make_float4(kp.x, kp.y, __int_as_float(kp.octave), __int_as_float(i));
            }
            __builtin_amdgcn_wave_barrier();
            if (c >= 0 && r == total - 1) cnt[c] = pos + 1;   // the last lane of every cell group moves the cell's cursor (one writer per cell)
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ __forceinline__ float radius_by_viewing_cos(float vc) { return ((double)vc > 0.998) ? 2.5f : 4.0f; }

__device__ __forceinline__ int hamming_g(const uint8_t *a, const uint8_t *b)
{
    const uint4 *pa = reinterpret_cast<const uint4 *>(a), *pb = reinterpret_cast<const uint4 *>(b);
    const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Frame::GetFeaturesInArea cell window
struct CellWin { int x0, x1, y0, y1; bool ok; };
__device__ __forceinline__ CellWin cell_window(const FrameDev &F, float x, float y, float r)
{
    CellWin w;
    w.ok = false;
    w.x0 = max(0, (int)floorf((x - F.min_x - r) * F.inv_w));
    if (w.x0 >= GRID_COLS) return w;
    w.x1 = min(GRID_COLS - 1, (int)ceilf((x - F.min_x + r) * F.inv_w));
    if (w.x1 < 0) return w;
    w.y0 = max(0, (int)floorf((y - F.min_y - r) * F.inv_h));
    if (w.y0 >= GRID_ROWS) return w;
    w.y1 = min(GRID_ROWS - 1, (int)ceilf((y - F.min_y + r) * F.inv_h));
    if (w.y1 < 0) return w;
    w.ok = true;
    return w;
}

// iterate the candidates of GetFeaturesInArea(x, y, r, minLevel, maxLevel) in the reference order
#define FOR_EACH_CANDIDATE(F, W, X, Y, R, MINL, MAXL, IDX, BODY)                                              \
    {                                                                                                         \
        const bool _chk = ((MINL) > 0) || ((MAXL) >= 0);                                                      \
        for (int _ix = (W).x0; _ix <= (W).x1; _ix++)                                                          \
            for (int _iy = (W).y0; _iy <= (W).y1; _iy++) {                                                    \
                const int _c = _ix * GRID_ROWS + _iy;                                                         \
                for (int _j = (F).cell_start[_c]; _j < (F).cell_start[_c + 1]; _j++) {                        \
                    const int IDX = (F).cell_idx[_j];                                                         \
                    const plf_keypoint _kp = (F).keys[IDX];                                                   \
                    if (_chk) {                                                                               \
                        if (_kp.octave < (MINL)) continue;                                                    \
                        if ((MAXL) >= 0 && _kp.octave > (MAXL)) continue;                                     \
                    }                                                                                         \
                    if (!(fabsf(_kp.x - (X)) < (R) && fabsf(_kp.y - (Y)) < (R))) continue;                    \
                    BODY                                                                                      \
                }                                                                                             \
            }                                                                                                 \
    }

struct MapDev {
    int m;
    const float *proj_x, *proj_y, *proj_xr;
    const int *level;
    const float *view_cos;
    const uint8_t *in_view;
    const uint8_t *desc;
    const uint8_t *obs_positive;
};

__device__ __forceinline__ bool blocked(const int *claim, int k, const uint8_t *obs_positive)
{
    const int c = claim[k];
    return c == -2 || (c >= 0 && (!obs_positive || obs_positive[c]));
}

// Fallback used only for frames whose candidate lists do not fit the cache of k_mp_candidates (gate:
// overflow[f] != 0): same round scheme with the conservative rule "an item must own ALL its free candidates" and
// the candidates re-enumerated from the grid in every round.
// one block per frame; claim[] and owner[] live in LDS (kp_cap ints each)
__global__ void __launch_bounds__(256) k_match_project_points_slow(const FrameDev *__restrict__ frames, MapDev MP, float th, float nnratio,
                                                                   int *__restrict__ match_all, int kp_stride, int *__restrict__ nmatches,
                                                                   uint8_t *__restrict__ done_all, int kp_cap, const int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + kp_cap;
    __shared__ int s_left, s_acc;
    const int f = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    if (!overflow[f]) return;
    FrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int *match = match_all + (size_t)f * kp_stride;
    uint8_t *done = done_all + (size_t)f * MP.m;
    const bool bFactor = th != 1.0f;
    for (int k = t; k < F.n; k += T) claim[k] = match[k] < -2 ? -1 : match[k];   // (-3 marks of an earlier check_orientation = 2 call read as free; -2 stays "occupied")
    if (t == 0) s_acc = 0;
    for (int m = t; m < MP.m; m += T) done[m] = MP.in_view[m] ? 0 : 1;
    __syncthreads();
    for (int round = 0; round <= MP.m; round++) {
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        if (t == 0) s_left = 0;
        __syncthreads();
        // post: every unfinished map point marks its free candidates with its index
        for (int m = t; m < MP.m; m += T) {
            if (done[m]) continue;
            const int lvl = MP.level[m];
            float r = radius_by_viewing_cos(MP.view_cos[m]);
            if (bFactor) r *= th;
            const float rad = r * F.scale_factors[lvl];
            const float x = MP.proj_x[m], y = MP.proj_y[m];
            const CellWin w = cell_window(F, x, y, rad);
            if (w.ok) FOR_EACH_CANDIDATE(F, w, x, y, rad, lvl - 1, lvl, idx, { if (!blocked(claim, idx, MP.obs_positive)) atomicMin(&owner[idx], m); })
        }
        __syncthreads();
        // decide: map points owning all their free candidates are independent of every unfinished predecessor
        for (int m = t; m < MP.m; m += T) {
            if (done[m]) continue;
            const int lvl = MP.level[m];
            float r = radius_by_viewing_cos(MP.view_cos[m]);
            if (bFactor) r *= th;
            const float rad = r * F.scale_factors[lvl];
            const float x = MP.proj_x[m], y = MP.proj_y[m];
            const CellWin w = cell_window(F, x, y, rad);
            bool safe = true;
            if (w.ok) FOR_EACH_CANDIDATE(F, w, x, y, rad, lvl - 1, lvl, idx, { if (!blocked(claim, idx, MP.obs_positive) && owner[idx] != m) safe = false; })
            if (!safe) { atomicAdd(&s_left, 1); continue; }
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            const uint8_t *d = MP.desc + (size_t)m * 32;
            if (w.ok) FOR_EACH_CANDIDATE(F, w, x, y, rad, lvl - 1, lvl, idx, {
                if (blocked(claim, idx, MP.obs_positive)) continue;
                if (F.uright) {
                    const float ur = F.uright[idx];
                    if (ur > 0) { const float er = fabsf(MP.proj_xr[m] - ur); if (er > rad) continue; }
                }
                const int dist = hamming_g(d, F.desc + (size_t)idx * 32);
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = _kp.octave; bestIdx = idx; }
                else if (dist < bestDist2) { bestLevel2 = _kp.octave; bestDist2 = dist; }
            })
            done[m] = 1;
            if (bestDist <= TH_HIGH) {
                if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
                claim[bestIdx] = m;  // only this map point can touch bestIdx in this round
                atomicAdd(&s_acc, 1);
            }
        }
        __syncthreads();
        if (s_left == 0) break;
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) nmatches[f] = s_acc;
}

// ------------------------------------------------------------------------------------------------
// SearchByProjection against the local map, fast path: two kernels.
// k_mp_candidates (one thread per (frame, map point), whole GPU busy) enumerates every map point's candidates ONCE
// (GetFeaturesInArea order, level / window / uRight tests, statically occupied key points dropped) and caches
// (index, Hamming distance, octave) per candidate in a span allocated from the frame's candidate pool.
// k_mp_rounds (one block per frame) then resolves the greedy order in rounds on the cached lists.  A map point's
// outcome is a function of its two best still-free candidates only (best = first occurrence of the minimum distance,
// second = first occurrence of the next value -- exactly what the reference's running best/second-best scan yields),
// so it can be decided as soon as no UNFINISHED EARLIER map point lists either of those two key points among its own
// free candidates:
//   post:   every unfinished map point writes its index (atomicMin) on all its free candidates;
//   decide: a map point whose top-2 candidates both carry its own index is final.
// Two map points decided in the same round can never take each other's top-2, the lowest unfinished one always
// qualifies, so the result is the sequential loop's for any schedule (and for any placement of the spans).
// cand entry: idx (16 bits) | dist (9 bits) << 16 | octave (4 bits) << 25.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mp_candidates(const FrameDev *__restrict__ frames, MapDev MP, float th,
                                                       const int *__restrict__ match_all, int kp_stride, uint8_t *__restrict__ done_all,
                                                       uint32_t *__restrict__ cand_all, int2 *__restrict__ span_all, int cand_cap,
                                                       int *__restrict__ overflow, int *__restrict__ total)
{
    const int f = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x;
    FrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    const int *claim = match_all + (size_t)f * kp_stride;   // occupancy before this call: never becomes free
    uint32_t *cand = cand_all + (size_t)f * cand_cap;
    const bool in = m < MP.m;
    const bool act = in && MP.in_view[m] != 0;
    int ub = 0, lvl = 0;
    float rad = 0.f, x = 0.f, y = 0.f, xr = 0.f;
    CellWin w;
    w.ok = false;
    if (act) {
        lvl = MP.level[m];
        float r = radius_by_viewing_cos(MP.view_cos[m]);
        if (th != 1.0f) r *= th;
        rad = r * F.scale_factors[lvl];
        x = MP.proj_x[m]; y = MP.proj_y[m];
        if (F.uright) xr = MP.proj_xr[m];
        w = cell_window(F, x, y, rad);
        // cells (ix, y0..y1) are consecutive in the CSR, so a window column is ONE index range; the pool span is
        // sized by the number of key points in the window (an upper bound of the candidates)
        if (w.ok)
            for (int ix = w.x0; ix <= w.x1; ix++) ub += F.cell_start[ix * GRID_ROWS + w.y1 + 1] - F.cell_start[ix * GRID_ROWS + w.y0];
    }
    const int excl = plf_wave_excl_scan(ub), wsum = plf_wave_sum(ub);
    int base = 0;
    if (wsum > 0) {
        if (plf_lane() == 0) base = atomicAdd(&total[f], wsum);
        base = __shfl(base, 0, 64);
    }
    const bool fits = base + wsum <= cand_cap;
    if (!fits && plf_lane() == 0) overflow[f] = 1;   // this frame is left to k_match_project_points_slow
    if (!in) return;
    const int o0 = base + excl;
    int o = o0;
    if (ub > 0 && fits) {
        const uint4 *dp = reinterpret_cast<const uint4 *>(MP.desc + (size_t)m * 32);
        const uint4 d0 = dp[0], d1 = dp[1];
        for (int ix = w.x0; ix <= w.x1; ix++) {
            const int j1 = F.cell_start[ix * GRID_ROWS + w.y1 + 1];
            for (int j = F.cell_start[ix * GRID_ROWS + w.y0]; j < j1; j++) {
                const float4 e = F.cell_kp[j];
                const int oct = __float_as_int(e.z), idx = __float_as_int(e.w);
                if (oct < lvl - 1 || oct > lvl) continue;
                if (!(fabsf(e.x - x) < rad && fabsf(e.y - y) < rad)) continue;
                if (blocked(claim, idx, MP.obs_positive)) continue;
                if (F.uright) { const float ur = F.uright[idx]; if (ur > 0 && fabsf(xr - ur) > rad) continue; }
                const uint4 *kp = reinterpret_cast<const uint4 *>(F.desc + (size_t)idx * 32);
                const uint4 k0 = kp[0], k1 = kp[1];
                const int dist = __popc(d0.x ^ k0.x) + __popc(d0.y ^ k0.y) + __popc(d0.z ^ k0.z) + __popc(d0.w ^ k0.w) + __popc(d1.x ^ k1.x) +
                                 __popc(d1.y ^ k1.y) + __popc(d1.z ^ k1.z) + __popc(d1.w ^ k1.w);
                cand[o++] = (uint32_t)idx | ((uint32_t)dist << 16) | ((uint32_t)(oct & 15) << 25);
            }
        }
    }
    span_all[(size_t)f * MP.m + m] = make_int2(o0, o - o0);
    done_all[(size_t)f * MP.m + m] = (o > o0) ? 0 : 1;
}

// LDS: claim[kp_cap], owner[kp_cap] (int), ONE list of unfinished map points (uint16, MP.m <= 65535), compacted in place after every round -- a second
// list made the block 28 KB at 5000 map points (5 blocks per CU, and no room next to an ORB tile in the shadow of the region-growing kernel)
// Round 4: ONE scan of an item's cached candidates per round.  The scan that posts the item's index on its free candidates also finds its two best free candidates
// (both depend only on the claims at the start of the round) and parks them, packed in 64 bits, in a per-frame scratch row; the decision pass reads that word back
// instead of scanning the candidates again -- the kernel moved 3.7 MB per frame through the L2 for 1.3 MB of candidate lists (FETCH_SIZE 31 GB per 8192-frame launch).
__global__ void __launch_bounds__(256) k_mp_rounds(const FrameDev *__restrict__ frames, MapDev MP, float nnratio, int *__restrict__ match_all,
                                                   int kp_stride, int *__restrict__ nmatches, const uint8_t *__restrict__ done_all, int kp_cap,
                                                   const uint32_t *__restrict__ cand_all, const int2 *__restrict__ span_all, int cand_cap,
                                                   const int *__restrict__ overflow, unsigned long long *__restrict__ top2_all, int top2_stride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + kp_cap;
    uint16_t *la = (uint16_t *)(owner + kp_cap);
    __shared__ int s_n[2], s_acc;
    const int f = blockIdx.x, t = threadIdx.x, T = blockDim.x, lane = plf_lane();
    if (overflow[f]) return;
    FrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int *match = match_all + (size_t)f * kp_stride;
    const uint8_t *done = done_all + (size_t)f * MP.m;
    const uint32_t *cand = cand_all + (size_t)f * cand_cap;
    const int2 *span = span_all + (size_t)f * MP.m;
    for (int k = t; k < F.n; k += T) claim[k] = match[k] < -2 ? -1 : match[k];   // (-3 marks of an earlier check_orientation = 2 call read as free; -2 stays "occupied")
    if (t == 0) { s_acc = 0; s_n[0] = 0; s_n[1] = 0; }
    __syncthreads();
    for (int m0 = 0; m0 < MP.m; m0 += T) {
        const int m = m0 + t;
        const bool un = m < MP.m && !done[m];
        const unsigned long long mask = __ballot(un);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&s_n[0], __popcll(mask));
        base = __shfl(base, 0, 64);
        if (un) la[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)m;
    }
    __syncthreads();
    int cur = 0;
    for (int round = 0; round <= MP.m; round++) {
        const int nact = s_n[cur];
        if (nact == 0) break;
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        __syncthreads();
        unsigned long long *top2 = top2_all + (size_t)f * top2_stride;
        for (int i = t; i < nact; i += T) {
            const int m = la[i];
            const int2 sp = span[m];
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1, idx2 = -1;
            for (int j = sp.x; j < sp.x + sp.y; j++) {
                const uint32_t e = cand[j];
                const int idx = (int)(e & 0xFFFF);
                if (blocked(claim, idx, MP.obs_positive)) continue;
                atomicMin(&owner[idx], m);
                const int dist = (int)((e >> 16) & 0x1FF), oct = (int)(e >> 25);
                if (dist < bestDist) { bestDist2 = bestDist; bestLevel2 = bestLevel; idx2 = bestIdx; bestDist = dist; bestLevel = oct; bestIdx = idx; }
                else if (dist < bestDist2) { bestLevel2 = oct; bestDist2 = dist; idx2 = idx; }
            }
            // bestIdx | idx2 << 16 | bestDist << 32 | bestDist2 << 41 | bestLevel << 50 | bestLevel2 << 57   (-1 -> all ones of the field; distances <= 256: 9 bits)
            top2[i] = (unsigned long long)(bestIdx & 0xFFFF) | ((unsigned long long)(idx2 & 0xFFFF) << 16) | ((unsigned long long)(bestDist & 0x1FF) << 32) |
                      ((unsigned long long)(bestDist2 & 0x1FF) << 41) | ((unsigned long long)(bestLevel & 0x7F) << 50) | ((unsigned long long)(bestLevel2 & 0x7F) << 57);
        }
        __syncthreads();
        for (int i0 = 0; i0 < nact; i0 += T) {
            const int i = i0 + t;
            bool keep = false;
            int m = 0;
            if (i < nact) {
                m = la[i];
                const unsigned long long tw = top2[i];
                int bestIdx = (int)(tw & 0xFFFF), idx2 = (int)((tw >> 16) & 0xFFFF);
                if (bestIdx == 0xFFFF) bestIdx = -1;
                if (idx2 == 0xFFFF) idx2 = -1;
                const int bestDist = (int)((tw >> 32) & 0x1FF), bestDist2 = (int)((tw >> 41) & 0x1FF);
                const int bestLevel = (int)((tw >> 50) & 0x7F), bestLevel2 = (int)((tw >> 57) & 0x7F);   // (0x7F = none: only ever compared for equality, with a real level on one side)
                const bool safe = (bestIdx < 0 || owner[bestIdx] == m) && (idx2 < 0 || owner[idx2] == m);
                keep = !safe;
                if (safe && bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)) {
                    claim[bestIdx] = m;   // only this map point can touch bestIdx in this round
                    atomicAdd(&s_acc, 1);
                }
            }
            // in place: the kept items of chunks 0 .. i0 / T land below i0 + T; every thread of the block has read its entry of THIS chunk before any is written
            __syncthreads();
            const unsigned long long mask = __ballot(keep);
            int base = 0;
            if (lane == 0 && mask) base = atomicAdd(&s_n[cur ^ 1], __popcll(mask));
            base = __shfl(base, 0, 64);
            if (keep) la[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)m;
        }
        __syncthreads();
        if (t == 0) s_n[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) nmatches[f] = s_acc;
}

struct LastDev {
    int n;
    const uint8_t *has_mp, *outlier;
    const float *xw;
    const plf_keypoint *keys;
    const uint8_t *mp_desc;
    const uint8_t *obs_positive;   // Observations() > 0 per last-frame map point; NULL = all (motion-model overload only)
};

// one block; items = key points of the last frame.  proj[i] = (u, v, invzc, radius) prepared in the first phase.
// Relocalisation overload, ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist)
// (include/ORBmatcher.h:82, so@0x7e8c0), shares the kernel: items = keyframe features with a usable map point, no depth-sign
// and no uRight test, level window from MapPoint::PredictScale (so@0x8fc20), acceptance threshold ORBdist.
struct RelocDev { int on; const float *min_dist, *max_dist; float log_scale; int orb_dist; };

__global__ void __launch_bounds__(256) k_match_lastframe(const FrameDev *__restrict__ frames, LastDev Lf, const plf_pose_pair *__restrict__ poses, RelocDev RL,
                                                         float th, int mono, int check_ori, int *__restrict__ match_all, int kp_stride,
                                                         int *__restrict__ nmatches_all, uint8_t *__restrict__ done_all, float4 *__restrict__ proj_all, int kp_cap,
                                                         int item_stride, const int *__restrict__ overflow)
{
    // one block per current frame (blockIdx.x); every frame is matched against the same last frame / keyframe with its own pose
    if (overflow && !overflow[blockIdx.x]) return;   // batched calls: this kernel is the fallback for frames k_lf_candidates could not cache
    FrameDev F = frames[blockIdx.x];
    const plf_pose_pair P = poses[blockIdx.x];
    int *match = match_all + (size_t)blockIdx.x * kp_stride, *nmatches = nmatches_all + blockIdx.x;
    uint8_t *done = done_all + (size_t)blockIdx.x * item_stride;
    float4 *proj = proj_all + (size_t)blockIdx.x * item_stride;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + kp_cap;
    __shared__ int s_left, s_acc, hist[HISTO_LENGTH], keepbin[3];
    const int t = threadIdx.x, T = blockDim.x;
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    // twc = -Rcw^T tcw (generic double-accumulating gemm), tlc = Rlw twc + tlw (float small-matrix path)
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++)
        twc[i] = (float)(-((double)P.Rcw[i] * P.tcw[0] + (double)P.Rcw[3 + i] * P.tcw[1] + (double)P.Rcw[6 + i] * P.tcw[2]));
    for (int i = 0; i < 3; i++) tlc[i] = P.Rlw[i * 3] * twc[0] + P.Rlw[i * 3 + 1] * twc[1] + P.Rlw[i * 3 + 2] * twc[2] + P.tlw[i];
    const bool bForward = tlc[2] > P.b && !mono, bBackward = -tlc[2] > P.b && !mono;
    // "CurrentFrame.mvpMapPoints[i2] && ->Observations() > 0" (so@0x81e3d): a key point taken by a last-frame point WITHOUT observations (a temporal
    // point of localisation mode) stays available to later points and is overwritten; the relocalisation overload tests the pointer only
    const uint8_t *obs = RL.on ? nullptr : Lf.obs_positive;
#define KP_FREE(idx) (claim[idx] == -1 || (obs && claim[idx] >= 0 && !obs[claim[idx]]))
    for (int k = t; k < F.n; k += T) claim[k] = match[k] < -2 ? -1 : match[k];   // (-3 marks of an earlier check_orientation = 2 call read as free; -2 stays "occupied")
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) s_acc = 0;
    for (int i = t; i < Lf.n; i += T) {
        bool act = Lf.has_mp[i] && (RL.on || !Lf.outlier[i]);
        float4 pr = make_float4(0, 0, 0, 0);
        int lvl = 0;
        if (act) {
            const float *xw = Lf.xw + 3 * (size_t)i;
            const float xc = P.Rcw[0] * xw[0] + P.Rcw[1] * xw[1] + P.Rcw[2] * xw[2] + P.tcw[0];
            const float yc = P.Rcw[3] * xw[0] + P.Rcw[4] * xw[1] + P.Rcw[5] * xw[2] + P.tcw[1];
            const float zc = P.Rcw[6] * xw[0] + P.Rcw[7] * xw[1] + P.Rcw[8] * xw[2] + P.tcw[2];
            const float invzc = (float)(1.0 / (double)zc);
            if (!RL.on && invzc < 0) act = false;
            // the reference binary contracts these into FMAs (so@0x81cba, so@0x81cd9; the uRight test below: so@0x81eb5)
            const float u = fmaf(P.fx * xc, invzc, P.cx), v = fmaf(P.fy * yc, invzc, P.cy);
            if (u < F.min_x || u > F.max_x) act = false;
            if (v < F.min_y || v > F.max_y) act = false;
            if (act && RL.on) {
                // dist3D = float(cv::norm(x3Dw - Ow)) (double accumulation), Ow = twc; invariance range; predicted level
                const float PO[3] = {xw[0] - twc[0], xw[1] - twc[1], xw[2] - twc[2]};
                double s2 = 0;
                for (int k = 0; k < 3; k++) s2 += (double)PO[k] * (double)PO[k];
                const float dist3D = (float)sqrt(s2);
                const float maxDistance = 1.2f * RL.max_dist[i], minDistance = 0.8f * RL.min_dist[i];
                if (dist3D < minDistance || dist3D > maxDistance) act = false;
                const float ratio = RL.max_dist[i] / dist3D;
                // logf of the reference's libm is (almost always) the correctly rounded value; the double log rounded to float is too
                lvl = (int)ceilf((float)log((double)ratio) / RL.log_scale);
                if (lvl < 0) lvl = 0;
                else if (lvl >= F.nlevels) lvl = F.nlevels - 1;
            }
            if (act) pr = make_float4(u, v, invzc, th * F.scale_factors[RL.on ? lvl : Lf.keys[i].octave]);
        }
        if (!act) pr.x = __int_as_float(-1);   // finished items keep their assignment (key point index, -1 none) in .x
        proj[i] = pr;
        done[i] = act ? (uint8_t)(lvl << 1) : 1;   // bit 0: finished, bits 1..: predicted level (relocalisation)
    }
    __syncthreads();
    for (int round = 0; round <= Lf.n; round++) {
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        if (t == 0) s_left = 0;
        __syncthreads();
        for (int i = t; i < Lf.n; i += T) {
            if (done[i] & 1) continue;
            const float4 pr = proj[i];
            const int oct = RL.on ? (done[i] >> 1) : Lf.keys[i].octave;
            const int minL = RL.on ? oct - 1 : (bForward ? oct : (bBackward ? 0 : oct - 1)), maxL = RL.on ? oct + 1 : (bForward ? -1 : (bBackward ? oct : oct + 1));
            const CellWin w = cell_window(F, pr.x, pr.y, pr.w);
            if (w.ok) FOR_EACH_CANDIDATE(F, w, pr.x, pr.y, pr.w, minL, maxL, idx, { if (KP_FREE(idx)) atomicMin(&owner[idx], i); })
        }
        __syncthreads();
        for (int i = t; i < Lf.n; i += T) {
            if (done[i] & 1) continue;
            const float4 pr = proj[i];
            const int oct = RL.on ? (done[i] >> 1) : Lf.keys[i].octave;
            const int minL = RL.on ? oct - 1 : (bForward ? oct : (bBackward ? 0 : oct - 1)), maxL = RL.on ? oct + 1 : (bForward ? -1 : (bBackward ? oct : oct + 1));
            const CellWin w = cell_window(F, pr.x, pr.y, pr.w);
            bool safe = true;
            if (w.ok) FOR_EACH_CANDIDATE(F, w, pr.x, pr.y, pr.w, minL, maxL, idx, { if (KP_FREE(idx) && owner[idx] != i) safe = false; })
            if (!safe) { atomicAdd(&s_left, 1); continue; }
            int bestDist = 256, bestIdx2 = -1;
            const uint8_t *d = Lf.mp_desc + 32 * (size_t)i;
            if (w.ok) FOR_EACH_CANDIDATE(F, w, pr.x, pr.y, pr.w, minL, maxL, idx, {
                if (!KP_FREE(idx)) continue;
                if (F.uright && !RL.on) {
                    const float urr = F.uright[idx];
                    if (urr > 0) { const float ur = fmaf(-P.bf, pr.z, pr.x); const float er = fabsf(ur - urr); if (er > pr.w) continue; }
                }
                const int dist = hamming_g(d, F.desc + 32 * (size_t)idx);
                if (dist < bestDist) { bestDist = dist; bestIdx2 = idx; }
            })
            done[i] |= 1;
            proj[i].x = __int_as_float(-1);   // the item's projection is no longer needed: the slot records its assignment
            if (bestDist <= (RL.on ? RL.orb_dist : TH_HIGH)) {
                claim[bestIdx2] = i;
                proj[i].x = __int_as_float(bestIdx2);
                atomicAdd(&s_acc, 1);
                if (check_ori) {
                    float rot = Lf.keys[i].angle - F.keys[bestIdx2].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * (1.0f / 12.0f));  // this binary: HISTO_LENGTH / 360 (so@0x829b5)
                    if (bin == HISTO_LENGTH) bin = 0;
                    atomicAdd(&hist[bin], 1);
                }
            }
        }
        __syncthreads();
        if (s_left == 0) break;
        __syncthreads();
    }
    if (check_ori) {
        if (t == 0) {  // ComputeThreeMaxima (so@0x823eb)
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            keepbin[0] = i1; keepbin[1] = i2; keepbin[2] = i3;
        }
        __syncthreads();
        // the reference culls per ASSIGNMENT (rotHist entries): an overwritten key point whose earlier or later assignment falls in a dropped bin
        // ends up NULL, and nmatches is decremented once per dropped assignment
        for (int i = t; i < Lf.n; i += T) {
            const int k = __float_as_int(proj[i].x);
            if (k < 0 || k >= F.n) continue;
            float rot = Lf.keys[i].angle - F.keys[k].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / 12.0f));
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin != keepbin[0] && bin != keepbin[1] && bin != keepbin[2]) { claim[k] = check_ori == 2 ? -3 : -1; atomicSub(&s_acc, 1); }
        }
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) *nmatches = s_acc;
#undef KP_FREE
}

// ------------------------------------------------------------------------------------------------
// Fast path of the motion-model search for batches (plf_match_project_lastframe_batch): the scheme of k_mp_candidates / k_mp_rounds.
//   k_lf_candidates   one thread per (frame, last-frame point): projection with the frame's pose, cell window, level window, uRight test and the
//                     Hamming distance of every admissible key point, cached as (index | distance << 16) in the reference's candidate order
//   k_lf_rounds       one block per frame: the greedy loop in conflict-free rounds on the cached lists.  This overload keeps the BEST distance only
//                     (no ratio test), so a point's outcome is decided by its best still-available candidate alone: it is final as soon as no
//                     unfinished EARLIER point lists that key point (owner[best] == i).  Then the rotation histogram and its per-assignment cull.
// Frames whose lists overflow the pool are left to k_match_lastframe (gate: overflow[f]).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lf_candidates(const FrameDev *__restrict__ frames, LastDev Lf, const plf_pose_pair *__restrict__ poses, float th, int mono,
                                                       const int *__restrict__ match_all, int kp_stride, uint8_t *__restrict__ done_all,
                                                       uint32_t *__restrict__ cand_all, int2 *__restrict__ span_all, int cand_cap, int item_stride,
                                                       int *__restrict__ overflow, int *__restrict__ total)
{
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    FrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    const plf_pose_pair P = poses[f];
    const int *claim = match_all + (size_t)f * kp_stride;   // occupancy before this call
    uint32_t *cand = cand_all + (size_t)f * cand_cap;
    float twc[3], tlc[3];
    for (int k = 0; k < 3; k++)
        twc[k] = (float)(-((double)P.Rcw[k] * P.tcw[0] + (double)P.Rcw[3 + k] * P.tcw[1] + (double)P.Rcw[6 + k] * P.tcw[2]));
    for (int k = 0; k < 3; k++) tlc[k] = P.Rlw[k * 3] * twc[0] + P.Rlw[k * 3 + 1] * twc[1] + P.Rlw[k * 3 + 2] * twc[2] + P.tlw[k];
    const bool bForward = tlc[2] > P.b && !mono, bBackward = -tlc[2] > P.b && !mono;
    const bool in = i < Lf.n;
    bool act = in && Lf.has_mp[i] && !Lf.outlier[i];
    float u = 0.f, v = 0.f, invzc = 0.f, rad = 0.f;
    int oct = 0, ub = 0;
    CellWin w;
    w.ok = false;
    if (act) {
        const float *xw = Lf.xw + 3 * (size_t)i;
        const float xc = P.Rcw[0] * xw[0] + P.Rcw[1] * xw[1] + P.Rcw[2] * xw[2] + P.tcw[0];
        const float yc = P.Rcw[3] * xw[0] + P.Rcw[4] * xw[1] + P.Rcw[5] * xw[2] + P.tcw[1];
        const float zc = P.Rcw[6] * xw[0] + P.Rcw[7] * xw[1] + P.Rcw[8] * xw[2] + P.tcw[2];
        invzc = (float)(1.0 / (double)zc);
        if (invzc < 0) act = false;
        u = fmaf(P.fx * xc, invzc, P.cx); v = fmaf(P.fy * yc, invzc, P.cy);   // FMA contractions of the binary: so@0x81cba, so@0x81cd9
        if (u < F.min_x || u > F.max_x) act = false;
        if (v < F.min_y || v > F.max_y) act = false;
        if (act) {
            oct = Lf.keys[i].octave;
            rad = th * F.scale_factors[oct];
            w = cell_window(F, u, v, rad);
            if (w.ok)
                for (int ix = w.x0; ix <= w.x1; ix++) ub += F.cell_start[ix * GRID_ROWS + w.y1 + 1] - F.cell_start[ix * GRID_ROWS + w.y0];
        }
    }
    const int excl = plf_wave_excl_scan(ub), wsum = plf_wave_sum(ub);
    int base = 0;
    if (wsum > 0) {
        if (plf_lane() == 0) base = atomicAdd(&total[f], wsum);
        base = __shfl(base, 0, 64);
    }
    const bool fits = base + wsum <= cand_cap;
    if (!fits && plf_lane() == 0) overflow[f] = 1;
    if (!in) return;
    const int o0 = base + excl;
    int o = o0;
    if (ub > 0 && fits) {
        const int minL = bForward ? oct : (bBackward ? 0 : oct - 1), maxL = bForward ? -1 : (bBackward ? oct : oct + 1);
        const bool chk = (minL > 0) || (maxL >= 0);
        const uint4 *dp = reinterpret_cast<const uint4 *>(Lf.mp_desc + (size_t)i * 32);
        const uint4 d0 = dp[0], d1 = dp[1];
        const float ur = fmaf(-P.bf, invzc, u);   // so@0x81eb5
        for (int ix = w.x0; ix <= w.x1; ix++) {
            const int j1 = F.cell_start[ix * GRID_ROWS + w.y1 + 1];
            for (int j = F.cell_start[ix * GRID_ROWS + w.y0]; j < j1; j++) {
                const float4 e = F.cell_kp[j];
                const int ko = __float_as_int(e.z), idx = __float_as_int(e.w);
                if (chk) {
                    if (ko < minL) continue;
                    if (maxL >= 0 && ko > maxL) continue;
                }
                if (!(fabsf(e.x - u) < rad && fabsf(e.y - v) < rad)) continue;
                if (blocked(claim, idx, Lf.obs_positive)) continue;
                if (F.uright) { const float urr = F.uright[idx]; if (urr > 0 && fabsf(ur - urr) > rad) continue; }
                const uint4 *kp = reinterpret_cast<const uint4 *>(F.desc + (size_t)idx * 32);
                const uint4 k0 = kp[0], k1 = kp[1];
                const int dist = __popc(d0.x ^ k0.x) + __popc(d0.y ^ k0.y) + __popc(d0.z ^ k0.z) + __popc(d0.w ^ k0.w) + __popc(d1.x ^ k1.x) +
                                 __popc(d1.y ^ k1.y) + __popc(d1.z ^ k1.z) + __popc(d1.w ^ k1.w);
                cand[o++] = (uint32_t)idx | ((uint32_t)dist << 16);
            }
        }
    }
    span_all[(size_t)f * item_stride + i] = make_int2(o0, o - o0);
    done_all[(size_t)f * item_stride + i] = (o > o0) ? 0 : 1;
}

// LDS: claim[kp_cap], owner[kp_cap] (int); la, lb (unfinished points) and assign (key point of a finished point, 0xFFFF none): uint16[item_cap] each
__global__ void __launch_bounds__(256) k_lf_rounds(const FrameDev *__restrict__ frames, LastDev Lf, int check_ori, int *__restrict__ match_all, int kp_stride,
                                                   int *__restrict__ nmatches, const uint8_t *__restrict__ done_all, int kp_cap, int item_cap,
                                                   const uint32_t *__restrict__ cand_all, const int2 *__restrict__ span_all, int cand_cap, int item_stride,
                                                   const int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + kp_cap;
    uint16_t *la = (uint16_t *)(owner + kp_cap), *lb = la + item_cap, *assign = lb + item_cap;
    __shared__ int s_n[2], s_acc, hist[HISTO_LENGTH], keepbin[3];
    const int f = blockIdx.x, t = threadIdx.x, T = blockDim.x, lane = plf_lane();
    if (overflow[f]) return;
    FrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int *match = match_all + (size_t)f * kp_stride;
    const uint8_t *done = done_all + (size_t)f * item_stride;
    const uint32_t *cand = cand_all + (size_t)f * cand_cap;
    const int2 *span = span_all + (size_t)f * item_stride;
    const uint8_t *obs = Lf.obs_positive;
    for (int k = t; k < F.n; k += T) claim[k] = match[k] < -2 ? -1 : match[k];   // (-3 marks of an earlier check_orientation = 2 call read as free; -2 stays "occupied")
    for (int i = t; i < Lf.n; i += T) assign[i] = 0xFFFFu;
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) { s_acc = 0; s_n[0] = 0; s_n[1] = 0; }
    __syncthreads();
    for (int m0 = 0; m0 < Lf.n; m0 += T) {
        const int m = m0 + t;
        const bool un = m < Lf.n && !done[m];
        const unsigned long long mask = __ballot(un);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&s_n[0], __popcll(mask));
        base = __shfl(base, 0, 64);
        if (un) la[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)m;
    }
    __syncthreads();
    int cur = 0;
    for (int round = 0; round <= Lf.n; round++) {
        const int nact = s_n[cur];
        if (nact == 0) break;
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        __syncthreads();
        for (int q = t; q < nact; q += T) {
            const int i = la[q];
            const int2 sp = span[i];
            for (int j = sp.x; j < sp.x + sp.y; j++) {
                const int idx = (int)(cand[j] & 0xFFFF);
                if (!blocked(claim, idx, obs)) atomicMin(&owner[idx], i);
            }
        }
        __syncthreads();
        for (int q0 = 0; q0 < nact; q0 += T) {
            const int q = q0 + t;
            bool keep = false;
            int i = 0;
            if (q < nact) {
                i = la[q];
                const int2 sp = span[i];
                int bestDist = 256, bestIdx = -1;
                for (int j = sp.x; j < sp.x + sp.y; j++) {
                    const uint32_t e = cand[j];
                    const int idx = (int)(e & 0xFFFF);
                    if (blocked(claim, idx, obs)) continue;
                    const int dist = (int)(e >> 16);
                    if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
                }
                const bool safe = bestIdx < 0 || owner[bestIdx] == i;
                keep = !safe;
                if (safe && bestDist <= TH_HIGH) {
                    claim[bestIdx] = i;   // only this point can touch bestIdx in this round
                    assign[i] = (uint16_t)bestIdx;
                    atomicAdd(&s_acc, 1);
                    if (check_ori) {
                        float rot = Lf.keys[i].angle - F.keys[bestIdx].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * (1.0f / 12.0f));  // this binary: HISTO_LENGTH / 360 (so@0x829b5)
                        if (bin == HISTO_LENGTH) bin = 0;
                        atomicAdd(&hist[bin], 1);
                    }
                }
            }
            const unsigned long long mask = __ballot(keep);
            int base = 0;
            if (lane == 0 && mask) base = atomicAdd(&s_n[cur ^ 1], __popcll(mask));
            base = __shfl(base, 0, 64);
            if (keep) lb[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)i;
        }
        __syncthreads();
        if (t == 0) s_n[cur] = 0;
        uint16_t *tmp = la; la = lb; lb = tmp;
        cur ^= 1;
        __syncthreads();
    }
    if (check_ori) {
        if (t == 0) {  // ComputeThreeMaxima (so@0x823eb)
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int b = 0; b < HISTO_LENGTH; b++) {
                const int sz = hist[b];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = b; }
                else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = b; }
                else if (sz > max3) { max3 = sz; i3 = b; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            keepbin[0] = i1; keepbin[1] = i2; keepbin[2] = i3;
        }
        __syncthreads();
        for (int i = t; i < Lf.n; i += T) {   // per ASSIGNMENT, as the reference's rotHist entries
            const int k = assign[i];
            if (k == 0xFFFF) continue;
            float rot = Lf.keys[i].angle - F.keys[k].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / 12.0f));
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin != keepbin[0] && bin != keepbin[1] && bin != keepbin[2]) { claim[k] = check_ori == 2 ? -3 : -1; atomicSub(&s_acc, 1); }
        }
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) nmatches[f] = s_acc;
}

// ------------------------------------------------------------------------------------------------
// "Project map points into a keyframe and take the best key point": the search half of
//   ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th)                      include/ORBmatcher.h:119, so@0x7a500
//   ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, th, ...)    include/ORBmatcher.h:122, so@0x7bb20
//   ORBmatcher::SearchBySim3 (each direction)                                      include/ORBmatcher.h:116, so@0x838b0
// One thread per map point: the searches read only immutable keyframe data, every map point is independent of the
// others (the map mutations that follow in Fuse stay on the host).  KeyFrame::GetFeaturesInArea (so@0x96fe0) is the
// Frame cell walk without a level filter; KeyFrame::IsInImage (so@0x97480) is half-open.
// ------------------------------------------------------------------------------------------------
struct Pts3Dev { int m; const float *xw, *normal, *min_dist, *max_dist; const uint8_t *desc, *valid; };
struct ProjKf {
    float R[9], t[3];      // camera <- world
    float R2[9], t2[3];    // second stage (SearchBySim3: sR21 / t21 applied to the camera-1 point)
    float Ow[3];
    float fx, fy, cx, cy, bf, log_scale;
    float inv_sigma2[16];
    int two_stage;         // 1: p = R2 * (R * xw + t) + t2, dist3D = |p|; 0: p = R * xw + t, dist3D = |xw - Ow|
    int view_test;         // PO . Pn < 0.5 * dist3D rejects
    int chi2;              // reprojection test of Fuse(KeyFrame*, ...)
    int accept;            // TH_LOW / TH_HIGH
};

// gates of one map point: camera point, KeyFrame::IsInImage, invariance range, viewing angle; predicted level and search radius
__device__ __forceinline__ bool project_gate(const FrameDev &F, const Pts3Dev &P, const ProjKf &C, float th, int i, float &u, float &v, float &invz,
                                             int &lvl, float &radius)
{
    if (!P.valid[i]) return false;
    const float *xw = P.xw + 3 * (size_t)i;
    float X = C.R[0] * xw[0] + C.R[1] * xw[1] + C.R[2] * xw[2] + C.t[0];
    float Y = C.R[3] * xw[0] + C.R[4] * xw[1] + C.R[5] * xw[2] + C.t[1];
    float Z = C.R[6] * xw[0] + C.R[7] * xw[1] + C.R[8] * xw[2] + C.t[2];
    if (C.two_stage) {
        const float a = X, b = Y, c = Z;
        X = C.R2[0] * a + C.R2[1] * b + C.R2[2] * c + C.t2[0];
        Y = C.R2[3] * a + C.R2[4] * b + C.R2[5] * c + C.t2[1];
        Z = C.R2[6] * a + C.R2[7] * b + C.R2[8] * c + C.t2[2];
    }
    if (Z < 0.0f) return false;
    invz = 1.0f / Z;
    const float x = X * invz, y = Y * invz;
    // contracted in the binary (so@0x7abc8 / 0x7abf0, so@0x7caac / 0x7cabe, so@0x8914e / 0x89160, so@0x84af1 / 0x84b03)
    u = fmaf(x, C.fx, C.cx); v = fmaf(C.fy, y, C.cy);
    if (!(u >= F.min_x && u < F.max_x && v >= F.min_y && v < F.max_y)) return false;
    const float PO[3] = {C.two_stage ? X : xw[0] - C.Ow[0], C.two_stage ? Y : xw[1] - C.Ow[1], C.two_stage ? Z : xw[2] - C.Ow[2]};
    double s2 = 0;
    for (int k = 0; k < 3; k++) s2 += (double)PO[k] * (double)PO[k];
    const float dist3D = (float)sqrt(s2);
    const float maxDistance = 1.2f * P.max_dist[i], minDistance = 0.8f * P.min_dist[i];
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    if (C.view_test) {
        const float *Pn = P.normal + 3 * (size_t)i;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn[k];
        if (dot < 0.5 * (double)dist3D) return false;
    }
    // MapPoint::PredictScale(float, KeyFrame*) so@0x8fb60 (logf of glibc: the correctly rounded value, see k_match_lastframe)
    const float ratio = P.max_dist[i] / dist3D;
    lvl = (int)ceilf((float)log((double)ratio) / C.log_scale);
    if (lvl < 0) lvl = 0;
    else if (lvl >= F.nlevels) lvl = F.nlevels - 1;
    radius = th * F.scale_factors[lvl];
    return true;
}

__global__ void __launch_bounds__(256) k_project_kf(FrameDev F, Pts3Dev P, ProjKf C, float th, int *__restrict__ best_idx,
                                                    int *__restrict__ best_dist, int *__restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.m) return;
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int bestDist = 256, bestIdx = -1, lvl = 0;
    float u = 0.f, v = 0.f, invz = 0.f, radius = 0.f;
    if (project_gate(F, P, C, th, i, u, v, invz, lvl, radius)) {
        const CellWin w = cell_window(F, u, v, radius);
        const uint8_t *d = P.desc + 32 * (size_t)i;
        if (w.ok) FOR_EACH_CANDIDATE(F, w, u, v, radius, -1, -1, idx, {
            const int kpLevel = _kp.octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            if (C.chi2) {
                const float ey = v - _kp.y;
                const float ex = u - _kp.x;
                const float kur = F.uright ? F.uright[idx] : -1.f;
                if (kur >= 0.0f) {
                    const float ur = fmaf(-C.bf, invz, u);   // so@0x7b63e
                    const float er = ur - kur;
                    const float e2 = fmaf(er, er, fmaf(ex, ex, ey * ey));
                    if ((double)(e2 * C.inv_sigma2[kpLevel]) > 7.8) continue;
                } else {
                    const float e2 = fmaf(ex, ex, ey * ey);
                    if ((double)(e2 * C.inv_sigma2[kpLevel]) > 5.99) continue;
                }
            }
            const int dist = hamming_g(d, F.desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        })
    }
    const bool hit = bestIdx >= 0 && bestDist <= C.accept;
    best_idx[i] = hit ? bestIdx : -1;
    if (best_dist) best_dist[i] = bestDist;
    if (count && hit) atomicAdd(count, 1);
}

// ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>& vpMatched, int th)
// include/ORBmatcher.h:86, so@0x880f0.  Greedy in list order through vpMatched: same conflict-free rounds as k_match_lastframe.
// One block; proj[i] = (u, v, radius, level bits).
__global__ void __launch_bounds__(256) k_project_kf_greedy(FrameDev F, Pts3Dev P, ProjKf C, float th, int *__restrict__ match, int *__restrict__ nmatches,
                                                           uint8_t *__restrict__ done, float4 *__restrict__ proj, int kp_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + kp_cap;
    __shared__ int s_left, s_acc;
    const int t = threadIdx.x, T = blockDim.x;
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    for (int k = t; k < F.n; k += T) claim[k] = match[k] < -2 ? -1 : match[k];   // (-3 marks of an earlier check_orientation = 2 call read as free; -2 stays "occupied")
    if (t == 0) s_acc = 0;
    for (int i = t; i < P.m; i += T) {
        float u = 0.f, v = 0.f, invz = 0.f, radius = 0.f;
        int lvl = 0;
        const bool act = project_gate(F, P, C, th, i, u, v, invz, lvl, radius);
        proj[i] = make_float4(u, v, radius, __int_as_float(lvl));
        done[i] = act ? 0 : 1;
    }
    __syncthreads();
    for (int round = 0; round <= P.m; round++) {
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        if (t == 0) s_left = 0;
        __syncthreads();
        for (int i = t; i < P.m; i += T) {
            if (done[i]) continue;
            const float4 pr = proj[i];
            const int lvl = __float_as_int(pr.w);
            const CellWin w = cell_window(F, pr.x, pr.y, pr.z);
            if (w.ok) FOR_EACH_CANDIDATE(F, w, pr.x, pr.y, pr.z, -1, -1, idx, {
                if (_kp.octave < lvl - 1 || _kp.octave > lvl) continue;
                if (claim[idx] == -1) atomicMin(&owner[idx], i);
            })
        }
        __syncthreads();
        for (int i = t; i < P.m; i += T) {
            if (done[i]) continue;
            const float4 pr = proj[i];
            const int lvl = __float_as_int(pr.w);
            const CellWin w = cell_window(F, pr.x, pr.y, pr.z);
            bool safe = true;
            if (w.ok) FOR_EACH_CANDIDATE(F, w, pr.x, pr.y, pr.z, -1, -1, idx, {
                if (_kp.octave < lvl - 1 || _kp.octave > lvl) continue;
                if (claim[idx] == -1 && owner[idx] != i) safe = false;
            })
            if (!safe) { atomicAdd(&s_left, 1); continue; }
            int bestDist = 256, bestIdx = -1;
            const uint8_t *d = P.desc + 32 * (size_t)i;
            if (w.ok) FOR_EACH_CANDIDATE(F, w, pr.x, pr.y, pr.z, -1, -1, idx, {
                if (claim[idx] != -1) continue;
                if (_kp.octave < lvl - 1 || _kp.octave > lvl) continue;
                const int dist = hamming_g(d, F.desc + 32 * (size_t)idx);
                if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
            })
            done[i] = 1;
            if (bestIdx >= 0 && bestDist <= C.accept) { claim[bestIdx] = i; atomicAdd(&s_acc, 1); }
        }
        __syncthreads();
        if (s_left == 0) break;
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) *nmatches = s_acc;
}

// SearchBySim3, last loop (so@0x838b0): a pair stands when both directions chose each other
__global__ void __launch_bounds__(256) k_sim3_agree(const int *__restrict__ vn1, int n1, const int *__restrict__ vn2, int n2, int *__restrict__ match12,
                                                    int *__restrict__ nfound)
{
    const int i1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i1 >= n1) return;
    const int idx2 = vn1[i1];
    const bool ok = idx2 >= 0 && idx2 < n2 && vn2[idx2] == i1;
    match12[i1] = ok ? idx2 : -1;
    if (ok) atomicAdd(nfound, 1);
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)   include/ORBmatcher.h:104, so@0x80150
// One block per (keyframe, frame) pair.  The two DBoW2 feature vectors arrive flattened (node ids ascending, CSR).
// The reference walks the common nodes in key order; a node only reads / writes the match slots of its own frame
// features, so as long as no frame feature sits in two common nodes (never the case for a DBoW2 FeatureVector, where
// every feature has exactly one node at the chosen level) the nodes are independent and each thread takes whole
// nodes.  Otherwise thread 0 replays the reference loop alone (same result, no parallelism).
// ------------------------------------------------------------------------------------------------
struct BowDev {
    int n_kf, n_f;
    const uint8_t *kf_desc, *f_desc;
    const float *kf_angle, *f_angle;
    const uint8_t *kf_has_mp, *f_has_mp;   // f_has_mp: second keyframe of the (KeyFrame, KeyFrame) overload, NULL for a Frame
    int kf_nodes, f_nodes;
    const uint32_t *kf_node_id, *f_node_id;
    const int *kf_node_start, *f_node_start;
    const int *kf_feat, *f_feat;
};

// one common node.  kfkf = 0: SearchByBoW(KeyFrame*, Frame&): slot = frame feature, value = keyframe feature, accept best <= TH_LOW.
// kfkf = 1: SearchByBoW(KeyFrame*, KeyFrame*) (include/ORBmatcher.h:105, so@0x82cc0): slot = KF1 feature, value = KF2 feature,
// KF2 features need a good map point and are marked in used2 (vbMatched2), accept best < TH_LOW (so@0x83490).
__device__ __forceinline__ int bow_node(const BowDev &P, int a, int b, float nnratio, int kfkf, int *__restrict__ match, int *__restrict__ used2)
{
    int acc = 0;
    for (int p = P.kf_node_start[a]; p < P.kf_node_start[a + 1]; p++) {
        const int ikf = P.kf_feat[p];
        if (!P.kf_has_mp[ikf]) continue;
        const uint8_t *dKF = P.kf_desc + (size_t)ikf * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int q = P.f_node_start[b]; q < P.f_node_start[b + 1]; q++) {
            const int jf = P.f_feat[q];
            if (kfkf ? (used2[jf] != 0 || !P.f_has_mp[jf]) : (match[jf] >= 0)) continue;
            const int dist = hamming_g(dKF, P.f_desc + (size_t)jf * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = jf; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if ((kfkf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) && (float)bestDist1 < nnratio * (float)bestDist2) {
            if (kfkf) { match[ikf] = bestIdxF; used2[bestIdxF] = 1; }
            else match[bestIdxF] = ikf;
            acc++;
        }
    }
    return acc;
}

// kfkf = 2: ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)  include/ORBmatcher.h:111, so@0x86b30 -- the same node
// walk; candidates are features WITHOUT a map point (kf_has_mp / f_has_mp = GetMapPoint(i) != NULL are skipped), gated by the epipole distance
// (two mono key points) and ORBmatcher::CheckDistEpipolarLine (so@0x79b90, contractions as in the binary).
struct TriDev { const plf_keypoint *keys1, *keys2; const float *uright1, *uright2, *scale2, *sigma2_2; float F[9]; float ex, ey; int only_stereo; };

__device__ __forceinline__ bool check_dist_epipolar_line(const plf_keypoint &k1, const plf_keypoint &k2, const float *F, float level_sigma2)
{
    const float b = fmaf(k1.x, F[1], k1.y * F[4]) + F[7];
    const float a = fmaf(k1.x, F[0], k1.y * F[3]) + F[6];
    const float den = fmaf(a, a, b * b);
    if (den == 0.0f) return false;
    const float c = fmaf(k1.y, F[5], k1.x * F[2]) + F[8];
    const float num = c + fmaf(b, k2.y, a * k2.x);
    const float dsqr = num * num / den;
    return 3.84 * (double)level_sigma2 > (double)dsqr;
}

__device__ __forceinline__ int tri_node(const BowDev &P, const TriDev &T, int a, int b, int *__restrict__ match, int *__restrict__ used2)
{
    int acc = 0;
    for (int p = P.kf_node_start[a]; p < P.kf_node_start[a + 1]; p++) {
        const int idx1 = P.kf_feat[p];
        if (P.kf_has_mp[idx1]) continue;
        const bool bStereo1 = T.uright1[idx1] >= 0.0f;
        if (T.only_stereo && !bStereo1) continue;
        const plf_keypoint kp1 = T.keys1[idx1];
        const uint8_t *d1 = P.kf_desc + (size_t)idx1 * 32;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int q = P.f_node_start[b]; q < P.f_node_start[b + 1]; q++) {
            const int idx2 = P.f_feat[q];
            if (used2[idx2] != 0 || P.f_has_mp[idx2]) continue;
            const bool bStereo2 = T.uright2[idx2] >= 0.0f;
            if (T.only_stereo && !bStereo2) continue;
            const int dist = hamming_g(d1, P.f_desc + (size_t)idx2 * 32);
            if (dist > TH_LOW || dist > bestDist) continue;   // (an equal distance replaces the earlier candidate, so@0x87a0d)
            const plf_keypoint kp2 = T.keys2[idx2];
            if (!bStereo1 && !bStereo2) {
                const float distex = T.ex - kp2.x, distey = T.ey - kp2.y;
                if (fmaf(distex, distex, distey * distey) < 100.0f * T.scale2[kp2.octave]) continue;
            }
            if (check_dist_epipolar_line(kp1, kp2, T.F, T.sigma2_2[kp2.octave])) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) { match[idx1] = bestIdx2; used2[bestIdx2] = 1; acc++; }
    }
    return acc;
}

__global__ void __launch_bounds__(256) k_match_bow(const BowDev *__restrict__ pairs, float nnratio, int check_ori, int kfkf, int *__restrict__ match_all,
                                                   int stride, int *__restrict__ nmatches, int *__restrict__ fnode_all, int *__restrict__ used_all, TriDev TR)
{
    __shared__ int hist[HISTO_LENGTH], keepbin[3], s_acc, s_shared, s_dup1;
    const int pr = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    const BowDev P = pairs[pr];
    int *match = match_all + (size_t)pr * stride;
    int *fnode = fnode_all + (size_t)pr * stride;   // second-side feature -> first common node that lists it (scratch)
    int *used2 = used_all + (size_t)pr * stride;    // kfkf: vbMatched2, then reused to detect KF1 features listed twice
    const int nslot = kfkf ? P.n_kf : P.n_f;
    for (int j = t; j < nslot; j += T) match[j] = -1;
    for (int j = t; j < P.n_f; j += T) { fnode[j] = 0x7fffffff; used2[j] = 0; }
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) { s_acc = 0; s_shared = 0; s_dup1 = 0; }
    __syncthreads();
    // common nodes: binary search of every first-side node in the second side's node list; detect second-side features listed twice
    for (int a = t; a < P.kf_nodes; a += T) {
        const uint32_t id = P.kf_node_id[a];
        int lo = 0, hi = P.f_nodes;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.f_node_id[mid] < id) lo = mid + 1; else hi = mid; }
        if (lo < P.f_nodes && P.f_node_id[lo] == id)
            for (int q = P.f_node_start[lo]; q < P.f_node_start[lo + 1]; q++)
                if (atomicMin(&fnode[P.f_feat[q]], a) != 0x7fffffff) s_shared = 1;
    }
    if (kfkf) {   // a KF1 feature listed twice would make the nodes depend on each other through its output slot
        for (int q = t; q < P.kf_node_start[P.kf_nodes]; q += T)
            if (atomicExch(&match[P.kf_feat[q]], -3) == -3) s_dup1 = 1;
        __syncthreads();
        for (int j = t; j < nslot; j += T) match[j] = -1;
    }
    __syncthreads();
    if (kfkf && s_dup1) {   // not a DBoW2 feature vector: refuse instead of guessing
        if (t == 0) nmatches[pr] = -1;
        return;
    }
    if (!s_shared) {
        int acc = 0;
        for (int a = t; a < P.kf_nodes; a += T) {
            const uint32_t id = P.kf_node_id[a];
            int lo = 0, hi = P.f_nodes;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.f_node_id[mid] < id) lo = mid + 1; else hi = mid; }
            if (lo < P.f_nodes && P.f_node_id[lo] == id) acc += kfkf == 2 ? tri_node(P, TR, a, lo, match, used2) : bow_node(P, a, lo, nnratio, kfkf, match, used2);
        }
        if (acc) atomicAdd(&s_acc, acc);
    } else if (t == 0) {
        int a = 0, b = 0, acc = 0;
        while (a < P.kf_nodes && b < P.f_nodes) {
            if (P.kf_node_id[a] < P.f_node_id[b]) { a++; continue; }
            if (P.kf_node_id[a] > P.f_node_id[b]) { b++; continue; }
            acc += kfkf == 2 ? tri_node(P, TR, a, b, match, used2) : bow_node(P, a, b, nnratio, kfkf, match, used2);
            a++; b++;
        }
        s_acc = acc;
    }
    __syncthreads();
    if (check_ori) {
        for (int pass = 0; pass < 2; pass++) {
            for (int j = t; j < nslot; j += T) {
                const int i = match[j];
                if (i < 0) continue;
                float rot = kfkf == 2 ? TR.keys1[j].angle - TR.keys2[i].angle : (kfkf ? P.kf_angle[j] - P.f_angle[i] : P.kf_angle[i] - P.f_angle[j]);
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * (1.0f / 12.0f));
                if (bin == HISTO_LENGTH) bin = 0;
                if (pass == 0) atomicAdd(&hist[bin], 1);
                else if (bin != keepbin[0] && bin != keepbin[1] && bin != keepbin[2]) { match[j] = -1; atomicSub(&s_acc, 1); }
            }
            __syncthreads();
            if (pass == 0 && t == 0) {  // ComputeThreeMaxima (so@0x79c40)
                int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
                for (int i = 0; i < HISTO_LENGTH; i++) {
                    const int sz = hist[i];
                    if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
                    else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
                    else if (sz > max3) { max3 = sz; i3 = i; }
                }
                if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
                else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
                keepbin[0] = i1; keepbin[1] = i2; keepbin[2] = i3;
            }
            __syncthreads();
        }
    }
    if (t == 0) nmatches[pr] = s_acc;
}

// cv::batchDistance, K = 2: strict '<' against the current worst, equal distances keep the earlier train index first
__global__ void __launch_bounds__(128) k_knn2(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ tr, int nt,
                                              int *__restrict__ idx, int *__restrict__ dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int d0 = 0x7fffffff, d1 = 0x7fffffff, i0 = -1, i1 = -1;
    for (int j = 0; j < nt; j++) {
        const int d = hamming_g(q + 32 * (size_t)i, tr + 32 * (size_t)j);
        if (d < d1) {
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else { d1 = d; i1 = j; }
        }
    }
    idx[2 * i] = i0; idx[2 * i + 1] = i1;
    dist[2 * i] = d0; dist[2 * i + 1] = d1;
}

// the same brute-force 2-NN for a batch of current frames against ONE query set (the last frame's line descriptors): blockIdx.y = frame, the
// frame's descriptors are staged through LDS in tiles of 128 so that every thread (= query) reads them as broadcasts
struct LineFrameDev { int n; const int *n_dev; const plf_keyline *lines; const uint8_t *desc; const float *scale_factors; };
__global__ void __launch_bounds__(128) k_knn2_batch(const uint8_t *__restrict__ q, int nq, const LineFrameDev *__restrict__ frames, int *__restrict__ idx_all,
                                                    int *__restrict__ dist_all, int stride)
{
    __shared__ uint32_t tile[128][9];   // (+1: conflict-free fill)
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, t = threadIdx.x;
    const uint8_t *tr = frames[f].desc;
    int nt = frames[f].n;
    if (frames[f].n_dev) nt = min(nt, *frames[f].n_dev);
    uint32_t qd[8];
    if (i < nq) {
        const uint4 *qp = reinterpret_cast<const uint4 *>(q + 32 * (size_t)i);
        const uint4 a = qp[0], b = qp[1];
        qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
    }
    int d0 = 0x7fffffff, d1 = 0x7fffffff, i0 = -1, i1 = -1;
    for (int j0 = 0; j0 < nt; j0 += 128) {
        const int m = min(128, nt - j0);
        __syncthreads();
        for (int k = t; k < m * 8; k += 128) tile[k >> 3][k & 7] = reinterpret_cast<const uint32_t *>(tr + 32 * (size_t)j0)[k];
        __syncthreads();
        if (i < nq)
            for (int j = 0; j < m; j++) {
                int d = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) d += __popc(qd[k] ^ tile[j][k]);
                if (d < d1) {
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j0 + j; }
                    else { d1 = d; i1 = j0 + j; }
                }
            }
    }
    if (i >= nq) return;
    int *idx = idx_all + (size_t)f * stride, *dist = dist_all + (size_t)f * stride;
    idx[2 * i] = i0; idx[2 * i + 1] = i1;
    dist[2 * i] = d0; dist[2 * i + 1] = d1;
}

__global__ void __launch_bounds__(128) k_knn2_to_dmatch(const int *__restrict__ idx, const int *__restrict__ dist, int nq, plf_dmatch *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * nq) return;
    plf_dmatch m;
    m.queryIdx = i >> 1; m.trainIdx = idx[i]; m.imgIdx = 0; m.distance = (float)dist[i];
    out[i] = m;
}

__device__ void block_sort_floats(float *v, int n, int P2)
{
    const int t = threadIdx.x, T = blockDim.x;
    for (int k2 = 2; k2 <= P2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P2; i += T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float a = v[i], b = v[ixj];
                    const bool asc = (i & k2) == 0;
                    if (asc ? (a > b) : (a < b)) { v[i] = b; v[ixj] = a; }
                }
            }
            __syncthreads();
        }
    (void)n;
}

// Frame::lineDescriptorMAD (include/Frame.h:75) / LineSegment::LineDescriptorMAD (include/ExtractLineSegment.h:44) on a 2-NN table of n queries:
// mad[0] = nn_mad = 1.4826 * median |d1 - median(d1)|, mad[1] = nn12_mad the same on d2 - d1 (float distances, double medians: oracle orc_line_mad).
// One block, n <= P2 (power of two) floats of LDS.
__global__ void __launch_bounds__(256) k_line_mad(const int *__restrict__ dist, int n, int P2, double *__restrict__ mad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *v = (float *)smem;
    const int t = threadIdx.x, T = blockDim.x;
    const float INF = 3.0e38f;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = t; i < P2; i += T) v[i] = i < n ? (pass ? (float)dist[2 * i + 1] - (float)dist[2 * i] : (float)dist[2 * i]) : INF;
        __syncthreads();
        block_sort_floats(v, n, P2);
        const double med = v[n / 2];
        __syncthreads();
        for (int i = t; i < P2; i += T)
            v[i] = i < n ? fabsf((float)((double)(pass ? (float)dist[2 * i + 1] - (float)dist[2 * i] : (float)dist[2 * i]) - med)) : INF;
        __syncthreads();
        block_sort_floats(v, n, P2);
        if (t == 0) mad[pass] = 1.4826 * (double)v[n / 2];
        __syncthreads();
    }
}

// Frame::lineDescriptorMAD + LSDmatcher::SearchByProjection(CurrentFrame, LastFrame); one block per current frame, nlast <= P2 (power of two)
// tri != NULL: LSDmatcher::SearchForTriangulation (include/LSDmatcher.h:54) on the same 2-NN table -- queries = keyframe-1 lines, a pair (q, t) is kept
// when neither line holds a MapLine (and, with only_stereo, both have stereo data); match_all[q] = t.
struct LineTriDev { const uint8_t *has_ml1, *has_ml2, *stereo1, *stereo2; int only_stereo; };
__global__ void __launch_bounds__(256) k_lines_lastframe(const int *__restrict__ idx_all, const int *__restrict__ dist_all, int nlast,
                                                         const uint8_t *__restrict__ last_has_mapline, int *__restrict__ match_all,
                                                         int *__restrict__ nmatches_all, int P2, int knn_stride, int line_stride,
                                                         const LineFrameDev *__restrict__ frames, double mad_factor, LineTriDev tri)
{
    // blockIdx.x = current frame of the batch (one frame: strides unused, frames NULL)
    const int *idx = idx_all + (size_t)blockIdx.x * knn_stride, *dist = dist_all + (size_t)blockIdx.x * knn_stride;
    int *match_of_line = match_all + (size_t)blockIdx.x * line_stride, *nmatches = nmatches_all + blockIdx.x;
    if (frames) {   // fewer than two current lines: knnMatch(k = 2) has no second neighbour, nothing is matched
        int nc = frames[blockIdx.x].n;
        if (frames[blockIdx.x].n_dev) nc = min(nc, *frames[blockIdx.x].n_dev);
        if (nc < 2) { if (threadIdx.x == 0) *nmatches = 0; return; }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *v = (float *)smem;
    __shared__ int s_cnt;
    const int t = threadIdx.x, T = blockDim.x;
    const float INF = 3.0e38f;
    // nn12 MAD: median of (d1 - d0), then median of |d12 - median|
    for (int i = t; i < P2; i += T) v[i] = i < nlast ? (float)dist[2 * i + 1] - (float)dist[2 * i] : INF;
    if (t == 0) s_cnt = 0;
    __syncthreads();
    block_sort_floats(v, nlast, P2);
    const double med = v[nlast / 2];
    __syncthreads();
    for (int i = t; i < P2; i += T)
        v[i] = i < nlast ? fabsf((float)((double)((float)dist[2 * i + 1] - (float)dist[2 * i]) - med)) : INF;
    __syncthreads();
    block_sort_floats(v, nlast, P2);
    const double th12 = 1.4826 * (double)v[nlast / 2] * mad_factor;
    __syncthreads();
    for (int q = t; q < nlast; q += T) {
        const double d12 = (double)((float)dist[2 * q + 1] - (float)dist[2 * q]);
        if (!(d12 > th12)) continue;
        const int tr = idx[2 * q];
        if (tri.has_ml1) {
            if (tri.has_ml1[q] || tri.has_ml2[tr]) continue;
            if (tri.only_stereo && (!tri.stereo1[q] || !tri.stereo2[tr])) continue;
            match_of_line[q] = tr;
            atomicAdd(&s_cnt, 1);
        } else if (last_has_mapline[q]) {
            atomicMax(&match_of_line[tr], q);  // queries are visited in increasing order: the last one wins
            atomicAdd(&s_cnt, 1);
        }
    }
    __syncthreads();
    if (t == 0) *nmatches = s_cnt;
}

// LSDmatcher::Fuse (include/LSDmatcher.h:58), search half: best[i] = nearest keyframe line of map line i (first minimum of the brute-force scan:
// idx[2i] of the 2-NN table) when the map line is valid and that distance is <= TH_LOW
__global__ void __launch_bounds__(256) k_lines_fuse_pick(const int *__restrict__ idx, const int *__restrict__ dist, const uint8_t *__restrict__ valid, int m,
                                                         int *__restrict__ best, int *__restrict__ nfused)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (i < m) {
        ok = valid[i] && idx[2 * i] >= 0 && dist[2 * i] <= TH_LOW;
        best[i] = ok ? idx[2 * i] : -1;
    }
    const unsigned long long mk = __ballot(ok);
    if (plf_lane() == 0 && mk) atomicAdd(nfused, __popcll(mk));
}

struct MapLineDev { int m; const float *x1, *y1, *x2, *y2; const int *level; const float *view_cos; const uint8_t *in_view; const uint8_t *desc; };

// Frame::GetLinesInArea test for line i
__device__ __forceinline__ bool line_in_area(const plf_keyline &kl, float x1, float y1, float x2, float y2, float r, int minLevel, int maxLevel)
{
    const double dx = 0.5 * (double)(x1 + x2) - (double)kl.pt_x, dy = 0.5 * (double)(y1 + y2) - (double)kl.pt_y;
    const double distance = dx * dx + dy * dy;
    if (distance > (double)(r * r)) return false;
    const float slope = (y1 - y2) / (x1 - x2) - kl.angle;
    if ((double)slope > (double)r * 0.01) return false;
    if ((minLevel > 0) || (maxLevel > 0)) {
        if (kl.octave < minLevel) return false;
        if (maxLevel >= 0 && kl.octave > maxLevel) return false;
    }
    return true;
}

// the same test on the four fields of a key line staged in LDS (x = pt_x, y = pt_y, z = angle, w = octave bits): identical arithmetic
__device__ __forceinline__ bool line_in_area4(const float4 kl, float x1, float y1, float x2, float y2, float r, int minLevel, int maxLevel)
{
    const double dx = 0.5 * (double)(x1 + x2) - (double)kl.x, dy = 0.5 * (double)(y1 + y2) - (double)kl.y;
    const double distance = dx * dx + dy * dy;
    if (distance > (double)(r * r)) return false;
    const float slope = (y1 - y2) / (x1 - x2) - kl.z;
    if ((double)slope > (double)r * 0.01) return false;
    const int octave = __float_as_int(kl.w);
    if ((minLevel > 0) || (maxLevel > 0)) {
        if (octave < minLevel) return false;
        if (maxLevel >= 0 && octave > maxLevel) return false;
    }
    return true;
}

// STAGED: the frame's lines live in LDS (56 bytes per line; host: while line_cap * 56 fits the 150 KB the kernel may ask for, i.e. up to 2688 lines); otherwise
// they are read from global memory as before round 5 (8 bytes of LDS per line: plf_matcher_create accepts max_lines up to 18000 -- ADVICE r05)
template <bool STAGED>
__device__ __forceinline__ void match_project_lines_body(const LineFrameDev *__restrict__ frames, MapLineDev ML, float th, float nnratio,
                                                             int *__restrict__ match_all, int line_stride, int *__restrict__ nmatches,
                                                             uint8_t *__restrict__ done_all, int line_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + line_cap;
    // Round 5: the four fields of every key line that the window test reads and its 32 descriptor bytes are staged in LDS once (line_cap is a multiple of 64: 56 bytes
    // per line).  Every thread walks ALL lines of the frame three times per round for each of its map lines -- from global memory that was 0.30 ms for one frame
    // of 100 lines against 500 map lines, on the critical path of a single frame: the line matchers can only start when LSD + LBD are done.
    float4 *lg = reinterpret_cast<float4 *>(owner + line_cap);
    uint4 *ld = reinterpret_cast<uint4 *>(lg + line_cap);
    __shared__ int s_left, s_acc;
    const int f = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    LineFrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int *match = match_all + (size_t)f * line_stride;
    uint8_t *done = done_all + (size_t)f * ML.m;
    const bool bFactor = th != 1.0f;
    for (int k = t; k < F.n; k += T) {
        claim[k] = match[k] < -2 ? -1 : match[k];   // (-3 marks of an earlier check_orientation = 2 call read as free; -2 stays "occupied")
        if (STAGED) {
            const plf_keyline kl = F.lines[k];
            lg[k] = make_float4(kl.pt_x, kl.pt_y, kl.angle, __int_as_float(kl.octave));
            const uint4 *dsrc = reinterpret_cast<const uint4 *>(F.desc + 32 * (size_t)k);
            ld[2 * k] = dsrc[0]; ld[2 * k + 1] = dsrc[1];
        }
    }
    // the four window-test fields / the two descriptor halves of key line i
    auto line4 = [&](int i) -> float4 {
        if (STAGED) return lg[i];
        const plf_keyline &kl = F.lines[i];
        return make_float4(kl.pt_x, kl.pt_y, kl.angle, __int_as_float(kl.octave));
    };
    auto desc_half = [&](int i, int hlf) -> uint4 { return STAGED ? ld[2 * i + hlf] : reinterpret_cast<const uint4 *>(F.desc + 32 * (size_t)i)[hlf]; };
    if (t == 0) s_acc = 0;
    for (int m = t; m < ML.m; m += T) done[m] = ML.in_view[m] ? 0 : 1;
    __syncthreads();
    for (int round = 0; round <= ML.m; round++) {
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        if (t == 0) s_left = 0;
        __syncthreads();
        for (int m = t; m < ML.m; m += T) {
            if (done[m]) continue;
            const int lvl = ML.level[m];
            float r = radius_by_viewing_cos(ML.view_cos[m]);
            if (bFactor) r *= th;
            const float rad = r * F.scale_factors[lvl];
            const float mx1 = ML.x1[m], my1 = ML.y1[m], mx2 = ML.x2[m], my2 = ML.y2[m];
            for (int i = 0; i < F.n; i++)
                if (claim[i] == -1 && line_in_area4(line4(i), mx1, my1, mx2, my2, rad, lvl - 1, lvl)) atomicMin(&owner[i], m);
        }
        __syncthreads();
        for (int m = t; m < ML.m; m += T) {
            if (done[m]) continue;
            const int lvl = ML.level[m];
            float r = radius_by_viewing_cos(ML.view_cos[m]);
            if (bFactor) r *= th;
            const float rad = r * F.scale_factors[lvl];
            const float mx1 = ML.x1[m], my1 = ML.y1[m], mx2 = ML.x2[m], my2 = ML.y2[m];
            bool safe = true;
            for (int i = 0; i < F.n; i++)
                if (claim[i] == -1 && owner[i] != m && line_in_area4(line4(i), mx1, my1, mx2, my2, rad, lvl - 1, lvl)) safe = false;
            if (!safe) { atomicAdd(&s_left, 1); continue; }
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            const uint4 d0 = reinterpret_cast<const uint4 *>(ML.desc + 32 * (size_t)m)[0], d1 = reinterpret_cast<const uint4 *>(ML.desc + 32 * (size_t)m)[1];
            for (int i = 0; i < F.n; i++) {
                if (claim[i] != -1) continue;
                const float4 kl = line4(i);
                if (!line_in_area4(kl, mx1, my1, mx2, my2, rad, lvl - 1, lvl)) continue;
                const uint4 b0 = desc_half(i, 0), b1 = desc_half(i, 1);
                const int dist = __popc(d0.x ^ b0.x) + __popc(d0.y ^ b0.y) + __popc(d0.z ^ b0.z) + __popc(d0.w ^ b0.w) + __popc(d1.x ^ b1.x) + __popc(d1.y ^ b1.y) +
                                 __popc(d1.z ^ b1.z) + __popc(d1.w ^ b1.w);
                const int oct = __float_as_int(kl.w);
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct; bestIdx = i; }
                else if (dist < bestDist2) { bestLevel2 = oct; bestDist2 = dist; }
            }
            done[m] = 1;
            if (bestDist <= TH_HIGH) {
                if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
                claim[bestIdx] = m;
                atomicAdd(&s_acc, 1);
            }
        }
        __syncthreads();
        if (s_left == 0) break;
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) nmatches[f] = s_acc;
}

__global__ void __launch_bounds__(256) k_match_project_lines(const LineFrameDev *__restrict__ frames, MapLineDev ML, float th, float nnratio,
                                                             int *__restrict__ match_all, int line_stride, int *__restrict__ nmatches,
                                                             uint8_t *__restrict__ done_all, int line_cap)
{
    match_project_lines_body<true>(frames, ML, th, nnratio, match_all, line_stride, nmatches, done_all, line_cap);
}
__global__ void __launch_bounds__(256) k_match_project_lines_g(const LineFrameDev *__restrict__ frames, MapLineDev ML, float th, float nnratio,
                                                               int *__restrict__ match_all, int line_stride, int *__restrict__ nmatches,
                                                               uint8_t *__restrict__ done_all, int line_cap)
{
    match_project_lines_body<false>(frames, ML, th, nnratio, match_all, line_stride, nmatches, done_all, line_cap);
}

// Few frames in flight (the reference's own regime: one frame per TrackRGBD call): the same greedy assignment with one WAVE per map line and the frame's key lines
// across its lanes -- the window test of 64 lines at once, the best / second-best candidate by two wave-wide minima of (distance << 16 | line): the reference's
// loop `if (dist < best) {second = best; best = dist} else if (dist < second) second = dist` in ascending line order keeps exactly the lexicographic minimum and the
// lexicographic minimum of the rest.  The thread-per-map-line kernel above walks all lines three times per round on every thread: 0.19-0.25 ms for 100 lines against
// 500 map lines, all of it on the critical path of a single frame (the line matchers start when LSD + LBD are done); 16 waves per frame here.
// Rounds, phases and barriers are those of match_project_lines_body: a map line is matched in the round in which it owns every free line of its window.
__device__ __forceinline__ int wave_min_key(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__global__ void __launch_bounds__(1024) k_match_project_lines_w(const LineFrameDev *__restrict__ frames, MapLineDev ML, float th, float nnratio,
                                                                int *__restrict__ match_all, int line_stride, int *__restrict__ nmatches,
                                                                uint8_t *__restrict__ done_all, int line_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *claim = (int *)smem, *owner = claim + line_cap;
    float4 *lg = reinterpret_cast<float4 *>(owner + line_cap);
    uint4 *ld = reinterpret_cast<uint4 *>(lg + line_cap);
    __shared__ int s_left, s_acc;
    const int f = blockIdx.x, t = threadIdx.x, T = blockDim.x, lane = t & 63, wv = t >> 6, NW = T >> 6;
    LineFrameDev F = frames[f];
    if (F.n_dev) F.n = min(F.n, *F.n_dev);
    int *match = match_all + (size_t)f * line_stride;
    uint8_t *done = done_all + (size_t)f * ML.m;
    const bool bFactor = th != 1.0f;
    for (int k = t; k < F.n; k += T) {
        claim[k] = match[k] < -2 ? -1 : match[k];
        const plf_keyline kl = F.lines[k];
        lg[k] = make_float4(kl.pt_x, kl.pt_y, kl.angle, __int_as_float(kl.octave));
        const uint4 *dsrc = reinterpret_cast<const uint4 *>(F.desc + 32 * (size_t)k);
        ld[2 * k] = dsrc[0]; ld[2 * k + 1] = dsrc[1];
    }
    if (t == 0) s_acc = 0;
    for (int m = t; m < ML.m; m += T) done[m] = ML.in_view[m] ? 0 : 1;
    __syncthreads();
    const int nchunk = (F.n + 63) >> 6;
    for (int round = 0; round <= ML.m; round++) {
        for (int k = t; k < F.n; k += T) owner[k] = 0x7fffffff;
        if (t == 0) s_left = 0;
        __syncthreads();
        for (int m = wv; m < ML.m; m += NW) {
            if (done[m]) continue;
            const int lvl = ML.level[m];
            float r = radius_by_viewing_cos(ML.view_cos[m]);
            if (bFactor) r *= th;
            const float rad = r * F.scale_factors[lvl];
            const float mx1 = ML.x1[m], my1 = ML.y1[m], mx2 = ML.x2[m], my2 = ML.y2[m];
            for (int c = 0; c < nchunk; c++) {
                const int i = c * 64 + lane;
                if (i < F.n && claim[i] == -1 && line_in_area4(lg[i], mx1, my1, mx2, my2, rad, lvl - 1, lvl)) atomicMin(&owner[i], m);
            }
        }
        __syncthreads();
        for (int m = wv; m < ML.m; m += NW) {
            if (done[m]) continue;
            const int lvl = ML.level[m];
            float r = radius_by_viewing_cos(ML.view_cos[m]);
            if (bFactor) r *= th;
            const float rad = r * F.scale_factors[lvl];
            const float mx1 = ML.x1[m], my1 = ML.y1[m], mx2 = ML.x2[m], my2 = ML.y2[m];
            const uint4 d0 = reinterpret_cast<const uint4 *>(ML.desc + 32 * (size_t)m)[0], d1 = reinterpret_cast<const uint4 *>(ML.desc + 32 * (size_t)m)[1];
            bool unsafe = false;
            int k1 = 0x7fffffff, k2 = 0x7fffffff;      // this lane's two smallest keys (distance << 16 | line); its lines come in ascending order
            for (int c = 0; c < nchunk; c++) {
                const int i = c * 64 + lane;
                const bool cand = i < F.n && claim[i] == -1 && line_in_area4(lg[i], mx1, my1, mx2, my2, rad, lvl - 1, lvl);
                if (cand) {
                    if (owner[i] != m) unsafe = true;
                    const uint4 b0 = ld[2 * i], b1 = ld[2 * i + 1];
                    const int dist = __popc(d0.x ^ b0.x) + __popc(d0.y ^ b0.y) + __popc(d0.z ^ b0.z) + __popc(d0.w ^ b0.w) + __popc(d1.x ^ b1.x) + __popc(d1.y ^ b1.y) +
                                     __popc(d1.z ^ b1.z) + __popc(d1.w ^ b1.w);
                    if (dist < 256) {                  // (a distance of 256 never enters the reference's `dist < 256` chain)
                        const int key = (dist << 16) | i;
                        if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                    }
                }
            }
            if (__ballot(unsafe)) { if (lane == 0) atomicAdd(&s_left, 1); continue; }
            const int kb = wave_min_key(k1);
            const int ks = wave_min_key(k1 == kb ? k2 : k1);
            if (lane == 0) {
                done[m] = 1;
                if (kb != 0x7fffffff) {
                    const int bestDist = kb >> 16, bestIdx = kb & 0xFFFF;
                    const int bestLevel = __float_as_int(lg[bestIdx].w);
                    const int bestDist2 = ks != 0x7fffffff ? ks >> 16 : 256, bestLevel2 = ks != 0x7fffffff ? __float_as_int(lg[ks & 0xFFFF].w) : -1;
                    if (bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)) {
                        claim[bestIdx] = m;
                        atomicAdd(&s_acc, 1);
                    }
                }
            }
        }
        __syncthreads();
        if (s_left == 0) break;
        __syncthreads();
    }
    for (int k = t; k < F.n; k += T) match[k] = claim[k];
    if (t == 0) nmatches[f] = s_acc;
}

__global__ void __launch_bounds__(256) k_hamming_matrix(const uint8_t *__restrict__ a, int na, const uint8_t *__restrict__ b, int nb,
                                                        int *__restrict__ dist)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nb || i >= na) return;
    dist[(size_t)i * nb + j] = hamming_g(a + 32 * (size_t)i, b + 32 * (size_t)j);
}
