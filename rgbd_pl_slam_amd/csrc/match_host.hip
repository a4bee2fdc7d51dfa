// match_host.hip -- host side of the matchers: scratch handle, grid construction, launch sequence, C ABI.
// Mirrors the tracking overloads of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:44,61,78) and
// ORB_SLAM2::LSDmatcher (include/LSDmatcher.h:32,40,43).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "plf_common.h"

#define GRID_COLS 64
#define GRID_ROWS 48
#define GRID_CELLS (GRID_COLS * GRID_ROWS)

struct FrameDev {
    int n;
    const int *n_dev;
    const plf_keypoint *keys;
    const float *uright;
    const uint8_t *desc;
    float min_x, min_y, max_x, max_y, inv_w, inv_h;
    const float *scale_factors;
    int nlevels;
    const int *cell_start;
    const int *cell_idx;
    const float4 *cell_kp;
};
struct MapDev { int m; const float *proj_x, *proj_y, *proj_xr; const int *level; const float *view_cos; const uint8_t *in_view, *desc, *obs_positive; };
struct RelocDev { int on; const float *min_dist, *max_dist; float log_scale; int orb_dist; };
struct LastDev { int n; const uint8_t *has_mp, *outlier; const float *xw; const plf_keypoint *keys; const uint8_t *mp_desc; const uint8_t *obs_positive; };
struct BowDev { int n_kf, n_f; const uint8_t *kf_desc, *f_desc; const float *kf_angle, *f_angle; const uint8_t *kf_has_mp, *f_has_mp; int kf_nodes, f_nodes;
                const uint32_t *kf_node_id, *f_node_id; const int *kf_node_start, *f_node_start; const int *kf_feat, *f_feat; };
struct Pts3Dev { int m; const float *xw, *normal, *min_dist, *max_dist; const uint8_t *desc, *valid; };
struct ProjKf { float R[9], t[3], R2[9], t2[3], Ow[3]; float fx, fy, cx, cy, bf, log_scale; float inv_sigma2[16]; int two_stage, view_test, chi2, accept; };
struct LineFrameDev { int n; const int *n_dev; const plf_keyline *lines; const uint8_t *desc; const float *scale_factors; };
struct MapLineDev { int m; const float *x1, *y1, *x2, *y2; const int *level; const float *view_cos; const uint8_t *in_view; const uint8_t *desc; };

__global__ void k_build_grid(const FrameDev *, int *, int *, int *, int);
__global__ void k_mp_candidates(const FrameDev *, MapDev, float, const int *, int, uint8_t *, uint32_t *, int2 *, int, int *, int *);
__global__ void k_mp_rounds(const FrameDev *, MapDev, float, int *, int, int *, const uint8_t *, int, const uint32_t *, const int2 *, int, const int *, unsigned long long *, int);
struct TriDev { const plf_keypoint *keys1, *keys2; const float *uright1, *uright2, *scale2, *sigma2_2; float F[9]; float ex, ey; int only_stereo; };
__global__ void k_match_bow(const BowDev *, float, int, int, int *, int, int *, int *, int *, TriDev);
__global__ void k_match_project_points_slow(const FrameDev *, MapDev, float, float, int *, int, int *, uint8_t *, int, const int *);
__global__ void k_match_lastframe(const FrameDev *, LastDev, const plf_pose_pair *, RelocDev, float, int, int, int *, int, int *, uint8_t *, float4 *, int, int, const int *);
__global__ void k_lf_candidates(const FrameDev *, LastDev, const plf_pose_pair *, float, int, const int *, int, uint8_t *, uint32_t *, int2 *, int, int, int *, int *);
__global__ void k_lf_rounds(const FrameDev *, LastDev, int, int *, int, int *, const uint8_t *, int, int, const uint32_t *, const int2 *, int, int, const int *);
__global__ void k_project_kf(FrameDev, Pts3Dev, ProjKf, float, int *, int *, int *);
__global__ void k_sim3_agree(const int *, int, const int *, int, int *, int *);
__global__ void k_project_kf_greedy(FrameDev, Pts3Dev, ProjKf, float, int *, int *, uint8_t *, float4 *, int);
__global__ void k_knn2(const uint8_t *, int, const uint8_t *, int, int *, int *);
__global__ void k_knn2_batch(const uint8_t *, int, const LineFrameDev *, int *, int *, int);
__global__ void k_knn2_to_dmatch(const int *, const int *, int, plf_dmatch *);
__global__ void k_line_mad(const int *, int, int, double *);
struct LineTriDev { const uint8_t *has_ml1, *has_ml2, *stereo1, *stereo2; int only_stereo; };
__global__ void k_lines_lastframe(const int *, const int *, int, const uint8_t *, int *, int *, int, int, int, const LineFrameDev *, double, LineTriDev);
__global__ void k_lines_fuse_pick(const int *, const int *, const uint8_t *, int, int *, int *);
__global__ void k_match_project_lines(const LineFrameDev *, MapLineDev, float, float, int *, int, int *, uint8_t *, int);
__global__ void k_match_project_lines_g(const LineFrameDev *, MapLineDev, float, float, int *, int, int *, uint8_t *, int);
__global__ void k_match_project_lines_w(const LineFrameDev *, MapLineDev, float, float, int *, int, int *, uint8_t *, int);
__global__ void k_hamming_matrix(const uint8_t *, int, const uint8_t *, int, int *);

struct plf_matcher {
    int device, max_kp, max_mp, max_lines, max_batch;
    hipStream_t stream;
    FrameDev *d_frames;
    LineFrameDev *d_lframes;
    int *d_cell_start, *d_cell_idx, *d_cell_of, *d_knn_idx, *d_knn_dist;
    float4 *d_cell_kp;
    BowDev *d_bow;       // pair table of plf_match_bow
    int *d_bow_fnode;    // scratch: frame feature -> first common node
    int *d_bow_used;     // scratch: vbMatched2 of the (KeyFrame, KeyFrame) overload
    uint8_t *d_done;
    float4 *d_proj;
    plf_dmatch *d_dm;
    double *d_mad;
    uint32_t *d_cand;        // cached candidate lists of the projection matcher
    unsigned long long *d_top2;   // [frame][max_mappoints]: the two best free candidates of every unfinished map point in the current round (k_mp_rounds)
    int *d_cand_off, *d_overflow;
    int cand_cap;
    FrameDev *h_frames;      // host copy of the frame table last uploaded (skips the upload + sync when unchanged)
    LineFrameDev *h_lframes;
    int h_nframes, h_nlframes;
    // per-frame poses of the last-frame / keyframe searches: a ring of (pinned staging, device table, "kernels done" event) slots, so that a call whose
    // poses changed -- every call of a tracking loop -- uploads them asynchronously on the caller's stream and never blocks the host (the
    // single device table of round 2 needed two hipStreamSynchronize per change)
    struct PoseSlot { plf_pose_pair *d, *h; hipEvent_t ev; bool used; } pose_ring[4];
    int pose_next;
    hipStream_t last_stream;   // stream of the most recent call (matcher_stream)
    bool last_stream_set;
    PlfStreamOrder order;
};

// Handle-owned scratch (frame tables, cell lists, candidate pools) is ordered by the stream the work was enqueued on.  A call on a DIFFERENT stream
// than the previous one first waits for that stream, so successive calls on one handle may use any streams (include/plf.h, "Streams").
static int matcher_stream(plf_matcher *h, void *stream, hipStream_t *out)
{
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    plf_order_begin(h->order, s);
    h->last_stream = s; h->last_stream_set = true;
    *out = s;
    return PLF_OK;
}

static void matcher_free(plf_matcher *h)
{
    void *ptrs[] = {h->d_frames, h->d_lframes, h->d_cell_start, h->d_cell_idx, h->d_cell_of, h->d_knn_idx, h->d_knn_dist, h->d_done, h->d_proj, h->d_dm, h->d_mad, h->d_cand, h->d_top2, h->d_cand_off, h->d_overflow, h->d_cell_kp, h->d_bow, h->d_bow_fnode, h->d_bow_used};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &ps : h->pose_ring) { if (ps.d) (void)hipFree(ps.d); if (ps.h) (void)hipHostFree(ps.h); if (ps.ev) (void)hipEventDestroy(ps.ev); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    plf_order_free(h->order);
    free(h->h_frames); free(h->h_lframes);
}

extern "C" int plf_matcher_create(int32_t device, int32_t max_keypoints, int32_t max_mappoints, int32_t max_lines, int32_t max_batch,
                                  plf_matcher **out)
{
    if (!out || max_keypoints < 1 || max_mappoints < 1 || max_lines < 1 || max_batch < 1) return PLF_E_BADARG;
    *out = nullptr;
    if (max_keypoints > 18000 || max_lines > 18000) return PLF_E_BADARG;  // claim/owner arrays live in LDS
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[plf] no HIP device available: the matchers have no CPU path\n");
        return PLF_E_HIP;
    }
    if (device < 0 || device >= ndev) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(device));
    plf_matcher *h = (plf_matcher *)calloc(1, sizeof(plf_matcher));
    if (!h) return PLF_E_NOMEM;
    h->device = device; h->max_kp = max_keypoints; h->max_mp = max_mappoints; h->max_lines = max_lines; h->max_batch = max_batch;
    const size_t B = (size_t)max_batch;
    const size_t items = (size_t)(max_mappoints > max_keypoints ? max_mappoints : max_keypoints);
#define ALLOC(ptr, bytes)                                                             \
    do {                                                                              \
        if (hipMalloc((void **)&(ptr), (bytes) > 0 ? (bytes) : 256) != hipSuccess) { \
            matcher_free(h); free(h); return PLF_E_NOMEM;                             \
        }                                                                             \
    } while (0)
    ALLOC(h->d_frames, B * sizeof(FrameDev));
    ALLOC(h->d_lframes, B * sizeof(LineFrameDev));
    ALLOC(h->d_cell_start, B * (GRID_CELLS + 1) * sizeof(int));
    ALLOC(h->d_cell_idx, B * (size_t)max_keypoints * sizeof(int));
    ALLOC(h->d_cell_of, B * (size_t)max_keypoints * sizeof(int));
    ALLOC(h->d_cell_kp, B * (size_t)max_keypoints * sizeof(float4));
    ALLOC(h->d_bow, B * sizeof(BowDev));
    ALLOC(h->d_bow_fnode, B * (size_t)max_keypoints * sizeof(int));
    ALLOC(h->d_bow_used, B * (size_t)max_keypoints * sizeof(int));
    ALLOC(h->d_done, B * items);
    {   // one row per map point (greedy Sim3 search, one frame) or max_keypoints entries per frame of a batched last-frame search
        const size_t one = (size_t)(max_keypoints > max_mappoints ? max_keypoints : max_mappoints), batch = B * (size_t)max_keypoints;
        ALLOC(h->d_proj, (one > batch ? one : batch) * sizeof(float4));
    }
    ALLOC(h->d_knn_idx, B * 2 * (size_t)max_lines * sizeof(int));
    ALLOC(h->d_knn_dist, B * 2 * (size_t)max_lines * sizeof(int));
    for (auto &ps : h->pose_ring) {
        ALLOC(ps.d, B * sizeof(plf_pose_pair));
        if (hipHostMalloc((void **)&ps.h, B * sizeof(plf_pose_pair), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); matcher_free(h); free(h); return PLF_E_NOMEM; }
    }
    ALLOC(h->d_dm, 2 * (size_t)max_lines * sizeof(plf_dmatch));
    ALLOC(h->d_mad, 2 * sizeof(double));
    // average of 64 cached candidates per map point; denser frames use the fallback kernel (PLF_MATCH_CAND_AVG: test hook)
    const char *avg_env = getenv("PLF_MATCH_CAND_AVG");
    const int cand_avg = (avg_env && atoi(avg_env) > 0) ? atoi(avg_env) : 64;
    h->cand_cap = cand_avg * max_mappoints;
    ALLOC(h->d_cand, B * (size_t)h->cand_cap * sizeof(uint32_t));
    ALLOC(h->d_cand_off, B * (size_t)max_mappoints * sizeof(int2));   // (start, count) of every map point's span in the pool
    ALLOC(h->d_top2, B * (size_t)max_mappoints * sizeof(unsigned long long));
    ALLOC(h->d_overflow, 2 * B * sizeof(int));                         // [0, B) overflow flags, [B, 2B) pool fill
#undef ALLOC
    h->h_frames = (FrameDev *)calloc(B, sizeof(FrameDev));
    h->h_lframes = (LineFrameDev *)calloc(B, sizeof(LineFrameDev));
    if (!h->h_frames || !h->h_lframes) { matcher_free(h); free(h); return PLF_E_NOMEM; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { matcher_free(h); free(h); return PLF_E_HIP; }
    (void)hipFuncSetAttribute((const void *)k_mp_rounds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_match_project_points_slow, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_match_lastframe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lf_rounds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_match_project_lines, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_match_project_lines_g, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_match_project_lines_w, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();  // the attribute call is advisory; never leave a sticky error behind for other HIP users
    *out = h;
    return PLF_OK;
}

extern "C" void plf_matcher_destroy(plf_matcher *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    matcher_free(h);
    free(h);
}

static int upload_frames(plf_matcher *h, const std::vector<FrameDev> &fd, hipStream_t s);
static int upload_lframes(plf_matcher *h, const std::vector<LineFrameDev> &fd, hipStream_t s);

static FrameDev make_frame(const plf_matcher *h, const plf_frame_view &v, int f)
{
    FrameDev d;
    memset(&d, 0, sizeof(d));  // padding bytes take part in the memcmp cache test
    d.n = v.n; d.n_dev = v.n_device; d.keys = v.keys_un; d.uright = v.uright; d.desc = v.desc;
    d.min_x = v.min_x; d.min_y = v.min_y; d.max_x = v.max_x; d.max_y = v.max_y;
    // mfGridElementWidthInv = float(FRAME_GRID_COLS) / float(mnMaxX - mnMinX)   (Frame ctor, so@0xfa27e)
    d.inv_w = (float)GRID_COLS / (v.max_x - v.min_x);
    d.inv_h = (float)GRID_ROWS / (v.max_y - v.min_y);
    d.scale_factors = v.scale_factors; d.nlevels = v.nlevels;
    d.cell_start = h->d_cell_start + (size_t)f * (GRID_CELLS + 1);
    d.cell_idx = h->d_cell_idx + (size_t)f * h->max_kp;
    d.cell_kp = h->d_cell_kp + (size_t)f * h->max_kp;
    return d;
}

extern "C" int plf_match_project_points(plf_matcher *h, const plf_frame_view *frames, int32_t n_frames, const plf_mappoint_view *mp, float th,
                                        float nnratio, int32_t *match_of_kp, int32_t kp_stride, int32_t *nmatches, void *stream)
{
    if (!h || !frames || !mp || !match_of_kp || !nmatches || n_frames < 1 || n_frames > h->max_batch || mp->m < 0 || mp->m > h->max_mp)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    std::vector<FrameDev> fd(n_frames);
    int maxn = 1;
    for (int f = 0; f < n_frames; f++) {
        if (frames[f].n < 0 || frames[f].n > h->max_kp || frames[f].n > kp_stride || !(frames[f].max_x > frames[f].min_x) ||
            !(frames[f].max_y > frames[f].min_y))
            return PLF_E_BADARG;
        fd[f] = make_frame(h, frames[f], f);
        if (frames[f].n > maxn) maxn = frames[f].n;
    }
    { const int urc = upload_frames(h, fd, s); if (urc != PLF_OK) return urc; }
    hipLaunchKernelGGL(k_build_grid, dim3(n_frames), dim3(256), 0, s, h->d_frames, h->d_cell_start, h->d_cell_idx, h->d_cell_of, h->max_kp);
    MapDev M;
    M.m = mp->m; M.proj_x = mp->proj_x; M.proj_y = mp->proj_y; M.proj_xr = mp->proj_xr; M.level = mp->level; M.view_cos = mp->view_cos;
    M.in_view = mp->in_view; M.desc = mp->desc; M.obs_positive = mp->obs_positive;
    const int kp_cap = (maxn + 63) & ~63;
    if (maxn > 65535) return PLF_E_BADARG;  // candidate cache packs key point indices in 16 bits
    // fast path needs 16-bit map point indices and its lists in LDS; otherwise every frame takes the fallback kernel
    const size_t lds_fast = (size_t)kp_cap * 8 + (size_t)((M.m + 1) & ~1) * sizeof(uint16_t);
    const bool fast = M.m <= 65535 && lds_fast <= 150 * 1024;
    PLF_HIP_TRY(hipMemsetAsync(h->d_overflow, fast ? 0 : 1, 2 * (size_t)h->max_batch * sizeof(int), s));
    if (fast && M.m > 0) {
        hipLaunchKernelGGL(k_mp_candidates, dim3((M.m + 255) / 256, n_frames), dim3(256), 0, s, h->d_frames, M, th, match_of_kp, kp_stride,
                           h->d_done, h->d_cand, (int2 *)h->d_cand_off, h->cand_cap, h->d_overflow, h->d_overflow + h->max_batch);
        hipLaunchKernelGGL(k_mp_rounds, dim3(n_frames), dim3(256), lds_fast, s, h->d_frames, M, nnratio, match_of_kp, kp_stride, nmatches,
                           h->d_done, kp_cap, h->d_cand, (const int2 *)h->d_cand_off, h->cand_cap, h->d_overflow, h->d_top2, h->max_mp);
    }
    hipLaunchKernelGGL(k_match_project_points_slow, dim3(n_frames), dim3(256), (size_t)kp_cap * 8, s, h->d_frames, M, th, nnratio, match_of_kp,
                       kp_stride, nmatches, h->d_done, kp_cap, h->d_overflow);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

static int match_bow_impl(plf_matcher *h, const plf_bow_view *pairs, int32_t n_pairs, float nnratio, int32_t check_orientation, int kfkf,
                          int32_t *match, int32_t stride, int32_t *nmatches, void *stream)
{
    if (!h || !pairs || !match || !nmatches || n_pairs < 1 || n_pairs > h->max_batch || stride < 1 || stride > h->max_kp) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    std::vector<BowDev> pd(n_pairs);
    for (int i = 0; i < n_pairs; i++) {
        const plf_bow_view &v = pairs[i];
        if (v.n_kf < 0 || v.n_f < 0 || v.n_f > stride || v.kf_nodes < 0 || v.f_nodes < 0 || (kfkf && (v.n_kf > stride || !v.f_has_mp))) return PLF_E_BADARG;
        BowDev d;
        d.n_kf = v.n_kf; d.n_f = v.n_f; d.kf_desc = v.kf_desc; d.f_desc = v.f_desc; d.kf_angle = v.kf_angle; d.f_angle = v.f_angle;
        d.kf_has_mp = v.kf_has_mp; d.f_has_mp = v.f_has_mp; d.kf_nodes = v.kf_nodes; d.f_nodes = v.f_nodes; d.kf_node_id = v.kf_node_id;
        d.f_node_id = v.f_node_id; d.kf_node_start = v.kf_node_start; d.f_node_start = v.f_node_start; d.kf_feat = v.kf_feat; d.f_feat = v.f_feat;
        pd[i] = d;
    }
    PLF_HIP_TRY(hipStreamSynchronize(s));   // a previous launch may still read the pair table
    PLF_HIP_TRY(hipMemcpyAsync(h->d_bow, pd.data(), sizeof(BowDev) * n_pairs, hipMemcpyHostToDevice, s));
    PLF_HIP_TRY(hipStreamSynchronize(s));
    TriDev none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(k_match_bow, dim3(n_pairs), dim3(256), 0, s, h->d_bow, nnratio, check_orientation, kfkf, match, stride, nmatches, h->d_bow_fnode,
                       h->d_bow_used, none);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_bow(plf_matcher *h, const plf_bow_view *pairs, int32_t n_pairs, float nnratio, int32_t check_orientation,
                             int32_t *match_of_f, int32_t stride, int32_t *nmatches, void *stream)
{
    return match_bow_impl(h, pairs, n_pairs, nnratio, check_orientation, 0, match_of_f, stride, nmatches, stream);
}

extern "C" int plf_match_bow_kf(plf_matcher *h, const plf_bow_view *pairs, int32_t n_pairs, float nnratio, int32_t check_orientation,
                                int32_t *match12, int32_t stride, int32_t *nmatches, void *stream)
{
    return match_bow_impl(h, pairs, n_pairs, nnratio, check_orientation, 1, match12, stride, nmatches, stream);
}

extern "C" int plf_match_triangulation(plf_matcher *h, const plf_tri_view *v, const float *F12, const float *Cw1, const plf_kf_pose *pose2,
                                       int32_t only_stereo, int32_t check_orientation, int32_t *match12, int32_t *nmatches, void *stream)
{
    if (!h || !v || !F12 || !Cw1 || !pose2 || !match12 || !nmatches || v->n1 < 0 || v->n2 < 0 || v->n1 > h->max_kp || v->n2 > h->max_kp ||
        v->nodes1 < 0 || v->nodes2 < 0 || !v->has_mp1 || !v->has_mp2 || !v->uright1 || !v->uright2 || !v->keys1 || !v->keys2 || !v->scale_factors2 ||
        !v->level_sigma2_2)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    BowDev d;
    memset(&d, 0, sizeof(d));
    d.n_kf = v->n1; d.n_f = v->n2; d.kf_desc = v->desc1; d.f_desc = v->desc2; d.kf_has_mp = v->has_mp1; d.f_has_mp = v->has_mp2;
    d.kf_nodes = v->nodes1; d.f_nodes = v->nodes2; d.kf_node_id = v->node_id1; d.f_node_id = v->node_id2; d.kf_node_start = v->node_start1;
    d.f_node_start = v->node_start2; d.kf_feat = v->feat1; d.f_feat = v->feat2;
    TriDev T;
    memset(&T, 0, sizeof(T));
    T.keys1 = v->keys1; T.keys2 = v->keys2; T.uright1 = v->uright1; T.uright2 = v->uright2; T.scale2 = v->scale_factors2; T.sigma2_2 = v->level_sigma2_2;
    memcpy(T.F, F12, sizeof(T.F));
    T.only_stereo = only_stereo != 0;
    // epipole of keyframe 1 in keyframe 2 (so@0x86b9c-0x86f95): C2 = R2w*Cw + t2w (gemm small-matrix float path), ex = fmaf(fx*C2x, 1/C2z, cx)
    float C2[3];
    for (int r = 0; r < 3; r++) {
        const float t = pose2->Rcw[r * 3] * Cw1[0] + pose2->Rcw[r * 3 + 1] * Cw1[1] + pose2->Rcw[r * 3 + 2] * Cw1[2];
        C2[r] = (float)((double)t + (double)pose2->tcw[r]);
    }
    const float invz = 1.0f / C2[2];
    T.ex = fmaf(pose2->fx * C2[0], invz, pose2->cx);
    T.ey = fmaf(pose2->fy * C2[1], invz, pose2->cy);
    PLF_HIP_TRY(hipStreamSynchronize(s));   // a previous launch may still read the pair table
    PLF_HIP_TRY(hipMemcpyAsync(h->d_bow, &d, sizeof(BowDev), hipMemcpyHostToDevice, s));
    PLF_HIP_TRY(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_match_bow, dim3(1), dim3(256), 0, s, h->d_bow, 0.f, check_orientation, 2, match12, h->max_kp, nmatches, h->d_bow_fnode, h->d_bow_used, T);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

// the frame table of the point matchers: uploaded only when it changed (the sync protects a launch that may still read the old one)
static int upload_frames(plf_matcher *h, const std::vector<FrameDev> &fd, hipStream_t s)
{
    const int n = (int)fd.size();
    if (h->h_nframes != n || memcmp(h->h_frames, fd.data(), sizeof(FrameDev) * n) != 0) {
        PLF_HIP_TRY(hipStreamSynchronize(s));
        memcpy(h->h_frames, fd.data(), sizeof(FrameDev) * n);
        h->h_nframes = n;
        PLF_HIP_TRY(hipMemcpyAsync(h->d_frames, h->h_frames, sizeof(FrameDev) * n, hipMemcpyHostToDevice, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
    }
    return PLF_OK;
}

static int upload_lframes(plf_matcher *h, const std::vector<LineFrameDev> &fd, hipStream_t s)
{
    const int n = (int)fd.size();
    if (h->h_nlframes != n || memcmp(h->h_lframes, fd.data(), sizeof(LineFrameDev) * n) != 0) {
        PLF_HIP_TRY(hipStreamSynchronize(s));
        memcpy(h->h_lframes, fd.data(), sizeof(LineFrameDev) * n);
        h->h_nlframes = n;
        PLF_HIP_TRY(hipMemcpyAsync(h->d_lframes, h->h_lframes, sizeof(LineFrameDev) * n, hipMemcpyHostToDevice, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
    }
    return PLF_OK;
}

// n_frames current frames against ONE last frame / keyframe, every frame with its own pose (poses: host array, pose_step 0 = one shared pose)
static int match_lastframe_impl(plf_matcher *h, const plf_frame_view *frames, int32_t n_frames, const plf_lastframe_view *last, const plf_pose_pair *poses,
                                int pose_step, RelocDev RL, float th, int32_t mono, int32_t check_orientation, int32_t *match_of_kp, int32_t kp_stride,
                                int32_t *nmatches, void *stream)
{
    if (!h || !frames || !last || !poses || !match_of_kp || !nmatches || n_frames < 1 || n_frames > h->max_batch || last->n < 0 || last->n > h->max_kp ||
        !last->has_mappoint || (!RL.on && !last->outlier))
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    std::vector<FrameDev> fd(n_frames);
    int maxn = 1;
    for (int f = 0; f < n_frames; f++) {
        const plf_frame_view &v = frames[f];
        if (v.n < 0 || v.n > h->max_kp || (n_frames > 1 && v.n > kp_stride) || !(v.max_x > v.min_x) || !(v.max_y > v.min_y)) return PLF_E_BADARG;
        fd[f] = make_frame(h, v, f);
        if (v.n > maxn) maxn = v.n;
    }
    int rc = upload_frames(h, fd, s);
    if (rc != PLF_OK) return rc;
    // poses: next slot of the ring (its previous user finished long ago in steady state: the wait is a formality), staged in pinned memory, uploaded in
    // stream order -- the host does not wait
    plf_matcher::PoseSlot &ps = h->pose_ring[h->pose_next];
    h->pose_next = (h->pose_next + 1) % 4;
    if (ps.used) PLF_HIP_TRY(hipEventSynchronize(ps.ev));
    for (int f = 0; f < n_frames; f++) ps.h[f] = poses[(size_t)f * pose_step];
    PLF_HIP_TRY(hipMemcpyAsync(ps.d, ps.h, sizeof(plf_pose_pair) * n_frames, hipMemcpyHostToDevice, s));
    const plf_pose_pair *d_poses = ps.d;
    hipLaunchKernelGGL(k_build_grid, dim3(n_frames), dim3(256), 0, s, h->d_frames, h->d_cell_start, h->d_cell_idx, h->d_cell_of, h->max_kp);
    LastDev L;
    L.n = last->n; L.has_mp = last->has_mappoint; L.outlier = last->outlier; L.xw = last->world_pos; L.keys = last->keys; L.mp_desc = last->mp_desc;
    L.obs_positive = RL.on ? nullptr : last->obs_positive;   // (the relocalisation overload tests the pointer only)
    const int kp_cap = (maxn + 63) & ~63;
    // batches of the motion-model overload: cached candidate lists + conflict-free rounds (k_lf_*); key point / last-frame indices must fit 16 bits,
    // the span table one row per last-frame point, the lists the LDS.  Frames that overflow the pool, and everything else, use k_match_lastframe.
    const int item_cap = (last->n + 1) & ~1;
    const size_t lds_fast = (size_t)kp_cap * 8 + 3 * (size_t)item_cap * sizeof(uint16_t);
    const bool fast = !RL.on && n_frames >= 1 && maxn <= 65535 && last->n <= 65534 && last->n <= h->max_mp && last->n > 0 && lds_fast <= 150 * 1024;
    if (fast) {
        PLF_HIP_TRY(hipMemsetAsync(h->d_overflow, 0, 2 * (size_t)h->max_batch * sizeof(int), s));
        hipLaunchKernelGGL(k_lf_candidates, dim3((last->n + 255) / 256, n_frames), dim3(256), 0, s, h->d_frames, L, d_poses, th, mono, match_of_kp, kp_stride,
                           h->d_done, h->d_cand, (int2 *)h->d_cand_off, h->cand_cap, h->max_mp, h->d_overflow, h->d_overflow + h->max_batch);
        hipLaunchKernelGGL(k_lf_rounds, dim3(n_frames), dim3(256), lds_fast, s, h->d_frames, L, check_orientation, match_of_kp, kp_stride, nmatches, h->d_done,
                           kp_cap, item_cap, h->d_cand, (const int2 *)h->d_cand_off, h->cand_cap, h->max_mp, h->d_overflow);
    }
    hipLaunchKernelGGL(k_match_lastframe, dim3(n_frames), dim3(256), (size_t)kp_cap * 8, s, h->d_frames, L, d_poses, RL, th, mono, check_orientation, match_of_kp,
                       kp_stride, nmatches, h->d_done, h->d_proj, kp_cap, h->max_kp, fast ? h->d_overflow : (const int *)nullptr);
    PLF_HIP_TRY(hipEventRecord(ps.ev, s));
    ps.used = true;
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_project_lastframe(plf_matcher *h, const plf_frame_view *cur, const plf_lastframe_view *last, const plf_pose_pair *pose,
                                           float th, int32_t mono, int32_t check_orientation, int32_t *match_of_kp, int32_t *nmatches,
                                           void *stream)
{
    RelocDev RL;
    memset(&RL, 0, sizeof(RL));
    return match_lastframe_impl(h, cur, 1, last, pose, 0, RL, th, mono, check_orientation, match_of_kp, cur ? cur->n : 0, nmatches, stream);
}

extern "C" int plf_match_project_lastframe_batch(plf_matcher *h, const plf_frame_view *frames, int32_t n_frames, const plf_lastframe_view *last,
                                                 const plf_pose_pair *poses, float th, int32_t mono, int32_t check_orientation, int32_t *match_of_kp,
                                                 int32_t kp_stride, int32_t *nmatches, void *stream)
{
    RelocDev RL;
    memset(&RL, 0, sizeof(RL));
    return match_lastframe_impl(h, frames, n_frames, last, poses, 1, RL, th, mono, check_orientation, match_of_kp, kp_stride, nmatches, stream);
}

extern "C" int plf_match_project_keyframe(plf_matcher *h, const plf_frame_view *cur, const plf_lastframe_view *kf, const float *min_distance,
                                          const float *max_distance, const plf_pose_pair *pose, float log_scale_factor, float th, int32_t orb_dist,
                                          int32_t check_orientation, int32_t *match_of_kp, int32_t *nmatches, void *stream)
{
    if (!min_distance || !max_distance || !(log_scale_factor > 0.f) || !cur || cur->nlevels < 1 || cur->nlevels > 64) return PLF_E_BADARG;
    RelocDev RL;
    RL.on = 1; RL.min_dist = min_distance; RL.max_dist = max_distance; RL.log_scale = log_scale_factor; RL.orb_dist = orb_dist;
    return match_lastframe_impl(h, cur, 1, kf, pose, 0, RL, th, 1, check_orientation, match_of_kp, cur->n, nmatches, stream);
}

// upload the keyframe's view as frame 0 of the handle's table and build its cell CSR
static int stage_keyframe(plf_matcher *h, const plf_frame_view *kf, hipStream_t s, FrameDev *out)
{
    if (!kf || kf->n < 0 || kf->n > h->max_kp || !(kf->max_x > kf->min_x) || !(kf->max_y > kf->min_y) || kf->nlevels < 1 || kf->nlevels > 16) return PLF_E_BADARG;
    FrameDev fd;
    memset(&fd, 0, sizeof(fd));
    fd = make_frame(h, *kf, 0);
    if (h->h_nframes != 1 || memcmp(h->h_frames, &fd, sizeof(FrameDev)) != 0) {
        PLF_HIP_TRY(hipStreamSynchronize(s));
        memcpy(h->h_frames, &fd, sizeof(FrameDev));
        h->h_nframes = 1;
        PLF_HIP_TRY(hipMemcpyAsync(h->d_frames, h->h_frames, sizeof(FrameDev), hipMemcpyHostToDevice, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_build_grid, dim3(1), dim3(256), 0, s, h->d_frames, h->d_cell_start, h->d_cell_idx, h->d_cell_of, h->max_kp);
    *out = fd;
    return PLF_OK;
}

static Pts3Dev make_points(const plf_points3d_view *p)
{
    Pts3Dev P;
    P.m = p->m; P.xw = p->world_pos; P.normal = p->normal; P.min_dist = p->min_distance; P.max_dist = p->max_distance; P.desc = p->desc; P.valid = p->valid;
    return P;
}

static void fill_camera(ProjKf *C, const plf_kf_pose *pose, int nlevels)
{
    memset(C, 0, sizeof(*C));
    memcpy(C->R, pose->Rcw, sizeof(C->R)); memcpy(C->t, pose->tcw, sizeof(C->t)); memcpy(C->Ow, pose->Ow, sizeof(C->Ow));
    C->fx = pose->fx; C->fy = pose->fy; C->cx = pose->cx; C->cy = pose->cy; C->bf = pose->bf; C->log_scale = pose->log_scale_factor;
    for (int l = 0; l < nlevels && l < 16; l++) C->inv_sigma2[l] = pose->inv_level_sigma2[l];
}

extern "C" int plf_match_assign_grid(plf_matcher *h, const plf_frame_view *frame, int32_t *cell_start, int32_t *cell_idx, void *stream)
{
    if (!h || !frame || !cell_start || !cell_idx) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    FrameDev fd;
    plf_frame_view v = *frame;
    if (v.nlevels < 1) v.nlevels = 1;   // (the grid does not read the scale pyramid)
    const int st = stage_keyframe(h, &v, s, &fd);
    if (st != PLF_OK) return st;
    PLF_HIP_TRY(hipMemcpyAsync(cell_start, h->d_cell_start, sizeof(int) * (GRID_CELLS + 1), hipMemcpyDeviceToHost, s));
    PLF_HIP_TRY(hipStreamSynchronize(s));
    const int total = cell_start[GRID_CELLS];
    if (total > 0) PLF_HIP_TRY(hipMemcpyAsync(cell_idx, h->d_cell_idx, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost, s));
    PLF_HIP_TRY(hipStreamSynchronize(s));
    return PLF_OK;
}

extern "C" int plf_match_fuse(plf_matcher *h, const plf_frame_view *kf, const plf_kf_pose *pose, const plf_points3d_view *pts, float th,
                              int32_t *best_idx, int32_t *nfused, void *stream)
{
    if (!h || !kf || !pose || !pts || !best_idx || !nfused || pts->m < 0 || pts->m > h->max_mp || !(pose->log_scale_factor > 0.f)) return PLF_E_BADARG;
    if (pts->m > 0 && (!pts->world_pos || !pts->normal || !pts->min_distance || !pts->max_distance || !pts->desc || !pts->valid)) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    FrameDev fd;
    const int st = stage_keyframe(h, kf, s, &fd);
    if (st != PLF_OK) return st;
    ProjKf C;
    fill_camera(&C, pose, kf->nlevels);
    C.view_test = 1; C.chi2 = 1; C.accept = 50;   // ORBmatcher::TH_LOW
    PLF_HIP_TRY(hipMemsetAsync(nfused, 0, sizeof(int), s));
    if (pts->m > 0)
        hipLaunchKernelGGL(k_project_kf, dim3((pts->m + 255) / 256), dim3(256), 0, s, fd, make_points(pts), C, th, best_idx, (int *)nullptr, nfused);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

// Sim3 decomposition at the top of the two Scw overloads (so@0x7bb20, so@0x880f0): scw = float(sqrt(row0 . row0)) with a double dot product,
// Rcw = sRcw / scw and tcw = Scw(0:3, 3) / scw through cv::operator/(Mat, double) (a scale expression assigned by Mat::convertTo, which
// for CV_32F multiplies by float(1.0 / double(scw)) -- OpenCV 3.3 convert.cpp), Ow = -Rcw.t() * tcw through cv::gemm's transposed path.
static void sim3_decompose(const float *Scw, float *Rcw, float *tcw, float *Ow)
{
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)sqrt(dot);
    const float alpha = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Scw[r * 4 + c] * alpha;
        tcw[r] = Scw[r * 4 + 3] * alpha;
    }
    for (int i = 0; i < 3; i++) Ow[i] = (float)(((double)Rcw[i] * tcw[0] + (double)Rcw[3 + i] * tcw[1] + (double)Rcw[6 + i] * tcw[2]) * -1.0);
}

static int sim3_common(plf_matcher *h, const plf_frame_view *kf, const float *Scw, const plf_kf_pose *intr, const plf_points3d_view *pts, hipStream_t s,
                       FrameDev *fd, ProjKf *C)
{
    if (!h || !kf || !Scw || !intr || !pts || pts->m < 0 || pts->m > h->max_mp || !(intr->log_scale_factor > 0.f)) return PLF_E_BADARG;
    if (pts->m > 0 && (!pts->world_pos || !pts->normal || !pts->min_distance || !pts->max_distance || !pts->desc || !pts->valid)) return PLF_E_BADARG;
    const int st = stage_keyframe(h, kf, s, fd);
    if (st != PLF_OK) return st;
    memset(C, 0, sizeof(*C));
    sim3_decompose(Scw, C->R, C->t, C->Ow);
    C->fx = intr->fx; C->fy = intr->fy; C->cx = intr->cx; C->cy = intr->cy; C->bf = intr->bf; C->log_scale = intr->log_scale_factor;
    C->view_test = 1; C->chi2 = 0; C->accept = 50;   // ORBmatcher::TH_LOW
    return PLF_OK;
}

extern "C" int plf_match_fuse_sim3(plf_matcher *h, const plf_frame_view *kf, const float *Scw, const plf_kf_pose *intr, const plf_points3d_view *pts,
                                   float th, int32_t *best_idx, int32_t *nfused, void *stream)
{
    if (!h || !best_idx || !nfused) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    FrameDev fd; ProjKf C;
    const int st = sim3_common(h, kf, Scw, intr, pts, s, &fd, &C);
    if (st != PLF_OK) return st;
    PLF_HIP_TRY(hipMemsetAsync(nfused, 0, sizeof(int), s));
    if (pts->m > 0)
        hipLaunchKernelGGL(k_project_kf, dim3((pts->m + 255) / 256), dim3(256), 0, s, fd, make_points(pts), C, th, best_idx, (int *)nullptr, nfused);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_project_sim3(plf_matcher *h, const plf_frame_view *kf, const float *Scw, const plf_kf_pose *intr, const plf_points3d_view *pts,
                                      int32_t th, int32_t *match_of_kp, int32_t *nmatches, void *stream)
{
    if (!h || !match_of_kp || !nmatches) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    FrameDev fd; ProjKf C;
    const int st = sim3_common(h, kf, Scw, intr, pts, s, &fd, &C);
    if (st != PLF_OK) return st;
    const int kp_cap = ((kf->n > 0 ? kf->n : 1) + 63) & ~63;
    hipLaunchKernelGGL(k_project_kf_greedy, dim3(1), dim3(256), (size_t)kp_cap * 8, s, fd, make_points(pts), C, (float)th, match_of_kp, nmatches, h->d_done,
                       h->d_proj, kp_cap);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_sim3(plf_matcher *h, const plf_frame_view *kf1, const plf_frame_view *kf2, const plf_kf_pose *pose1, const plf_kf_pose *pose2,
                              float s12, const float *R12, const float *t12, float th, const plf_points3d_view *pts1, const plf_points3d_view *pts2,
                              int32_t *match12, int32_t *nfound, void *stream)
{
    if (!h || !kf1 || !kf2 || !pose1 || !pose2 || !R12 || !t12 || !pts1 || !pts2 || !match12 || !nfound || pts1->m != kf1->n || pts2->m != kf2->n ||
        pts1->m > h->max_mp || pts2->m > h->max_mp || kf1->n < 0 || kf2->n < 0 || kf1->n > h->max_kp || kf2->n > h->max_kp /* vn1 / vn2 hold max_kp ints */ ||
        !(s12 > 0.f) || !(pose1->log_scale_factor > 0.f) || !(pose2->log_scale_factor > 0.f))
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    // sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t() (scale expressions, float work type), t21 = -sR21 * t12 (gemm, alpha = -1)
    float sR12[9], sR21[9], t21[3];
    const float a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sR12[r * 3 + c] = R12[r * 3 + c] * s12; sR21[r * 3 + c] = R12[c * 3 + r] * a21; }
    for (int r = 0; r < 3; r++) {
        const float t = sR21[r * 3] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
        t21[r] = (float)((double)t * -1.0);
    }
    int *vn1 = h->d_bow_fnode, *vn2 = h->d_bow_used;   // max_kp ints each
    PLF_HIP_TRY(hipMemsetAsync(nfound, 0, sizeof(int), s));
    for (int dir = 0; dir < 2; dir++) {
        const plf_frame_view *target = dir == 0 ? kf2 : kf1;
        const plf_kf_pose *src = dir == 0 ? pose1 : pose2, *tgt = dir == 0 ? pose2 : pose1;
        const plf_points3d_view *pts = dir == 0 ? pts1 : pts2;
        FrameDev fd;
        const int st = stage_keyframe(h, target, s, &fd);
        if (st != PLF_OK) return st;
        ProjKf C;
        memset(&C, 0, sizeof(C));
        memcpy(C.R, src->Rcw, sizeof(C.R)); memcpy(C.t, src->tcw, sizeof(C.t));
        memcpy(C.R2, dir == 0 ? sR21 : sR12, sizeof(C.R2)); memcpy(C.t2, dir == 0 ? t21 : t12, sizeof(C.t2));
        // the binary loads fx, fy, cx, cy from pKF1 for BOTH directions (so@0x84acc, so@0x8586c)
        C.fx = pose1->fx; C.fy = pose1->fy; C.cx = pose1->cx; C.cy = pose1->cy; C.log_scale = tgt->log_scale_factor;
        C.two_stage = 1; C.view_test = 0; C.chi2 = 0; C.accept = 100;   // ORBmatcher::TH_HIGH
        if (pts->m > 0)
            hipLaunchKernelGGL(k_project_kf, dim3((pts->m + 255) / 256), dim3(256), 0, s, fd, make_points(pts), C, th, dir == 0 ? vn1 : vn2, (int *)nullptr,
                               (int *)nullptr);
    }
    if (pts1->m > 0)
        hipLaunchKernelGGL(k_sim3_agree, dim3((pts1->m + 255) / 256), dim3(256), 0, s, vn1, pts1->m, vn2, pts2->m, match12, nfound);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_lines_knn(plf_matcher *h, const uint8_t *query, int32_t nq, const uint8_t *train, int32_t nt, plf_dmatch *out,
                                   int32_t mem, void *stream)
{
    if (!h || !query || !train || !out || nq < 1 || nt < 1 || nq > h->max_lines) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    hipLaunchKernelGGL(k_knn2, dim3((nq + 127) / 128), dim3(128), 0, s, query, nq, train, nt, h->d_knn_idx, h->d_knn_dist);
    plf_dmatch *d_out = mem == PLF_MEM_DEVICE ? out : h->d_dm;
    hipLaunchKernelGGL(k_knn2_to_dmatch, dim3((2 * nq + 127) / 128), dim3(128), 0, s, h->d_knn_idx, h->d_knn_dist, nq, d_out);
    PLF_HIP_TRY(hipGetLastError());
    if (mem != PLF_MEM_DEVICE) {
        PLF_HIP_TRY(hipMemcpyAsync(out, h->d_dm, sizeof(plf_dmatch) * 2 * nq, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
    }
    return PLF_OK;
}

extern "C" double plf_line_segment_overlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj)
{
    const double sln = spl_obs < epl_obs ? spl_obs : epl_obs, eln = spl_obs < epl_obs ? epl_obs : spl_obs;
    const double spn = spl_proj < epl_proj ? spl_proj : epl_proj, epn = spl_proj < epl_proj ? epl_proj : spl_proj;
    const double length = eln - spn;
    double overlap;
    if (epn < sln || spn > eln) overlap = 0.0;
    else if (epn > eln && spn < sln) overlap = eln - sln;
    else overlap = (eln < epn ? eln : epn) - (sln > spn ? sln : spn);
    return length > (double)0.01f ? overlap / length : 0.0;
}

extern "C" int plf_line_descriptor_mad(plf_matcher *h, const uint8_t *ldesc1, int32_t n1, const uint8_t *ldesc2, int32_t n2, plf_dmatch *knn,
                                       double *mad, int32_t mem, void *stream)
{
    if (!h || !ldesc1 || !ldesc2 || !mad || n1 < 1 || n2 < 2 || n1 > h->max_lines) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    hipLaunchKernelGGL(k_knn2, dim3((n1 + 127) / 128), dim3(128), 0, s, ldesc1, n1, ldesc2, n2, h->d_knn_idx, h->d_knn_dist);
    int P2 = 1;
    while (P2 < n1) P2 <<= 1;
    double *d_mad = mem == PLF_MEM_DEVICE ? mad : h->d_mad;
    hipLaunchKernelGGL(k_line_mad, dim3(1), dim3(256), (size_t)P2 * sizeof(float), s, h->d_knn_dist, n1, P2, d_mad);
    plf_dmatch *d_out = mem == PLF_MEM_DEVICE ? knn : h->d_dm;
    if (knn) hipLaunchKernelGGL(k_knn2_to_dmatch, dim3((2 * n1 + 127) / 128), dim3(128), 0, s, h->d_knn_idx, h->d_knn_dist, n1, d_out);
    PLF_HIP_TRY(hipGetLastError());
    if (mem != PLF_MEM_DEVICE) {
        if (knn) PLF_HIP_TRY(hipMemcpyAsync(knn, h->d_dm, sizeof(plf_dmatch) * 2 * n1, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipMemcpyAsync(mad, h->d_mad, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
    }
    return PLF_OK;
}

extern "C" int plf_match_lines_lastframe(plf_matcher *h, const uint8_t *last_desc, int32_t nlast, const uint8_t *cur_desc, int32_t ncur,
                                         const uint8_t *last_has_mapline, int32_t *match_of_line, int32_t *nmatches, void *stream)
{
    if (!h || !last_desc || !cur_desc || !last_has_mapline || !match_of_line || !nmatches || nlast > h->max_lines || ncur > h->max_lines)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    if (nlast <= 0 || ncur < 2) { PLF_HIP_TRY(hipMemsetAsync(nmatches, 0, sizeof(int), s)); return PLF_OK; }
    hipLaunchKernelGGL(k_knn2, dim3((nlast + 127) / 128), dim3(128), 0, s, last_desc, nlast, cur_desc, ncur, h->d_knn_idx, h->d_knn_dist);
    int P2 = 1;
    while (P2 < nlast) P2 <<= 1;
    hipLaunchKernelGGL(k_lines_lastframe, dim3(1), dim3(256), (size_t)P2 * sizeof(float), s, h->d_knn_idx, h->d_knn_dist, nlast,
                       last_has_mapline, match_of_line, nmatches, P2, 0, 0, (const LineFrameDev *)nullptr, 0.5, LineTriDev{nullptr, nullptr, nullptr, nullptr, 0});
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_lines_lastframe_batch(plf_matcher *h, const uint8_t *last_desc, int32_t nlast, const uint8_t *last_has_mapline,
                                               const plf_lineframe_view *frames, int32_t n_frames, int32_t *match_of_line, int32_t line_stride,
                                               int32_t *nmatches, void *stream)
{
    if (!h || !last_desc || !last_has_mapline || !frames || !match_of_line || !nmatches || nlast > h->max_lines || n_frames < 1 || n_frames > h->max_batch)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    if (nlast <= 0) { PLF_HIP_TRY(hipMemsetAsync(nmatches, 0, sizeof(int) * n_frames, s)); return PLF_OK; }
    std::vector<LineFrameDev> fd(n_frames);
    memset(fd.data(), 0, sizeof(LineFrameDev) * n_frames);
    for (int f = 0; f < n_frames; f++) {
        if (frames[f].n < 0 || frames[f].n > h->max_lines || frames[f].n > line_stride || !frames[f].desc) return PLF_E_BADARG;
        fd[f].n = frames[f].n; fd[f].n_dev = frames[f].n_device; fd[f].lines = frames[f].lines_un; fd[f].desc = frames[f].desc; fd[f].scale_factors = frames[f].scale_factors;
    }
    const int rc = upload_lframes(h, fd, s);
    if (rc != PLF_OK) return rc;
    const int stride = 2 * h->max_lines;
    hipLaunchKernelGGL(k_knn2_batch, dim3((nlast + 127) / 128, n_frames), dim3(128), 0, s, last_desc, nlast, h->d_lframes, h->d_knn_idx, h->d_knn_dist, stride);
    int P2 = 1;
    while (P2 < nlast) P2 <<= 1;
    hipLaunchKernelGGL(k_lines_lastframe, dim3(n_frames), dim3(256), (size_t)P2 * sizeof(float), s, h->d_knn_idx, h->d_knn_dist, nlast, last_has_mapline,
                       match_of_line, nmatches, P2, stride, line_stride, h->d_lframes, 0.5, LineTriDev{nullptr, nullptr, nullptr, nullptr, 0});
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_lines_triangulation(plf_matcher *h, const uint8_t *desc1, int32_t n1, const uint8_t *desc2, int32_t n2, const uint8_t *has_ml1,
                                             const uint8_t *has_ml2, const uint8_t *stereo1, const uint8_t *stereo2, int32_t only_stereo, float mad_factor,
                                             int32_t *match12, int32_t *nmatches, void *stream)
{
    if (!h || !desc1 || !desc2 || !has_ml1 || !has_ml2 || !match12 || !nmatches || n1 < 0 || n2 < 0 || n1 > h->max_lines || n2 > h->max_lines ||
        (only_stereo && (!stereo1 || !stereo2)) || !(mad_factor >= 0.f))
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    if (n1 > 0) PLF_HIP_TRY(hipMemsetAsync(match12, 0xFF, sizeof(int32_t) * (size_t)n1, s));
    if (n1 <= 0 || n2 < 2) { PLF_HIP_TRY(hipMemsetAsync(nmatches, 0, sizeof(int), s)); return PLF_OK; }
    hipLaunchKernelGGL(k_knn2, dim3((n1 + 127) / 128), dim3(128), 0, s, desc1, n1, desc2, n2, h->d_knn_idx, h->d_knn_dist);
    int P2 = 1;
    while (P2 < n1) P2 <<= 1;
    hipLaunchKernelGGL(k_lines_lastframe, dim3(1), dim3(256), (size_t)P2 * sizeof(float), s, h->d_knn_idx, h->d_knn_dist, n1, (const uint8_t *)nullptr, match12,
                       nmatches, P2, 0, 0, (const LineFrameDev *)nullptr, (double)mad_factor, LineTriDev{has_ml1, has_ml2, stereo1, stereo2, only_stereo ? 1 : 0});
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_lines_fuse(plf_matcher *h, const uint8_t *kf_desc, int32_t n_kf, const uint8_t *ml_desc, const uint8_t *valid, int32_t m,
                                    int32_t *best_idx, int32_t *nfused, void *stream)
{
    if (!h || !kf_desc || !ml_desc || !valid || !best_idx || !nfused || n_kf < 0 || m < 0 || n_kf > h->max_lines || m > h->max_lines) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    PLF_HIP_TRY(hipMemsetAsync(nfused, 0, sizeof(int), s));
    if (m == 0) return PLF_OK;
    // brute force over ALL keyframe lines: the 2-NN table of (map lines -> keyframe lines); n_kf == 0 leaves idx = -1
    hipLaunchKernelGGL(k_knn2, dim3((m + 127) / 128), dim3(128), 0, s, ml_desc, m, kf_desc, n_kf, h->d_knn_idx, h->d_knn_dist);
    hipLaunchKernelGGL(k_lines_fuse_pick, dim3((m + 255) / 256), dim3(256), 0, s, h->d_knn_idx, h->d_knn_dist, valid, m, best_idx, nfused);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_match_project_lines(plf_matcher *h, const plf_lineframe_view *frames, int32_t n_frames, const plf_mapline_view *ml, float th,
                                       float nnratio, int32_t *match_of_line, int32_t line_stride, int32_t *nmatches, void *stream)
{
    if (!h || !frames || !ml || !match_of_line || !nmatches || n_frames < 1 || n_frames > h->max_batch || ml->m < 0 || ml->m > h->max_mp)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s;
    { const int src_ = matcher_stream(h, stream, &s); if (src_ != PLF_OK) return src_; }
    PlfOrderGuard order_guard_{h->order, s};
    std::vector<LineFrameDev> fd(n_frames);
    memset(fd.data(), 0, sizeof(LineFrameDev) * n_frames);
    int maxn = 1;
    for (int f = 0; f < n_frames; f++) {
        if (frames[f].n < 0 || frames[f].n > h->max_lines || frames[f].n > line_stride) return PLF_E_BADARG;
        fd[f].n = frames[f].n; fd[f].n_dev = frames[f].n_device; fd[f].lines = frames[f].lines_un; fd[f].desc = frames[f].desc; fd[f].scale_factors = frames[f].scale_factors;
        if (frames[f].n > maxn) maxn = frames[f].n;
    }
    { const int urc = upload_lframes(h, fd, s); if (urc != PLF_OK) return urc; }
    MapLineDev M;
    M.m = ml->m; M.x1 = ml->x1; M.y1 = ml->y1; M.x2 = ml->x2; M.y2 = ml->y2; M.level = ml->level; M.view_cos = ml->view_cos;
    M.in_view = ml->in_view; M.desc = ml->desc;
    const int cap = (maxn + 63) & ~63;
    // the frame's lines staged in LDS while they fit (56 bytes per line: up to 2688 lines in 150 KB), else read from global memory (8 bytes of LDS per line: the
    // 18000 lines plf_matcher_create accepts take 144 KB) -- ADVICE r05
    // (PLF_MATCH_LINES_STAGE_MAX: test hook -- 0 forces the global-memory variant on small frames)
    const char *stage_env = getenv("PLF_MATCH_LINES_STAGE_MAX");
    const int stage_lines = stage_env ? atoi(stage_env) : 150 * 1024 / 56;
    // few frames in flight: one wave per map line, the key lines across the lanes (k_match_project_lines_w; PLF_MATCH_LINES_WAVE_MAX: test hook, 0 = never)
    const char *wave_env = getenv("PLF_MATCH_LINES_WAVE_MAX");
    const int wave_frames = wave_env ? atoi(wave_env) : 64;
    if (cap <= stage_lines && n_frames <= wave_frames && cap <= 65536)
        hipLaunchKernelGGL(k_match_project_lines_w, dim3(n_frames), dim3(1024), (size_t)cap * 56, s, h->d_lframes, M, th, nnratio, match_of_line,
                           line_stride, nmatches, h->d_done, cap);
    else if (cap <= stage_lines)
        hipLaunchKernelGGL(k_match_project_lines, dim3(n_frames), dim3(256), (size_t)cap * 56, s, h->d_lframes, M, th, nnratio, match_of_line,
                           line_stride, nmatches, h->d_done, cap);
    else
        hipLaunchKernelGGL(k_match_project_lines_g, dim3(n_frames), dim3(256), (size_t)cap * 8, s, h->d_lframes, M, th, nnratio, match_of_line,
                           line_stride, nmatches, h->d_done, cap);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_hamming256_matrix(const uint8_t *a, int32_t na, const uint8_t *b, int32_t nb, int32_t *dist, int32_t mem, int32_t device,
                                     void *stream)
{
    if (!a || !b || !dist || na < 1 || nb < 1) return PLF_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PLF_E_HIP;
    PLF_HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *da = a, *db = b;
    int *dd = dist;
    void *ta = nullptr, *tb = nullptr, *td = nullptr;
    if (mem != PLF_MEM_DEVICE) {
        PLF_HIP_TRY(hipMalloc(&ta, (size_t)na * 32)); PLF_HIP_TRY(hipMalloc(&tb, (size_t)nb * 32)); PLF_HIP_TRY(hipMalloc(&td, (size_t)na * nb * 4));
        PLF_HIP_TRY(hipMemcpyAsync(ta, a, (size_t)na * 32, hipMemcpyHostToDevice, s));
        PLF_HIP_TRY(hipMemcpyAsync(tb, b, (size_t)nb * 32, hipMemcpyHostToDevice, s));
        da = (const uint8_t *)ta; db = (const uint8_t *)tb; dd = (int *)td;
    }
    hipLaunchKernelGGL(k_hamming_matrix, dim3((nb + 255) / 256, na), dim3(256), 0, s, da, na, db, nb, dd);
    PLF_HIP_TRY(hipGetLastError());
    if (mem != PLF_MEM_DEVICE) {
        PLF_HIP_TRY(hipMemcpyAsync(dist, td, (size_t)na * nb * 4, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(ta); (void)hipFree(tb); (void)hipFree(td);
    }
    return PLF_OK;
}
