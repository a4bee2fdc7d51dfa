/*
 * plf.h -- C ABI of the MI355X-native point+line feature front-end ("plf") for RGB-D PL-SLAM.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: the hot path sits
 * behind C++ member calls, so every entry point below names the reference interface it replaces
 * (file:line under /root/reference).  Plain pointers and sizes only; no OpenCV, torch or C++ types.
 * INTEGRATION.md shows the adapter a maintainer adds on the reference side
 * (ORB_SLAM2::ORBextractor::operator() etc. forwarding to these calls).
 *
 * Memory kinds: every image/feature pointer is either host memory (PLF_MEM_HOST) or device (HBM)
 * memory of the handle's GPU (PLF_MEM_DEVICE).  `stream` arguments are hipStream_t passed as void*
 * (NULL = the handle's own stream).  Calls with device-resident inputs AND outputs only enqueue work
 * (asynchronous); calls touching host memory synchronise before returning.
 *
 * Error handling: 0 on success, negative plf_status otherwise; no exceptions cross this boundary
 * (the reference asserts or returns silently: include/ORBextractor.h:58-61, so@0x76dda).
 * Threading: a handle is confined to one thread at a time; different handles are independent
 * (PL-SLAM forks run ORB and LSD extraction concurrently from two threads).
 * Streams: the scratch buffers a handle owns are ordered by the stream its work was enqueued on.  Successive calls on one
 * handle may name different streams: every call records an event on its stream once its work is enqueued, and a call on
 * another stream makes that stream wait for the event on the DEVICE (no host wait; the previous call's stream is not touched
 * again, so it may have been destroyed meanwhile).  Size changes (a new width / height) re-upload the handle's tables after a
 * device-wide synchronisation.
 */
#ifndef PLF_H
#define PLF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PLF_OK = 0,
    PLF_E_EMPTY = -1,    /* empty image: outputs untouched (reference: silent return) */
    PLF_E_BADARG = -2,   /* unsupported size / parameter (reference: assert or UB) */
    PLF_E_CAPACITY = -3, /* caller's output capacity too small; outputs truncated */
    PLF_E_HIP = -4,      /* HIP runtime failure (no device, launch error, ...) */
    PLF_E_NOMEM = -5,
    PLF_E_RECTS = -6     /* line extractor, fully device-resident batches only: the batch produced more LSD rectangles than the pooled NFA buffers
                            hold (thousands per frame on average); its outputs are invalid -- redo it in smaller batches (calls that hand
                            host buffers in or out do that themselves) */
    , PLF_W_TRUNCATED = 1 /* not an error (plf_line_last_status only): at least one frame of the batch spent its plf_line_params.max_ms and was finished
                            with the line segments found so far; extraction calls themselves return PLF_OK for such a batch */
    , PLF_W_SLOW = 2      /* not an error, outputs complete and exact: plf_line_extract / plf_line_extract_batch with host outputs took more than `slow_factor`
                            (10) times the median per-frame time of this handle's recent calls at this image size, and more than `slow_floor_ms` (20 ms) per frame.
                            Images that are pathological for LSD (quantised textures: thousands of tiny regions) cost the GPU's serial region chain seconds per
                            frame; a caller with a frame deadline can answer the warning by creating the handle with plf_line_params.max_ms.  plf_line_tune
                            "slow_factor" = 0 switches the warning off.  Also reported by plf_line_last_status until the next call. */
} plf_status;

enum { PLF_MEM_HOST = 0, PLF_MEM_DEVICE = 1 };

/* bit-compatible with cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } plf_keypoint;

/* field order of cv::line_descriptor::KeyLine (68 bytes) */
typedef struct {
    float angle; int32_t class_id; int32_t octave; float pt_x, pt_y; float response; float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int32_t numOfPixels;
} plf_keyline;

/* cv::DMatch */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } plf_dmatch;

const char *plf_version(void);
const char *plf_status_string(int status);
int plf_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor -- replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:44-112)
 * ---------------------------------------------------------------------------------------------- */
typedef struct plf_orb plf_orb;

typedef struct {
    int32_t nfeatures;    /* ORBextractor.nFeatures   (Examples/RGB-D/TUM1.yaml:42) */
    float scale_factor;   /* ORBextractor.scaleFactor (TUM1.yaml:45) */
    int32_t nlevels;      /* ORBextractor.nLevels     (TUM1.yaml:48) */
    int32_t ini_th_fast;  /* ORBextractor.iniThFAST   (TUM1.yaml:54) */
    int32_t min_th_fast;  /* ORBextractor.minThFAST   (TUM1.yaml:55) */
    int32_t device;       /* HIP device ordinal */
    int32_t max_width, max_height; /* largest image the handle must accept */
    int32_t max_batch;    /* frames in flight per call (>= 1) */
} plf_orb_params;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  include/ORBextractor.h:51-52 */
int plf_orb_create(const plf_orb_params *params, plf_orb **out);
void plf_orb_destroy(plf_orb *h);

/* GetLevels/GetScaleFactor(s)/GetInverseScaleFactors/GetScaleSigmaSquares/GetInverseScaleSigmaSquares
 * include/ORBextractor.h:63-83; per_level = mnFeaturesPerLevel.  Arrays of nlevels entries; any may be NULL. */
int plf_orb_get_tables(const plf_orb *h, int32_t *nlevels, float *scale, float *inv_scale, float *sigma2,
                       float *inv_sigma2, int32_t *per_level);

/* Required output capacity (keypoints per frame) for this handle: nfeatures + 4 * nlevels. */
int plf_orb_capacity(const plf_orb *h);

/* void ORBextractor::operator()(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray descriptors)
 * include/ORBextractor.h:59-61 -- single frame, host memory in and out (mask is ignored by the reference).
 * gray: 8-bit single channel, `pitch` bytes per row.  kps/desc: capacity entries / capacity*32 bytes. */
int plf_orb_extract(plf_orb *h, const uint8_t *gray, int32_t width, int32_t height, ptrdiff_t pitch,
                    plf_keypoint *kps, uint8_t *desc, int32_t capacity, int32_t *n_out);

/* Same operator over a batch of independent frames (BASELINE configs 3-4).  Frame f starts at
 * gray + f*frame_stride; its outputs at kps + f*capacity, desc + f*capacity*32, n_out[f].
 * in_mem / out_mem: PLF_MEM_HOST or PLF_MEM_DEVICE (n_out follows out_mem). */
int plf_orb_extract_batch(plf_orb *h, const uint8_t *gray, int32_t in_mem, int32_t n_frames, int32_t width,
                          int32_t height, ptrdiff_t pitch, ptrdiff_t frame_stride, plf_keypoint *kps, uint8_t *desc,
                          int32_t *n_out, int32_t out_mem, int32_t capacity, void *stream);

/* Public member mvImagePyramid (include/ORBextractor.h:85): copy level `level` of frame `frame` of the last
 * call, INCLUDING its 19-pixel REFLECT_101 border, to host memory `dst` (pitch = level width + 38). */
int plf_orb_get_pyramid_level(plf_orb *h, int32_t frame, int32_t level, uint8_t *dst, int32_t *level_w, int32_t *level_h);
/* Test hook: the 7x7-blurred level image the descriptors were sampled from (pitch = level width). */
int plf_orb_get_blurred_level(plf_orb *h, int32_t frame, int32_t level, uint8_t *dst);
/* Test hook: FAST candidates (x, y relative to the 16-px border, response) handed to DistributeOctTree, reference order. */
int plf_orb_get_candidates(plf_orb *h, int32_t frame, int32_t level, float *xyr, int32_t capacity, int32_t *n_out);

/* ------------------------------------------------------------------------------------------------
 * Line extractor -- replaces ORB_SLAM2::LineSegment::ExtractLineSegment (include/ExtractLineSegment.h:38)
 * ---------------------------------------------------------------------------------------------- */
typedef struct plf_line plf_line;

typedef struct {
    int32_t nlines;      /* lines kept after sorting by response (lsdNFeatures of the fork; BASELINE: 100/200/400) */
    int32_t seed_order;  /* 0 = OpenCV 3.0-3.3 raster seed order (default), 1 = published bin order */
    int32_t device;
    int32_t max_width, max_height, max_batch;
    int32_t lbd_sobel_input; /* what BinaryDescriptor's two cv::Sobel calls read: PLF_LBD_BLURRED (0, default) = octave 0 of
                                BinaryDescriptor::computeGaussianPyramid, i.e. cv::GaussianBlur(image, Size(5, 5), 1) -- believed to be
                                opencv_contrib 3.3 behaviour; PLF_LBD_RAW (1) = the image as handed in.  DESIGN.md section 2 lists every
                                such version-dependent choice. */
    float max_ms;        /* time budget of LSD region growing per call, milliseconds; 0 (default) = unlimited, the reference's behaviour.  The reference has
                            no bound: a pathological image (e.g. a coarse checkerboard: net-like regions of 10^4-10^5 pixels) keeps one frame's serial chain
                            busy for seconds -- on the CPU as on the GPU.  With max_ms > 0 every frame of a call stops seeding regions once the budget is
                            spent and is finished with the rectangles found so far (NFA validation, top-N, LBD as usual): a DELIBERATE, opt-in deviation
                            from the reference -- the lines of a truncated frame are a prefix-like subset of the reference's, not bit-equal.  Reported
                            as PLF_W_TRUNCATED by plf_line_last_status and per frame by plf_line_truncated; frames that finish in time are bit-exact. */
} plf_line_params;
enum { PLF_LBD_BLURRED = 0, PLF_LBD_RAW = 1 };

int plf_line_create(const plf_line_params *params, plf_line **out);
void plf_line_destroy(plf_line *h);
/* Schedule knobs of a line-extractor handle (frames-in-flight thresholds of its schedules, band counts ...).  They are read from the environment ONCE, when the handle
 * is created (PLF_LSD_*, PLF_NFA_FUSED: experiments), and never again; this call changes one of them afterwards -- a tuning and test hook, not needed in production:
 * "spec_max" (frames in flight up to which the banded speculative schedule is used, 256), "spec_bands", "spec_z", "spec_rounds", "spec_halo", "spec_clip", "spec_fill",
 * "spec_fill_tol", "spec_stagger", "spec_nofuse", "spec_spins", "spec_reccap", "lat_max", "wpg", "slow_factor", "slow_floor_ms" (PLF_W_SLOW), "nfa_fused" (frames in flight up to which one wave per
 * rectangle runs all NFA stages, 64).  Every schedule gives the same bits.  PLF_E_BADARG for an unknown name or a value out of range. */
int plf_line_tune(plf_line *h, const char *name, double value);

/* void LineSegment::ExtractLineSegment(const Mat& img, vector<KeyLine>&, Mat& ldesc, vector<Vector3d>& lineFunctions,
 *                                      int scale = 1.2, int numOctaves = 1)      include/ExtractLineSegment.h:38
 * single frame, host memory.  ldesc: capacity*32 bytes (CV_8U rows); line_eq: capacity*3 doubles. */
int plf_line_extract(plf_line *h, const uint8_t *gray, int32_t width, int32_t height, ptrdiff_t pitch,
                     plf_keyline *lines, uint8_t *ldesc, double *line_eq, int32_t capacity, int32_t *n_out);

int plf_line_extract_batch(plf_line *h, const uint8_t *gray, int32_t in_mem, int32_t n_frames, int32_t width,
                           int32_t height, ptrdiff_t pitch, ptrdiff_t frame_stride, plf_keyline *lines, uint8_t *ldesc,
                           double *line_eq, int32_t *n_out, int32_t out_mem, int32_t capacity, void *stream);

/* Scheduling hook for pipelines that run several extractors on different streams: makes `stream` wait (device side,
 * hipStreamWaitEvent) until the throughput-bound front stages of the most recent plf_line_extract_batch -- blur,
 * resize, gradient, Sobel: everything before region growing -- have finished.  Region growing is a latency-bound
 * chain that leaves most issue slots idle; work queued behind this point overlaps with it instead of competing with
 * the front stages.  No-op if no batch was enqueued yet. */
int plf_line_wait_front(plf_line *h, void *stream);
/* Status of the last batch enqueued with device-resident inputs AND outputs (that call returns before the GPU has run): waits for `stream`
 * (NULL: the handle's stream), then PLF_OK, PLF_E_CAPACITY (a frame had more lines than `capacity`), PLF_E_RECTS, or PLF_W_TRUNCATED (> 0: a frame ran
 * out of plf_line_params.max_ms; valid after host-memory calls too). */
int plf_line_last_status(plf_line *h, void *stream);
/* flags[f] = 1 if frame f of the last batch ran out of plf_line_params.max_ms (waits for the stream of that call) */
int plf_line_truncated(plf_line *h, int32_t *flags, int32_t n);

/* Diagnostics of the banded speculative region growing used for <= 256 frames in flight, frame 0 of the last batch:
 * out8 = {regions committed from the speculation, regions grown by the commit wave, chunks committed in one step, records checked pixel by pixel,
 * kilo-cycles spent regrowing, validating, in total, in per-band setup}.  PLF_E_BADARG if the path has not run on this handle. */
int plf_line_debug_spec_stats(plf_line *h, int32_t *out8);

/* Diagnostics (tools/spec_redo.py) of the validation-round schedule (few frames in flight), per frame f < n_frames of the last batch: out[4 f + 0 / 1] = bands whose
 * marks changed in the last even / odd round, out[4 f + 2] = the round that found nothing left to change (0: none within the enqueued rounds), out[4 f + 3] = 1 if
 * the frame was finished by the serial commit wave (log overflow or no fixpoint): the schedule's redo.  PLF_E_BADARG if that schedule has not run on this handle. */
int plf_line_debug_spec_rounds(plf_line *h, int32_t *out, int32_t n_frames);

/* Diagnostics (tools/nfa_stats.py): out16[s] = rectangles of the last batch that entered rect_improve stage s (0..4; [5] = left over after stage 4), for batches
 * that took the staged NFA kernels (more than 64 frames in flight).  Synchronises the device. */
int plf_line_debug_nfa_counters(plf_line *h, int32_t *out16);

/* Diagnostics (bench.py): out[f] = length of frame f's region-growing chain in the last batch = pixels left marked USED (accept steps minus the pixels
 * refine released again), n <= frames of that batch.  The launch of the one-wave-per-frame kernel lasts as long as its longest chain.  Zero for batches
 * that took the speculative schedule (its flags live in LDS).  Synchronises the device. */
int plf_line_chain_lengths(plf_line *h, int32_t *out, int32_t n);

/* Diagnostics (bench.py, natural-image extras): out[f] = rectangles frame f of the last batch handed to the NFA validation (regions of min_reg_size pixels or more
 * that survived refine), n <= frames of that batch.  Synchronises the device. */
int plf_line_rect_counts(plf_line *h, int32_t *out, int32_t n);

/* Measurement hook (bench.py roofline): when enabled, every launch of the region-growing kernel -- the dominant
 * kernel of the whole front-end -- is bracketed by HIP events on the stream it is launched on.  The call
 * synchronises, then returns the accumulated kernel milliseconds and the number of launches since the last reset. */
int plf_line_profile(plf_line *h, int32_t enable, int32_t reset, double *ms_total, int32_t *launches);

/* Test hook: all LSD segments (x1,y1,x2,y2) of frame `frame` in detection order, before the top-N cut. */
int plf_line_get_segments(plf_line *h, int32_t frame, float *segs, int32_t capacity, int32_t *n_out);

/* ------------------------------------------------------------------------------------------------
 * Matchers -- replace ORB_SLAM2::ORBmatcher / LSDmatcher tracking overloads
 * ---------------------------------------------------------------------------------------------- */
/* static int ORBmatcher::DescriptorDistance(const Mat&, const Mat&)  include/ORBmatcher.h:44 (= LSDmatcher.h:43,
 * Thirdparty/DBoW2/DBoW2/FORB.cpp:82-102).  Host scalar utility: 256-bit Hamming distance. */
int plf_hamming256(const uint8_t *a, const uint8_t *b);
/* Batched on the GPU: dist[i*nb + j] = Hamming(a_i, b_j); pointers in `mem`. */
int plf_hamming256_matrix(const uint8_t *a, int32_t na, const uint8_t *b, int32_t nb, int32_t *dist, int32_t mem,
                          int32_t device, void *stream);

/* View of the Frame members the matchers read (include/Frame.h): all pointers device memory. */
typedef struct {
    int32_t n;                 /* N (upper bound when n_device is given) */
    const int32_t *n_device;   /* optional: N lives in device memory (output of plf_orb_extract_batch); NULL = use n */
    const plf_keypoint *keys_un; /* mvKeysUn (pt, octave, angle are read) */
    const float *uright;       /* mvuRight, NULL = monocular (-1 everywhere) */
    const uint8_t *desc;       /* mDescriptors, n x 32 */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    const float *scale_factors;  /* mvScaleFactors (device, nlevels) */
    int32_t nlevels;
} plf_frame_view;

/* MapPoint tracking fields (include/MapPoint.h:94-105) flattened; device memory */
typedef struct {
    int32_t m;
    const float *proj_x, *proj_y, *proj_xr; /* mTrackProjX, mTrackProjY, mTrackProjXR */
    const int32_t *level;                   /* mnTrackScaleLevel */
    const float *view_cos;                  /* mTrackViewCos */
    const uint8_t *in_view;                 /* mbTrackInView && !isBad() */
    const uint8_t *desc;                    /* GetDescriptor(), m x 32 */
    const uint8_t *obs_positive;            /* Observations() > 0, NULL = all */
} plf_mappoint_view;

typedef struct plf_matcher plf_matcher;
int plf_matcher_create(int32_t device, int32_t max_keypoints, int32_t max_mappoints, int32_t max_lines,
                       int32_t max_batch, plf_matcher **out);
void plf_matcher_destroy(plf_matcher *h);

/* int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
 * include/ORBmatcher.h:61 (so@0x79f10), with F.GetFeaturesInArea include/Frame.h:113 (so@0xfbc60) and
 * AssignFeaturesToGrid (so@0xf9120) built on the device.  Batched over n_frames frames that each
 * match the same map-point set (replicas).
 * match_of_kp (device, n_frames x kp_stride int32), in/out: -1 = free, -2 = already holds a MapPoint with
 * observations; on return >= 0 = index of the map point assigned by this call.
 * nmatches (device, n_frames int32) = the reference's return value. nnratio = mfNNratio. */
int plf_match_project_points(plf_matcher *h, const plf_frame_view *frames /*host array of views*/, int32_t n_frames,
                             const plf_mappoint_view *mp, float th, float nnratio, int32_t *match_of_kp,
                             int32_t kp_stride, int32_t *nmatches, void *stream);

/* LastFrame members read by the motion-model overload; device memory */
typedef struct {
    int32_t n;
    const uint8_t *has_mappoint; /* mvpMapPoints[i] != NULL */
    const uint8_t *outlier;      /* mvbOutlier[i] */
    const float *world_pos;      /* n x 3, pMP->GetWorldPos() */
    const plf_keypoint *keys;    /* mvKeys / mvKeysUn (octave, angle) */
    const uint8_t *mp_desc;      /* n x 32, pMP->GetDescriptor() */
    const uint8_t *obs_positive; /* pMP->Observations() > 0; NULL = all.  0 marks the temporal points of localisation mode
                                    (Tracking::UpdateLastFrame, include/Tracking.h:152, mlpTemporalPoints :252): the reference skips a key point
                                    only if its map point HAS observations (so@0x81e3d), so one taken by such a point is overwritten by a later
                                    last-frame point; the return value and the rotation histogram count every assignment.  Read by
                                    plf_match_project_lastframe only (plf_match_project_keyframe tests the pointer alone). */
} plf_lastframe_view;

typedef struct { float Rcw[9], tcw[3], Rlw[9], tlw[3]; float fx, fy, cx, cy, bf, b; } plf_pose_pair;

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
 * include/ORBmatcher.h:78 (so@0x80d00).  match_of_kp as above (values >= 0 are last-frame indices).  check_orientation: 0 / 1 = mbCheckOrientation;
 * 2 = as 1, and a key point whose assignment the rotation-histogram check removed is marked -3 instead of -1: the reference sets
 * CurrentFrame.mvpMapPoints[k] = NULL there, which an adapter can only tell from "never assigned" (possibly still holding a map point without observations)
 * with the distinct value (plf.hpp does so).  On INPUT -3 reads as -1 (free) in every matcher, so the array of one call can be handed to the next. */
int plf_match_project_lastframe(plf_matcher *h, const plf_frame_view *cur, const plf_lastframe_view *last,
                                const plf_pose_pair *pose, float th, int32_t mono, int32_t check_orientation,
                                int32_t *match_of_kp, int32_t *nmatches, void *stream);

/* The same overload for a batch of independent current frames against ONE last frame (a replica, like the local map of
 * plf_match_project_points): frame f uses poses[f] (HOST array of n_frames), its assignments go to match_of_kp + f * kp_stride and its
 * return value to nmatches[f]. */
int plf_match_project_lastframe_batch(plf_matcher *h, const plf_frame_view *frames, int32_t n_frames, const plf_lastframe_view *last,
                                      const plf_pose_pair *poses, float th, int32_t mono, int32_t check_orientation, int32_t *match_of_kp,
                                      int32_t kp_stride, int32_t *nmatches, void *stream);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint*> &sAlreadyFound,
 *                                    const float th, const int ORBdist)   include/ORBmatcher.h:82 (so@0x7e8c0, relocalisation).
 * kf: the keyframe's features -- has_mappoint[i] = pKF->GetMapPointMatches()[i] != NULL && !isBad() && !sAlreadyFound.count(pMP),
 * world_pos, mp_desc, keys = pKF->mvKeysUn (angle); `outlier` is not read.  min_distance / max_distance (device, kf->n floats) =
 * MapPoint::mfMinDistance / mfMaxDistance (the 0.8 / 1.2 invariance factors are applied here); log_scale_factor =
 * CurrentFrame.mfLogScaleFactor; pose: Rcw, tcw, fx, fy, cx, cy of the current frame (Rlw, tlw, bf, b unused).
 * match_of_kp as above: values >= 0 are keyframe feature indices; EVERY key point that already holds a map point must be
 * pre-set to -2 (this overload tests the pointer only, not Observations()). */
int plf_match_project_keyframe(plf_matcher *h, const plf_frame_view *cur, const plf_lastframe_view *kf, const float *min_distance,
                               const float *max_distance, const plf_pose_pair *pose, float log_scale_factor, float th, int32_t orb_dist,
                               int32_t check_orientation, int32_t *match_of_kp, int32_t *nmatches, void *stream);

/* void Frame::AssignFeaturesToGrid()  include/Frame.h:222 (so@0xf9120) with Frame::PosInGrid (so@0xf5fa0): the public
 * std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS] (include/Frame.h:226) of one frame as a CSR -- every matcher above builds
 * it on the device; this entry hands it out.  cell_start (HOST, 64*48 + 1 int32; cell = ix*48 + iy), cell_idx (HOST, frame->n int32:
 * key point indices, insertion order inside a cell). */
int plf_match_assign_grid(plf_matcher *h, const plf_frame_view *frame, int32_t *cell_start, int32_t *cell_idx, void *stream);

/* Map points handed to Fuse / the Sim3 searches (vector<MapPoint*>), flattened; DEVICE memory */
typedef struct {
    int32_t m;
    const float *world_pos;      /* m x 3: GetWorldPos() */
    const float *normal;         /* m x 3: GetNormal() (unread by overloads without a viewing-angle test) */
    const float *min_distance;   /* mfMinDistance (the 0.8 factor of GetMinDistanceInvariance is applied here) */
    const float *max_distance;   /* mfMaxDistance (the 1.2 factor of GetMaxDistanceInvariance is applied here) */
    const uint8_t *desc;         /* m x 32: GetDescriptor() */
    const uint8_t *valid;        /* pMP != NULL && !isBad() && not already in the keyframe / found set */
} plf_points3d_view;

/* KeyFrame pose and intrinsics read by these searches; HOST struct */
typedef struct {
    float Rcw[9], tcw[3], Ow[3];   /* GetRotation(), GetTranslation(), GetCameraCenter() */
    float fx, fy, cx, cy, bf;      /* KeyFrame::fx, fy, cx, cy, mbf */
    float log_scale_factor;        /* mfLogScaleFactor */
    const float *inv_level_sigma2; /* mvInvLevelSigma2 (host, kf->nlevels <= 16 floats) */
} plf_kf_pose;

/* int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th)   include/ORBmatcher.h:119 (so@0x7a500)
 * -- the search half: for every map point the keyframe key point it is fused with.  kf = the keyframe's mvKeysUn / mvuRight /
 * mDescriptors / image bounds / mvScaleFactors (KeyFrame::GetFeaturesInArea so@0x96fe0, IsInImage so@0x97480).
 * best_idx (device, m int32): key point index, -1 = none; nfused (device int32) = the reference's return value.
 * The map mutation of the reference loop -- Replace / AddObservation / AddMapPoint, decided by pKF->GetMapPoint(best_idx) --
 * reads nothing the search depends on and is applied by the caller in list order. */
int plf_match_fuse(plf_matcher *h, const plf_frame_view *kf, const plf_kf_pose *pose, const plf_points3d_view *pts, float th,
                   int32_t *best_idx, int32_t *nfused, void *stream);

/* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs,
 *                                        const bool bOnlyStereo)   include/ORBmatcher.h:111 (so@0x86b30; LocalMapping::CreateNewMapPoints), with
 * ORBmatcher::CheckDistEpipolarLine (so@0x79b90).  All arrays of the view in DEVICE memory; the DBoW2 feature vectors flattened as for plf_match_bow.
 * has_mp1/2[i] = pKF->GetMapPoint(i) != NULL (only features WITHOUT a map point are paired).  F12 (9, row-major), Cw1 = pKF1->GetCameraCenter() (3): HOST.
 * pose2: Rcw, tcw, fx, fy, cx, cy of pKF2 (epipole).  mbCheckOrientation = check_orientation.
 * match12 (device, n1 int32, overwritten): keyframe-2 feature paired with keyframe-1 feature i1, -1 = none -- the reference's vMatchedPairs is
 * {(i1, match12[i1])} in ascending i1.  nmatches (device int32) = the return value (-1: a feature listed in two nodes of keyframe 1). */
typedef struct plf_tri_view {
    int32_t n1, n2;
    const plf_keypoint *keys1, *keys2;   /* mvKeysUn */
    const float *uright1, *uright2;      /* mvuRight */
    const uint8_t *desc1, *desc2;        /* mDescriptors */
    const uint8_t *has_mp1, *has_mp2;
    int32_t nodes1, nodes2;
    const uint32_t *node_id1, *node_id2;
    const int32_t *node_start1, *node_start2;
    const int32_t *feat1, *feat2;
    const float *scale_factors2;         /* pKF2->mvScaleFactors */
    const float *level_sigma2_2;         /* pKF2->mvLevelSigma2 */
} plf_tri_view;
int plf_match_triangulation(plf_matcher *h, const plf_tri_view *v, const float *F12, const float *Cw1, const plf_kf_pose *pose2,
                            int32_t only_stereo, int32_t check_orientation, int32_t *match12, int32_t *nmatches, void *stream);

/* int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint)
 * include/ORBmatcher.h:122 (so@0x7bb20, loop closing) -- the search half.  Scw: the 4x4 CV_32F Sim3 matrix, row-major, HOST; it is
 * decomposed as the reference does (scale from row 0, Rcw = sRcw / s, tcw, Ow = -Rcw.t() * tcw).  intr: fx, fy, cx, cy,
 * log_scale_factor of the keyframe (Rcw, tcw, Ow, bf, inv_level_sigma2 unread).  valid[i] = !isBad() && !pKF->GetMapPoints().count(pMP).
 * best_idx[i]: the caller sets vpReplacePoint[i] = pKF->GetMapPoint(best_idx[i]) when that slot holds a good point, otherwise adds
 * the observation -- in list order, as the reference loop does. */
int plf_match_fuse_sim3(plf_matcher *h, const plf_frame_view *kf, const float *Scw, const plf_kf_pose *intr, const plf_points3d_view *pts,
                        float th, int32_t *best_idx, int32_t *nfused, void *stream);

/* int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
 * include/ORBmatcher.h:86 (so@0x880f0, loop closing).  valid[i] = !isBad() && pMP not among vpMatched on entry.
 * match_of_kp (device, kf->n int32), in/out: -1 = vpMatched[k] == NULL, -2 = occupied; on return >= 0 = index of the map point the call
 * stored there (greedy in list order, exactly the reference loop).  nmatches (device int32) = the return value. */
int plf_match_project_sim3(plf_matcher *h, const plf_frame_view *kf, const float *Scw, const plf_kf_pose *intr, const plf_points3d_view *pts,
                           int32_t th, int32_t *match_of_kp, int32_t *nmatches, void *stream);

/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
 *                              const cv::Mat &t12, const float th)   include/ORBmatcher.h:116 (so@0x838b0, loop closing).
 * pts1 / pts2: one entry per key point of the keyframe (m == n): GetMapPointMatches() flattened, valid = pointer non-null && !isBad() &&
 * not matched on entry (vbAlreadyMatched1/2).  pose1 / pose2: Rcw, tcw, log_scale_factor; fx, fy, cx, cy are taken from pose1 for BOTH
 * directions, as the reference does.  R12 (9, row-major), t12 (3): HOST.  match12 (device, kf1->n int32): key point of keyframe 2 whose
 * map point the caller stores in vpMatches12[i1], -1 = untouched.  nfound (device int32) = the return value. */
int plf_match_sim3(plf_matcher *h, const plf_frame_view *kf1, const plf_frame_view *kf2, const plf_kf_pose *pose1, const plf_kf_pose *pose2,
                   float s12, const float *R12, const float *t12, float th, const plf_points3d_view *pts1, const plf_points3d_view *pts2,
                   int32_t *match12, int32_t *nfound, void *stream);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)
 * include/ORBmatcher.h:104 (so@0x80150) -- the tracker's reference-keyframe / relocalisation matcher.
 * One view per (keyframe, frame) pair, all arrays in DEVICE memory.  The DBoW2 feature vectors
 * (std::map<NodeId, std::vector<unsigned>>, KeyFrame::mFeatVec / Frame::mFeatVec) are passed flattened in key order:
 * node ids (ascending), CSR starts (nodes + 1), feature indices.  kf_has_mp[i] = pKF->GetMapPointMatches()[i] != NULL
 * && !isBad(); kf_angle = pKF->mvKeysUn[i].angle, f_angle = F.mvKeys[j].angle.  mfNNratio and mbCheckOrientation are
 * the ORBmatcher constructor arguments.
 * match_of_f (device, n_pairs x stride int32, overwritten): index of the keyframe feature whose map point frame
 * feature j received (vpMapPointMatches[j] = vpMapPointsKF[match_of_f[j]]), -1 = NULL.  nmatches (device, n_pairs). */
typedef struct plf_bow_view {
    int32_t n_kf, n_f;
    const uint8_t *kf_desc, *f_desc;
    const float *kf_angle, *f_angle;
    const uint8_t *kf_has_mp;
    const uint8_t *f_has_mp;                  /* plf_match_bow_kf only (second keyframe); NULL for a Frame */
    int32_t kf_nodes, f_nodes;
    const uint32_t *kf_node_id, *f_node_id;
    const int32_t *kf_node_start, *f_node_start;
    const int32_t *kf_feat, *f_feat;
} plf_bow_view;
int plf_match_bow(plf_matcher *h, const plf_bow_view *pairs, int32_t n_pairs, float nnratio, int32_t check_orientation,
                  int32_t *match_of_f, int32_t stride, int32_t *nmatches, void *stream);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12)
 * include/ORBmatcher.h:105 (so@0x82cc0, loop closing).  Same view type: kf_* = pKF1, f_* = pKF2 (f_angle =
 * pKF2->mvKeysUn[j].angle, f_has_mp = pKF2 map point present && !isBad()).  match12 (device, n_pairs x stride, overwritten):
 * per KF1 feature the KF2 feature whose map point it received, -1 = NULL.  nmatches[i] = -1 flags a pair whose first
 * feature vector lists a feature twice (not a DBoW2 FeatureVector). */
int plf_match_bow_kf(plf_matcher *h, const plf_bow_view *pairs, int32_t n_pairs, float nnratio, int32_t check_orientation,
                     int32_t *match12, int32_t stride, int32_t *nmatches, void *stream);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2) as used by LSDmatcher (include/LSDmatcher.h:19-35).
 * out: nq x 2 plf_dmatch in `mem`. */
int plf_match_lines_knn(plf_matcher *h, const uint8_t *query, int32_t nq, const uint8_t *train, int32_t nt,
                        plf_dmatch *out, int32_t mem, void *stream);

/* void LineSegment::LineSegmentMathch(Mat &ldesc1, Mat &ldesc2) + void LineSegment::LineDescriptorMAD()  include/ExtractLineSegment.h:41,44
 * (the same statistic as Frame::lineDescriptorMAD(vector<vector<DMatch>>, double &nn_mad, double &nn12_mad)  include/Frame.h:75): the 2-NN table of
 * ldesc1 (n1 x 32) against ldesc2 (n2 x 32, n2 >= 2) and its two robust spreads, mad[0] = nn_mad = 1.4826 * median |d1 - median d1|,
 * mad[1] = nn12_mad = the same on d2 - d1.  knn (optional, n1 x 2 plf_dmatch = mvlineMatches) and mad (2 doubles) both live in `mem`. */
int plf_line_descriptor_mad(plf_matcher *h, const uint8_t *ldesc1, int32_t n1, const uint8_t *ldesc2, int32_t n2, plf_dmatch *knn,
                            double *mad, int32_t mem, void *stream);

/* double LineSegment::LineSegmentOverlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj)  include/ExtractLineSegment.h:47
 * (declared without a body in the snapshot; restated from the PL-SLAM family's lineSegmentOverlap, PARITY UNPINNED): overlap of the observed
 * interval [min, max](spl_obs, epl_obs) with the projected one, divided by length = max(obs) - min(proj); 0 when the intervals are disjoint or
 * length <= 0.01.  A host scalar (no device work), exported so that every binding shares one definition. */
double plf_line_segment_overlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj);

/* int LSDmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)  include/LSDmatcher.h:32
 * with Frame::lineDescriptorMAD include/Frame.h:75.  last_has_mapline[q] = LastFrame.mvpMapLines[q] != NULL.
 * match_of_line (device, ncur int32, pre-set to -1): last-frame line index assigned to each current line.
 * The keyframe overload, int LSDmatcher::SearchByProjection(KeyFrame *pKF, Frame &F, vector<MapLine*> &vpMapLineMatches)
 * include/LSDmatcher.h:35 (KeyFrame::lineDescriptorMAD include/KeyFrame.h:159), is the same rule with the keyframe's line descriptors as the
 * query side: pass pKF->mLineDescriptors as last_desc and last_has_mapline[q] = pKF->GetMapLineMatches()[q] != NULL. */
int plf_match_lines_lastframe(plf_matcher *h, const uint8_t *last_desc, int32_t nlast, const uint8_t *cur_desc,
                              int32_t ncur, const uint8_t *last_has_mapline, int32_t *match_of_line, int32_t *nmatches,
                              void *stream);

/* Frame members read by the line matchers (include/Frame.h:201-214); device memory */
typedef struct {
    int32_t n;                   /* number of lines (upper bound when n_device is given) */
    const int32_t *n_device;     /* optional: count in device memory (output of plf_line_extract_batch) */
    const plf_keyline *lines_un; /* mvKeylinesUn (pt, angle, octave) */
    const uint8_t *desc;         /* mLdesc */
    const float *scale_factors;
} plf_lineframe_view;

/* LSDmatcher::SearchByProjection(Cur, Last) for a batch of independent current frames against ONE last frame (BF kNN k = 2 on the LBD
 * descriptors + the lineDescriptorMAD rule per frame): frames[f].desc / n / n_device are read; frame f writes match_of_line + f * line_stride
 * (pre-set to -1 by the caller) and nmatches[f]. */
int plf_match_lines_lastframe_batch(plf_matcher *h, const uint8_t *last_desc, int32_t nlast, const uint8_t *last_has_mapline,
                                    const plf_lineframe_view *frames, int32_t n_frames, int32_t *match_of_line, int32_t line_stride,
                                    int32_t *nmatches, void *stream);

/* int LSDmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo)
 * include/LSDmatcher.h:54 (LocalMapping::CreateNewMapLines; body absent from the snapshot -- PL-SLAM family rule, "parity unpinned"): brute-force
 * Hamming kNN (k = 2) of pKF1->mLineDescriptors against pKF2->mLineDescriptors, KeyFrame::lineDescriptorMAD (include/KeyFrame.h:159), a pair is
 * kept when d2 - d1 > mad_factor * nn12_mad (upstream 0.1) and neither line holds a MapLine (has_ml1 / has_ml2 = GetMapLine(i) != NULL); with
 * bOnlyStereo both lines also need stereo data (stereo1 / stereo2; may be NULL otherwise).  All arrays DEVICE memory.
 * match12 (n1 int32, overwritten): keyframe-2 line paired with keyframe-1 line q, -1 = none -- vMatchedPairs = {(q, match12[q])} in ascending q.
 * nmatches (device int32) = the return value. */
int plf_match_lines_triangulation(plf_matcher *h, const uint8_t *desc1, int32_t n1, const uint8_t *desc2, int32_t n2, const uint8_t *has_ml1,
                                  const uint8_t *has_ml2, const uint8_t *stereo1, const uint8_t *stereo2, int32_t only_stereo, float mad_factor,
                                  int32_t *match12, int32_t *nmatches, void *stream);

/* int LSDmatcher::Fuse(KeyFrame *pKF, const vector<MapLine*> &vpMapLines)   include/LSDmatcher.h:58 (LocalMapping::SearchInNeighbors; body absent --
 * PL-SLAM family rule, "parity unpinned") -- the search half: for every map line with valid[i] (non-NULL, !isBad(), !IsInKeyFrame(pKF)) the
 * keyframe line nearest in Hamming distance over ALL of pKF->mLineDescriptors (first minimum), fused when the distance is <= TH_LOW (50).
 * best_idx (device, m int32): keyframe line index or -1; nfused (device int32) = the return value.  As for ORBmatcher::Fuse the map mutation
 * (Replace / AddObservation / AddMapLine, decided by pKF->GetMapLine(best_idx[i])) is applied by the caller in list order. */
int plf_match_lines_fuse(plf_matcher *h, const uint8_t *kf_desc, int32_t n_kf, const uint8_t *ml_desc, const uint8_t *valid, int32_t m,
                         int32_t *best_idx, int32_t *nfused, void *stream);

/* MapLine tracking fields include/MapLine.h:113-129 */
typedef struct {
    int32_t m;
    const float *x1, *y1, *x2, *y2; /* mTrackProjX1, Y1, X2, Y2 */
    const int32_t *level;
    const float *view_cos;
    const uint8_t *in_view;
    const uint8_t *desc;
} plf_mapline_view;

/* int LSDmatcher::SearchByProjection(Frame &F, const vector<MapLine*> &vpMapLines, const float th)
 * include/LSDmatcher.h:40 with Frame::GetLinesInArea include/Frame.h:116. */
int plf_match_project_lines(plf_matcher *h, const plf_lineframe_view *frames, int32_t n_frames,
                            const plf_mapline_view *ml, float th, float nnratio, int32_t *match_of_line,
                            int32_t line_stride, int32_t *nmatches, void *stream);

/* ------------------------------------------------------------------------------------------------
 * RGB-D ingest, Frame tail and frustum projection -- the stages either side of
 * the extractor + matcher hot path.  Stateless; all pointers are DEVICE memory; asynchronous on `stream`.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float fx, fy, cx, cy, k1, k2, p1, p2, k3, bf; } plf_camera;      /* Camera.* keys of TUM1.yaml:8-32 */
typedef struct { float Rcw[9], tcw[3], Ow[3]; } plf_frustum_pose;                  /* mRcw, mtcw, mOw of the Frame */

/* Tracking::GrabImageRGBD colour conversion, cv::cvtColor(RGB2GRAY / BGR2GRAY) (so@0x522e6); 3 bytes per pixel */
int plf_rgb_to_gray(const uint8_t *rgb, int32_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch, ptrdiff_t frame_stride,
                    int32_t bgr_order, uint8_t *gray, ptrdiff_t gray_pitch, ptrdiff_t gray_frame_stride, int32_t device, void *stream);
/* imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) (so@0x5206d); out: n_frames x height x width floats */
int plf_depth_to_float(const uint16_t *depth, int32_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch_elems,
                       ptrdiff_t frame_stride_elems, float factor, float *out, int32_t device, void *stream);
/* Frame::UndistortKeyPoints (so@0xf8630) + Frame::ComputeStereoFromRGBD (so@0xf6860) for n_frames key point sets laid out
 * like the outputs of plf_orb_extract_batch (kp_stride entries per frame; count per frame from n_device, or n_host).
 * depth: n_frames x height x width floats (NULL: monocular, uright = -1).  keys_un = mvKeysUn, uright = mvuRight,
 * kp_depth = mvDepth (may be NULL). */
int plf_frame_tail(const plf_keypoint *keys, const int32_t *n_device, int32_t n_host, int32_t n_frames, int32_t kp_stride,
                   const float *depth, int32_t width, int32_t height, const plf_camera *cam, plf_keypoint *keys_un, float *uright,
                   float *kp_depth, int32_t device, void *stream);
/* The line half of the Frame tail: void Frame::UndistortKeyLines() include/Frame.h:267 (-> mvKeylinesUn, :207) and the end-point fields
 * mvuRightLineStart / mvuRightLineEnd / mvDepthLineStart / mvDepthLineEnd (include/Frame.h:208-211; KeyFrame.h:224-229 copies them).  The
 * fork snapshot declares these without a body (and the binary is the point-only build), so they are defined exactly like the point fields:
 * each end point goes through the UndistortKeyPoints arithmetic and the ComputeStereoFromRGBD rule (depth at the truncated distorted position,
 * uRight = undistorted x - bf / d, -1 without a positive depth); every other KeyLine field is copied.  Layout like plf_line_extract_batch's
 * outputs (line_stride entries per frame; counts from n_device or n_host).  Output pointers other than lines_un may be NULL. */
int plf_frame_line_tail(const plf_keyline *lines, const int32_t *n_device, int32_t n_host, int32_t n_frames, int32_t line_stride,
                        const float *depth, int32_t width, int32_t height, const plf_camera *cam, plf_keyline *lines_un,
                        float *uright_start, float *uright_end, float *depth_start, float *depth_end, int32_t device, void *stream);
/* bool Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit) include/Frame.h:104 (so@0xf5190) with
 * MapPoint::PredictScale (so@0x8fc20) for m map points: fills the plf_mappoint_view fields the matcher reads.
 * min/max_distance = mfMinDistance / mfMaxDistance (the 0.8 / 1.2 invariance factors are applied inside). */
int plf_frustum_points(const float *world_pos, const float *normal, const float *min_distance, const float *max_distance, int32_t m,
                       const plf_frustum_pose *pose, const plf_camera *cam, float min_x, float min_y, float max_x, float max_y,
                       float log_scale_factor, int32_t nlevels, float viewing_cos_limit, float *proj_x, float *proj_y, float *proj_xr,
                       int32_t *level, float *view_cos, uint8_t *in_view, int32_t device, void *stream);
/* bool Frame::isInFrustum(MapLine *pML, float viewingCosLimit) include/Frame.h:107 for m map lines; fills the plf_mapline_view fields
 * (mTrackProjX1/Y1/X1R, X2/Y2/X2R, mnTrackScaleLevel, mTrackViewCos, mbTrackInView: include/MapLine.h:113-129).  The snapshot declares it
 * without a body: it is the MapPoint routine applied to the segment -- both end points (world_pos: m x 6 floats, start xyz then end xyz =
 * MapLine::mWorldPos, include/MapLine.h:158) must project in front of the camera and inside the bounds; distance gate, viewing cosine against
 * `normal` (mNormalVector, :166) and MapLine::PredictScale (:101) are taken at the midpoint.  x1r / x2r may be NULL. */
int plf_frustum_lines(const float *world_pos, const float *normal, const float *min_distance, const float *max_distance, int32_t m,
                      const plf_frustum_pose *pose, const plf_camera *cam, float min_x, float min_y, float max_x, float max_y,
                      float log_scale_factor, int32_t nlevels, float viewing_cos_limit, float *x1, float *y1, float *x1r, float *x2,
                      float *y2, float *x2r, int32_t *level, float *view_cos, uint8_t *in_view, int32_t device, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Batch driver -- the caller loop of the reference, Examples/RGB-D/rgbd_tum.cc:84-128 (read image -> SLAM.TrackRGBD ->
 * Tracking::GrabImageRGBD include/Tracking.h:69 -> Frame ctor -> ExtractORB / ExtractLSD), for a batch of INDEPENDENT
 * frames in HOST memory (BASELINE configs 3-4).  Frames are cut into contiguous blocks, one block per GPU
 * (plf_batch_shard); every GPU has one host worker thread, its own ORB / line / matcher handles and HIP streams, pinned
 * double-buffered staging and asynchronous H2D / D2H copies, so that the upload of chunk k+1 and the download of chunk
 * k-1 overlap the kernels of chunk k.  No collective, no peer traffic: frames are independent, the local map is a
 * replica per GPU.  One process can drive all GPUs of a node (n_devices = 0), or one GPU per process / rank
 * (n_devices = 1 with the rank's share of the frames from plf_batch_shard) -- same code path per device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct plf_batch plf_batch;

enum { PLF_FMT_GRAY8 = 0, PLF_FMT_RGB8 = 1, PLF_FMT_BGR8 = 2 };   /* Tracking::GrabImageRGBD converts RGB / BGR to gray (so@0x522e6) */

typedef struct {
    plf_orb_params orb;        /* orb.nfeatures <= 0: no ORB extractor.  orb.device and orb.max_batch are set by the driver */
    plf_line_params line;      /* line.nlines <= 0: no line extractor.   line.device and line.max_batch likewise */
    int32_t n_devices;         /* 0 = every visible GPU */
    const int32_t *devices;    /* n_devices ordinals, NULL = 0 .. n_devices-1 */
    int32_t frames_in_flight;  /* frames per GPU and pipeline slot (the handles' max_batch); BASELINE config 3-4: 8 */
    int32_t input_format;      /* PLF_FMT_* of the frames handed to plf_batch_extract */
    int32_t max_mappoints;     /* > 0: a matcher per GPU; capacity of the local-map replicas (plf_batch_set_local_map) */
    int32_t max_maplines;
    int32_t rgbd;              /* 1: also allocate the depth / Frame-tail buffers of plf_batch_extract_rgbd */
} plf_batch_params;

/* Host-side outputs of one batch: n_frames x capacity entries, frame f at index f * capacity (same layout as
 * plf_orb_extract_batch / plf_line_extract_batch with PLF_MEM_HOST).  Pointers of a disabled extractor may be NULL;
 * the match arrays are only written when a local map is set and may be NULL otherwise.
 * match_of_kp / match_of_line: -1 = none, >= 0 = index into the local map (ORBmatcher::SearchByProjection(Frame&, map points, th)
 * include/ORBmatcher.h:61; LSDmatcher::SearchByProjection(Frame&, map lines, th) include/LSDmatcher.h:40).
 * n_kp_matches / n_line_matches count the entries >= 0 of the frame's match_of_* row AS RETURNED: when kp_capacity / line_capacity cut
 * features (PLF_E_CAPACITY), the matches of the cut ones are not counted. */
typedef struct {
    plf_keypoint *kps; uint8_t *desc; int32_t *n_kps; int32_t kp_capacity;
    plf_keyline *lines; uint8_t *ldesc; double *line_eq; int32_t *n_lines; int32_t line_capacity;
    int32_t *match_of_kp; int32_t *n_kp_matches;
    int32_t *match_of_line; int32_t *n_line_matches;
} plf_batch_outputs;

/* Contiguous block partition used by the driver (and by bench.py's ranks): part `part` of `parts` owns frames
 * [*first, *first + *count), with first = part * n / parts.  Pure host arithmetic. */
int plf_batch_shard(int64_t n_frames, int32_t parts, int32_t part, int64_t *first, int64_t *count);

int plf_batch_create(const plf_batch_params *params, plf_batch **out);
void plf_batch_destroy(plf_batch *b);
int plf_batch_device_count(const plf_batch *b);
/* device ordinal of worker `i` */
int plf_batch_device(const plf_batch *b, int32_t i);

/* Replicates the tracking fields of the local map (HOST arrays of the two views; either may be NULL) on every GPU of the
 * driver.  th / nnratio: the arguments of the two SearchByProjection calls.  m = 0 clears. */
int plf_batch_set_local_map(plf_batch *b, const plf_mappoint_view *points, const plf_mapline_view *lines, float th, float nnratio,
                            float min_x, float min_y, float max_x, float max_y);

/* The frames of one batch, HOST memory (pageable, or pinned -- plf_host_alloc -- in which case the staging copy is
 * skipped): frame f starts at images + f * frame_stride, rows `pitch` bytes apart, 1 (gray) or 3 (RGB / BGR) bytes per
 * pixel.  Returns when every output is in place: PLF_OK, PLF_E_CAPACITY (some frame had more features than the caller's
 * capacity; outputs truncated) or the first hard error of any worker. */
int plf_batch_extract(plf_batch *b, const uint8_t *images, int64_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch,
                      ptrdiff_t frame_stride, const plf_batch_outputs *out);

/* RGB-D frames: the whole RGB-D Frame constructor (include/Frame.h:60; called by Tracking::GrabImageRGBD include/Tracking.h:69, which first
 * converts the colour image to gray, so@0x522e6, and the depth image with mDepthMapFactor, so@0x5206d, include/Tracking.h:233) for a batch of
 * independent frames: ExtractORB + ExtractLSD, then UndistortKeyPoints (so@0xf8630), ComputeStereoFromRGBD (so@0xf6860) and the line half
 * (UndistortKeyLines include/Frame.h:267, end-point depths :208-211) on the device, i.e. plf_frame_tail / plf_frame_line_tail applied to every
 * frame of the batch before anything is copied back.  With a local map set, SearchByProjection then reads mvKeysUn / mvuRight / mvKeylinesUn as
 * the reference does (without this call it sees the distorted key points and uright = none).  Requires plf_batch_params.rgbd = 1.
 * depth: n_frames images of uint16 (the TUM png files; depth_factor = 1 / DepthMapFactor, Examples/RGB-D/TUM1.yaml:35: 5000), HOST memory, rows
 * depth_pitch_elems apart; NULL = no depth (uright = -1, depths = -1: the monocular rule of the same functions).
 * Outputs use kp_capacity / line_capacity of `out`; any pointer may be NULL. */
typedef struct {
    plf_camera cam;
    float depth_factor;
    const uint16_t *depth; ptrdiff_t depth_pitch_elems, depth_frame_stride_elems;
    plf_keypoint *kps_un; float *uright; float *kp_depth;                                   /* mvKeysUn, mvuRight, mvDepth */
    plf_keyline *lines_un; float *uright_start, *uright_end, *depth_start, *depth_end;     /* mvKeylinesUn, mvuRightLineStart/End, mvDepthLineStart/End */
} plf_batch_rgbd;
int plf_batch_extract_rgbd(plf_batch *b, const uint8_t *images, int64_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch,
                           ptrdiff_t frame_stride, const plf_batch_outputs *out, const plf_batch_rgbd *rgbd);

/* Frames of the last plf_batch_extract* call whose LSD region growing ran out of plf_batch_params.line.max_ms and were finished with the segments found
 * until then (0 without a budget; a deliberate opt-in deviation, see plf_line_params.max_ms). */
int64_t plf_batch_truncated_frames(const plf_batch *b);

/* Seconds the workers of the last plf_batch_extract spent (max over workers): [0] total, [1] staging copies into pinned
 * memory, [2] waiting for the GPU, [3] unpacking outputs. */
int plf_batch_last_timing(const plf_batch *b, double *out4);
/* The same four figures of ONE worker (= GPU devices[worker]) for the last plf_batch_extract: on an 8-GPU host the first thing to look at when the frames/s do not
 * scale -- staging (host memory bandwidth: every worker copies its frames into its own pinned slots) against gpu_wait. */
int plf_batch_worker_timing(const plf_batch *b, int32_t worker, double *out4);
/* Where worker `worker` (= GPU devices[worker]) runs: the NUMA node of its GPU (-1: unknown / single node) and the number of CPUs its thread was bound to before it
 * allocated its pinned staging slots (0: not bound -- no sysfs view, or PLF_BATCH_NO_AFFINITY=1).  Eight GPUs on a two-socket host: every worker reads its
 * images through its own socket's memory controllers. */
int plf_batch_worker_affinity(const plf_batch *b, int32_t worker, int32_t *numa_node, int32_t *n_cpus);

/* Device memory helpers for host code above this ABI that does not link the HIP runtime itself (the exact-signature adapters of include/plf.hpp
 * stage the Frame / MapPoint / MapLine members the matchers read with them).  With a stream: plf_upload / plf_fill enqueue on it, plf_download
 * enqueues and waits for it.  stream == NULL means synchronous: uploads and fills are complete on return, a download first waits for ALL work
 * of the device (the handles run on their own non-blocking streams, which the null stream does not order with). */
int plf_device_alloc(int32_t device, size_t bytes, void **out);
void plf_device_free(void *p);
int plf_upload(void *dst_device, const void *src_host, size_t bytes, void *stream);
int plf_download(void *dst_host, const void *src_device, size_t bytes, void *stream);
int plf_fill(void *dst_device, int32_t byte_value, size_t bytes, void *stream);

/* Pinned (page-locked, portable across the GPUs of the driver) host memory for frame buffers handed to plf_batch_extract */
int plf_host_alloc(size_t bytes, void **out);
void plf_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* PLF_H */
