/* oracle.h -- declarations of the CPU oracle (TEST INFRASTRUCTURE ONLY; see orb_oracle.c header). */
#ifndef PLF_ORACLE_H
#define PLF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bit-compatible with cv::KeyPoint (28 bytes) */
typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint;

/* field order of cv::line_descriptor::KeyLine (17 x 4 bytes) */
typedef struct {
    float angle; int class_id; int octave; float pt_x, pt_y; float response; float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int numOfPixels;
} orc_keyline;

typedef struct {
    int want_planes;
    int ncand[16], nsel[16], ncells[16], lw[16], lh[16];
    uint8_t *pyr[16];  /* padded planes (malloc'd when want_planes) */
    uint8_t *blur[16]; /* blurred interiors */
    orc_keypoint *level_kps[16]; /* per-level keypoints, level coords, with angle (nsel[l] entries) */
} orc_orb_debug;

void orc_orb_tables(int nfeatures, float scaleFactor, int nlevels, float *scale, float *inv, float *sigma2,
                    float *invsigma2, int *perLevel, int *umax);
void orc_level_size(int w, int h, float inv, int *lw, int *lh);
void orc_resize_linear_8u(const uint8_t *src, int sw, int sh, ptrdiff_t spitch, uint8_t *dst, int dw, int dh,
                          ptrdiff_t dpitch);
void orc_border_reflect101(uint8_t *plane, int w, int h, ptrdiff_t pitch, int b);
int orc_compute_pyramid(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nlevels, const float *inv,
                        uint8_t **planes, int *lw, int *lh);
int orc_fast9_16(const uint8_t *img, int cols, int rows, ptrdiff_t pitch, int threshold, orc_keypoint *out, int cap);
int orc_cell_rects(int cols, int rows, int *rects, int cap, int *wCell_out, int *hCell_out);
int orc_fast_cells(const uint8_t *img, int cols, int rows, ptrdiff_t pitch, int iniTh, int minTh,
                   orc_keypoint *out, int cap, int *ncells_out);
int orc_distribute_octree(const orc_keypoint *kps, int nk, int minX, int maxX, int minY, int maxY, int N,
                          orc_keypoint *out, int cap);
float orc_fast_atan2(float y, float x);
void orc_ic_moments(const uint8_t *img, ptrdiff_t pitch, int cx, int cy, const int *umax, int *m01, int *m10);
float orc_ic_angle(const uint8_t *img, ptrdiff_t pitch, float x, float y, const int *umax);
void orc_gauss_taps_8u(int ksize, double sigma, int *taps);
void orc_gaussian_blur7_8u(const uint8_t *src, ptrdiff_t spitch, uint8_t *dst, ptrdiff_t dpitch, int w, int h);
/* the same 8-bit fixed-point separable path for any odd ksize <= 33 (ksize 5, sigma 1: the LBD pre-blur) */
void orc_gaussian_blur_8u(const uint8_t *src, ptrdiff_t spitch, uint8_t *dst, ptrdiff_t dpitch, int w, int h, int ksize, double sigma);
void orc_brief_descriptor(const uint8_t *blur, ptrdiff_t pitch, float x, float y, float angle_deg, uint8_t *desc);
int orc_orb_extract(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nfeatures, float scaleFactor,
                    int nlevels, int iniTh, int minTh, orc_keypoint *kps, uint8_t *desc, int cap,
                    orc_orb_debug *dbg);
void orc_sincosf_glibc(float y, float *sinp, float *cosp);
long orc_sincosf_selftest(long count);
void orc_free(void *p);

/* ---- matchers (match_oracle.c) */
int orc_hamming256(const uint8_t *a, const uint8_t *b);

typedef struct {
    int n;                    /* keypoints */
    const float *ux, *uy;     /* mvKeysUn[i].pt */
    const int *octave;        /* mvKeysUn[i].octave */
    const float *uright;      /* mvuRight (<=0: none) */
    const uint8_t *desc;      /* n x 32 */
    const float *angle;       /* mvKeys[i].angle (deg), used by the last-frame overload only */
    float minx, miny, maxx, maxy; /* mnMinX.. image bounds */
    float grid_inv_w, grid_inv_h; /* mfGridElementWidthInv / HeightInv */
    const float *scale_factors;   /* mvScaleFactors */
    int nlevels;
} orc_frame;

typedef struct {
    int m;
    const float *proj_x, *proj_y, *proj_xr; /* mTrackProjX/Y/XR */
    const int *level;                       /* mnTrackScaleLevel */
    const float *view_cos;                  /* mTrackViewCos */
    const uint8_t *in_view;                 /* mbTrackInView && !isBad() */
    const uint8_t *desc;                    /* m x 32 */
    const uint8_t *obs_positive;            /* Observations()>0 per map point; NULL = all */
} orc_mappoints;

typedef struct { /* LastFrame view for the motion-model overload (8a-12) */
    int n;
    const uint8_t *has_mp, *outlier; /* mvpMapPoints[i]!=NULL, mvbOutlier[i] */
    const float *xw;                 /* n x 3 world positions */
    const int *octave;               /* mvKeys[i].octave */
    const float *angle;              /* mvKeysUn[i].angle */
    const uint8_t *mp_desc;          /* n x 32, pMP->GetDescriptor() */
    const uint8_t *obs_positive;     /* pMP->Observations() > 0; NULL = all.  0 = a temporal point of localisation mode (Tracking::UpdateLastFrame,
                                        include/Tracking.h:152): a key point it was assigned to stays available to later last-frame points */
} orc_lastframe;

typedef struct { /* current frame, lines */
    int n;
    const float *pt_x, *pt_y, *angle; /* mvKeylinesUn[i].pt / .angle */
    const int *octave;
    const uint8_t *desc;              /* n x 32 (mLdesc) */
    const float *scale_factors;
} orc_lineframe;

typedef struct {
    int m;
    const float *x1, *y1, *x2, *y2; /* mTrackProjX1.. */
    const int *level;
    const float *view_cos;
    const uint8_t *in_view;
    const uint8_t *desc;
} orc_maplines;

int orc_features_in_area(const orc_frame *F, float x, float y, float r, int minLevel, int maxLevel,
                         int *out, int cap);
/* match_of_kp: in: >=0 where the keypoint already holds a map point with Observations()>0,
 * -1 otherwise (use -2 for "holds a map point with 0 observations" if needed).  out: index of
 * the matched map point (>= 0 new matches are written as map point index + 0). */
int orc_search_by_projection_map(const orc_frame *F, const orc_mappoints *MP, float th, float nnratio,
                                 int32_t *match_of_kp);
void orc_assign_grid(const orc_frame *F, int32_t *cell_start, int32_t *cell_idx);
float orc_radius_by_viewing_cos(float viewCos);
void orc_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3);
int orc_search_by_projection_last(const orc_frame *Cur, const orc_lastframe *Last, const float *Rcw,
                                  const float *tcw, const float *Rlw, const float *tlw, float fx, float fy,
                                  float cx, float cy, float bf, float b, float th, int bMono, int checkOri,
                                  int32_t *match_of_kp);
void orc_line_mad(const int32_t *dist, int n, double *nn_mad, double *nn12_mad);
double orc_line_segment_overlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj);   /* LineSegment::LineSegmentOverlap, include/ExtractLineSegment.h:47 */
int orc_match_lines_knn(const uint8_t *last_desc, int nlast, const uint8_t *cur_desc, int ncur,
                        const uint8_t *last_has_mapline, int32_t *match_of_line);
int orc_lines_search_for_triangulation(const uint8_t *desc1, int n1, const uint8_t *desc2, int n2, const uint8_t *has_ml1, const uint8_t *has_ml2,
                                       const uint8_t *stereo1, const uint8_t *stereo2, int only_stereo, double mad_factor, int32_t *match12);
int orc_lines_fuse(const uint8_t *kf_desc, int n_kf, const uint8_t *ml_desc, const uint8_t *valid, int m, int32_t *best_idx);
int orc_lines_in_area(const orc_lineframe *F, float x1, float y1, float x2, float y2, float r, int minLevel,
                      int maxLevel, int *out, int cap);
int orc_search_by_projection_lines(const orc_lineframe *F, const orc_maplines *ML, float th, float nnratio,
                                   int32_t *match_of_line);
int orc_search_by_projection_reloc(const orc_frame *Cur, const orc_lastframe *KF, const float *min_dist, const float *max_dist,
                                   const float *Rcw, const float *tcw, float fx, float fy, float cx, float cy, float log_scale_factor,
                                   float th, int ORBdist, int checkOri, int32_t *match_of_kp);
int orc_search_by_bow(int n_kf, int n_f, const uint8_t *kf_desc, const uint8_t *f_desc, const float *kf_angle, const float *f_angle,
                      const uint8_t *kf_has_mp, int kf_nodes, const uint32_t *kf_node_id, const int32_t *kf_node_start,
                      const int32_t *kf_feat, int f_nodes, const uint32_t *f_node_id, const int32_t *f_node_start,
                      const int32_t *f_feat, float nnratio, int checkOri, int32_t *match_of_f);
int orc_search_by_bow_kf(int n1, int n2, const uint8_t *desc1, const uint8_t *desc2, const float *angle1, const float *angle2,
                         const uint8_t *has_mp1, const uint8_t *has_mp2, int nodes1, const uint32_t *node_id1, const int32_t *node_start1,
                         const int32_t *feat1, int nodes2, const uint32_t *node_id2, const int32_t *node_start2, const int32_t *feat2,
                         float nnratio, int checkOri, int32_t *match12);
typedef struct { /* map points handed to Fuse / the Sim3 searches */
    int m;
    const float *xw;                  /* m x 3: GetWorldPos() */
    const float *normal;              /* m x 3: GetNormal() */
    const float *min_dist, *max_dist; /* mfMinDistance, mfMaxDistance (the members, not the 0.8 / 1.2 scaled getters) */
    const uint8_t *desc;              /* m x 32: GetDescriptor() */
    const uint8_t *valid;             /* non-null, !isBad(), not already in the keyframe / found set */
} orc_points3d;
typedef struct { /* KeyFrame pose and intrinsics */
    float Rcw[9], tcw[3], Ow[3];      /* GetRotation(), GetTranslation(), GetCameraCenter() */
    float fx, fy, cx, cy, bf, log_scale_factor;
    const float *inv_level_sigma2;    /* mvInvLevelSigma2 */
} orc_kf_pose;
int orc_fuse(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, float th, int32_t *best_idx, int32_t *best_dist);
void orc_sim3_decompose(const float *Scw, float *Rcw, float *tcw, float *Ow);
int orc_fuse_sim3(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, float th, int32_t *best_idx);
int orc_search_by_projection_sim3(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, int th, int32_t *match_of_kp);
int orc_search_by_sim3(const orc_frame *KF1, const orc_frame *KF2, const orc_kf_pose *C1, const orc_kf_pose *C2, float s12, const float *R12,
                       const float *t12, float th, const orc_points3d *P1, const orc_points3d *P2, int32_t *match12);
int orc_check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float *F12, float level_sigma2);
void orc_epipole(const float *Cw, const float *R2w, const float *t2w, float fx, float fy, float cx, float cy, float *ex, float *ey);
int orc_search_for_triangulation(int n1, int n2, const float *x1, const float *y1, const float *angle1, const float *uright1, const uint8_t *desc1,
                                 const uint8_t *has_mp1, const float *x2, const float *y2, const float *angle2, const int32_t *octave2,
                                 const float *uright2, const uint8_t *desc2, const uint8_t *has_mp2, int nodes1, const uint32_t *node_id1,
                                 const int32_t *node_start1, const int32_t *feat1, int nodes2, const uint32_t *node_id2, const int32_t *node_start2,
                                 const int32_t *feat2, const float *F12, float ex, float ey, const float *scale_factors2, const float *level_sigma2_2,
                                 int bOnlyStereo, int checkOri, int32_t *match12);
int orc_knn2_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx /*nq x 2*/,
                     int32_t *dist /*nq x 2*/);

/* ---- lines (lsd_oracle.c, lbd_oracle.c) */
#define ORC_LSD_SEED_RASTER 0 /* OpenCV 3.0-3.3 behaviour (list[i] iterated by index) */
#define ORC_LSD_SEED_BINNED 1 /* published LSD order: gradient bins descending */

typedef struct {
    int want_maps;
    int sw, sh, nregions, min_reg_size;
    double max_grad;
    double *scaled, *angles, *modgrad; /* malloc'd sw*sh maps when want_maps */
    uint8_t *used;
} orc_lsd_debug;

void orc_gauss_kernel_f64(int n, double sigma, double *k);
int orc_lsd_detect(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int seed_order, float *lines, int cap,
                   orc_lsd_debug *dbg);
void orc_lsd_band_speculation_halo(int rows);
int orc_lsd_band_speculation(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nbands, long *stats /*8*/);
int orc_lsd_band_rounds(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nbands, long *stats /*8*/);
int orc_keylines_from_segments(const float *lines, int n, int w, int h, orc_keyline *out);
void orc_sobel3_16s(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int16_t *dxImg, int16_t *dyImg);
void orc_lbd_gauss_coefs(double *gaussCoefL, double *gaussCoefG);
void orc_lbd_compute(const uint8_t *gray, int w, int h, ptrdiff_t pitch, const orc_keyline *kl, int n, uint8_t *desc,
                     float *fdesc);
int orc_line_extract(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nkeep, int seed_order, orc_keyline *out,
                     uint8_t *desc, double *lineeq, int cap, int *ndetected);
/* version-dependent choice: what BinaryDescriptor's Sobel reads.  The plain functions above use ORC_LBD_BLURRED. */
#define ORC_LBD_BLURRED 0 /* GaussianBlur(5x5, sigma 1) of octave 0 first (opencv_contrib 3.3 BinaryDescriptor::computeGaussianPyramid) */
#define ORC_LBD_RAW 1     /* the image as handed in */
void orc_lbd_compute_ex(const uint8_t *gray, int w, int h, ptrdiff_t pitch, const orc_keyline *kl, int n, uint8_t *desc,
                        float *fdesc, int sobel_input);
int orc_line_extract_ex(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nkeep, int seed_order, orc_keyline *out,
                        uint8_t *desc, double *lineeq, int cap, int *ndetected, int sobel_input);

/* ---- per-stage wall-clock split of the CPU baseline (bench_oracle.c).  Thread-local accumulators, filled only
 * while orc_stage_timing is non-zero; the timed functions are otherwise untouched. */
enum { ORC_ST_PYRAMID = 0, ORC_ST_FAST, ORC_ST_OCTREE, ORC_ST_ORIENT, ORC_ST_BLUR, ORC_ST_BRIEF, ORC_ST_LSD, ORC_ST_LBD, ORC_ST_MATCH_POINTS,
       ORC_ST_MATCH_LINES, ORC_NSTAGES };
extern int orc_stage_timing;
extern __thread double orc_stage_s[ORC_NSTAGES];
double orc_now_s(void);
#define ORC_T0(var) const double var = orc_stage_timing ? orc_now_s() : 0.0
#define ORC_T1(var, stage) do { if (orc_stage_timing) orc_stage_s[stage] += orc_now_s() - (var); } while (0)

/* ---- Frame tail / ingest / frustum (frame_oracle.c) */
void orc_rgb_to_gray(const uint8_t *rgb, int w, int h, ptrdiff_t pitch, int bgr_order, uint8_t *gray, ptrdiff_t gpitch);
void orc_depth_to_float(const uint16_t *d, int w, int h, ptrdiff_t pitch_elems, float factor, float *out);
void orc_undistort_keypoints(const orc_keypoint *keys, int n, const float *cam, orc_keypoint *keys_un);
void orc_stereo_from_rgbd(const orc_keypoint *keys, const orc_keypoint *keys_un, int n, const float *depth, int w, int h, float bf,
                          float *uright, float *kdepth);
void orc_line_tail(const orc_keyline *kl, int n, const float *cam, const float *depth, int w, int h, float bf, orc_keyline *kl_un,
                   float *ur_s, float *ur_e, float *d_s, float *d_e);
void orc_is_in_frustum(const float *xw, const float *normal, const float *min_dist, const float *max_dist, int m, const float *Rcw,
                       const float *tcw, const float *Ow, const float *cam, const float *bounds, float bf, float log_scale_factor, int nlevels,
                       float cos_limit, float *proj_x, float *proj_y, float *proj_xr, int32_t *level, float *view_cos, uint8_t *in_view);
void orc_is_in_frustum_line(const float *xw, const float *normal, const float *min_dist, const float *max_dist, int m, const float *Rcw,
                            const float *tcw, const float *Ow, const float *cam, const float *bounds, float bf, float log_scale_factor,
                            int nlevels, float cos_limit, float *out6, int32_t *level, float *view_cos, uint8_t *in_view);

/* ---- CPU baseline driver (bench_oracle.c) */
double orc_frontend_throughput(const uint8_t *imgs, int n_distinct, int w, int h, int n_frames, int threads, int nfeatures, int nlines,
                               const orc_mappoints *MP, const orc_maplines *ML, float th, float nnratio, long *checksum);

#ifdef __cplusplus
}
#endif
#endif
