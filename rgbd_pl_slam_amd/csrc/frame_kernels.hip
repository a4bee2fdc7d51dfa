// frame_kernels.hip -- RGB-D ingest, Frame tail stages and the frustum projection that feeds the matcher
// (SURVEY.md 8f ranks 1, 2, 5): the callers / data formats on either side of the extractor + matcher hot path.
//   k_rgb_to_gray        Tracking::GrabImageRGBD cv::cvtColor(RGB|BGR -> GRAY)     so@0x522e6  (8U fixed point, yuv_shift 14)
//   k_depth_to_float     imDepth.convertTo(CV_32F, mDepthMapFactor)                so@0x5206d
//   k_frame_tail         Frame::UndistortKeyPoints so@0xf8630 (cv::undistortPoints, 5 iterations, double) +
//                        Frame::ComputeStereoFromRGBD so@0xf6860 (depth lookup, uRight = u - bf/d)
//   k_frustum_points     Frame::isInFrustum(MapPoint*, viewingCosLimit) include/Frame.h:104, so@0xf5190 +
//                        MapPoint::PredictScale so@0x8fc20 (logf / ceilf)
// Thread-per-element kernels, HBM-bound; float/double operation order follows the reference statement by statement.
#include "plf_common.h"

__global__ void __launch_bounds__(256) k_rgb_to_gray(const uint8_t *__restrict__ rgb, ptrdiff_t pitch, ptrdiff_t fstride, int bgr,
                                                     uint8_t *__restrict__ gray, ptrdiff_t gpitch, ptrdiff_t gfstride, int w)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t *p = rgb + (size_t)f * fstride + (size_t)y * pitch + 3 * x;
    const int r = bgr ? p[2] : p[0], g = p[1], b = bgr ? p[0] : p[2];
    gray[(size_t)f * gfstride + (size_t)y * gpitch + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
}

__global__ void __launch_bounds__(256) k_depth_to_float(const uint16_t *__restrict__ d, ptrdiff_t pitch_elems, ptrdiff_t fstride_elems,
                                                        float factor, float *__restrict__ out, int w, int h)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    out[((size_t)f * h + y) * w + x] = (float)d[(size_t)f * fstride_elems + (size_t)y * pitch_elems + x] * factor + 0.0f;
}

// cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) of OpenCV 3.3 for one point: five fixed-point iterations in double (so@0xf8630 calls it)
__device__ __forceinline__ void undistort_point(const plf_camera &cam, float px, float py, float &ux, float &uy)
{
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = cam.k1, k1 = cam.k2, k2 = cam.p1, k3 = cam.p2, k4 = cam.k3;
    double x = px, y = py;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    ux = (float)(xx * ww);
    uy = (float)(yy * ww);
}

// Frame::ComputeStereoFromRGBD (so@0xf6860) for one image point: depth at the truncated DISTORTED position, right coordinate from the undistorted x
__device__ __forceinline__ void stereo_from_depth(const float *__restrict__ depth, int f, int w, int h, float px, float py, float ux, float bf, float &ur, float &dd)
{
    ur = -1.f; dd = -1.f;
    if (!depth) return;
    const int v = (int)py, u = (int)px;
    if (u >= 0 && v >= 0 && u < w && v < h) {
        const float d = depth[((size_t)f * h + v) * w + u];
        if (d > 0) { dd = d; ur = ux - bf / d; }
    }
}

__global__ void __launch_bounds__(256) k_frame_tail(const plf_keypoint *__restrict__ keys, const int *__restrict__ n_dev, int n_host, int stride,
                                                    const float *__restrict__ depth, int w, int h, plf_camera cam,
                                                    plf_keypoint *__restrict__ keys_un, float *__restrict__ uright, float *__restrict__ kdepth)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_dev ? min(n_dev[f], stride) : n_host;
    if (i >= n) return;
    const plf_keypoint kp = keys[(size_t)f * stride + i];
    plf_keypoint ku = kp;
    if (cam.k1 != 0.0f) undistort_point(cam, kp.x, kp.y, ku.x, ku.y);
    keys_un[(size_t)f * stride + i] = ku;
    float ur, dd;
    stereo_from_depth(depth, f, w, h, kp.x, kp.y, ku.x, cam.bf, ur, dd);
    if (uright) uright[(size_t)f * stride + i] = ur;
    if (kdepth) kdepth[(size_t)f * stride + i] = dd;
}

// The line half of the Frame tail: Frame::UndistortKeyLines (include/Frame.h:267 -> mvKeylinesUn, :207) and the end-point fields mvuRightLineStart/End,
// mvDepthLineStart/End (include/Frame.h:208-211).  The fork snapshot declares them without a body, so they are defined the way the point fields are:
// both end points go through undistort_point / stereo_from_depth above (the two routines pinned for key points); every other KeyLine field is copied.
__global__ void __launch_bounds__(256) k_frame_line_tail(const plf_keyline *__restrict__ lines, const int *__restrict__ n_dev, int n_host, int stride,
                                                         const float *__restrict__ depth, int w, int h, plf_camera cam, plf_keyline *__restrict__ lines_un,
                                                         float *__restrict__ ur_s, float *__restrict__ ur_e, float *__restrict__ d_s, float *__restrict__ d_e)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_dev ? min(n_dev[f], stride) : n_host;
    if (i >= n) return;
    const size_t o = (size_t)f * stride + i;
    const plf_keyline kl = lines[o];
    plf_keyline ku = kl;
    if (cam.k1 != 0.0f) {
        undistort_point(cam, kl.startPointX, kl.startPointY, ku.startPointX, ku.startPointY);
        undistort_point(cam, kl.endPointX, kl.endPointY, ku.endPointX, ku.endPointY);
    }
    lines_un[o] = ku;
    float a, b;
    stereo_from_depth(depth, f, w, h, kl.startPointX, kl.startPointY, ku.startPointX, cam.bf, a, b);
    if (ur_s) ur_s[o] = a;
    if (d_s) d_s[o] = b;
    stereo_from_depth(depth, f, w, h, kl.endPointX, kl.endPointY, ku.endPointX, cam.bf, a, b);
    if (ur_e) ur_e[o] = a;
    if (d_e) d_e[o] = b;
}

__global__ void __launch_bounds__(256) k_frustum_points(const float *__restrict__ xw, const float *__restrict__ normal,
                                                        const float *__restrict__ min_dist, const float *__restrict__ max_dist, int m,
                                                        plf_frustum_pose P, plf_camera cam, float4 bounds, float log_scale_factor, int nlevels,
                                                        float cos_limit, float *__restrict__ proj_x, float *__restrict__ proj_y,
                                                        float *__restrict__ proj_xr, int *__restrict__ level, float *__restrict__ view_cos,
                                                        uint8_t *__restrict__ in_view)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    in_view[i] = 0;
    const float p0 = xw[3 * (size_t)i], p1 = xw[3 * (size_t)i + 1], p2 = xw[3 * (size_t)i + 2];
    const float PcX = P.Rcw[0] * p0 + P.Rcw[1] * p1 + P.Rcw[2] * p2 + P.tcw[0];
    const float PcY = P.Rcw[3] * p0 + P.Rcw[4] * p1 + P.Rcw[5] * p2 + P.tcw[1];
    const float PcZ = P.Rcw[6] * p0 + P.Rcw[7] * p1 + P.Rcw[8] * p2 + P.tcw[2];
    if (PcZ < 0.0f) return;
    const float invz = 1.0f / PcZ;
    // the reference binary contracts both projections into FMAs (so@0xf5772, so@0xf57c0: vfmadd213ss)
    const float u = fmaf(cam.fx * PcX, invz, cam.cx);
    const float v = fmaf(cam.fy * PcY, invz, cam.cy);
    if (u < bounds.x || u > bounds.z) return;
    if (v < bounds.y || v > bounds.w) return;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO0 = p0 - P.Ow[0], PO1 = p1 - P.Ow[1], PO2 = p2 - P.Ow[2];
    double s = 0;
    s += (double)PO0 * (double)PO0; s += (double)PO1 * (double)PO1; s += (double)PO2 * (double)PO2;
    const float dist = (float)sqrt(s);
    if (dist < minDistance || dist > maxDistance) return;
    double dot = 0;
    dot += (double)PO0 * (double)normal[3 * (size_t)i]; dot += (double)PO1 * (double)normal[3 * (size_t)i + 1];
    dot += (double)PO2 * (double)normal[3 * (size_t)i + 2];
    const float viewCos = (float)(dot / (double)dist);
    if (viewCos < cos_limit) return;
    const float ratio = max_dist[i] / dist;
    // logf of the reference's libm is (almost always) the correctly rounded value; the double log rounded to float is too
    int nScale = (int)ceilf((float)log((double)ratio) / log_scale_factor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    in_view[i] = 1;
    // mTrackProjXR = u - mbf * invz is contracted as well (so@0xf5dec: vfnmadd132ss)
    proj_x[i] = u; proj_xr[i] = fmaf(-cam.bf, invz, u); proj_y[i] = v; level[i] = nScale; view_cos[i] = viewCos;
}

static int frame_dev_ok(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[plf] no HIP device available: the frame stages have no CPU path\n");
        return PLF_E_HIP;
    }
    if (device < 0 || device >= ndev) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(device));
    return PLF_OK;
}

extern "C" int plf_rgb_to_gray(const uint8_t *rgb, int32_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch, ptrdiff_t frame_stride,
                               int32_t bgr_order, uint8_t *gray, ptrdiff_t gray_pitch, ptrdiff_t gray_frame_stride, int32_t device, void *stream)
{
    if (!rgb || !gray || n_frames < 1 || width < 1 || height < 1 || pitch < 3 * (ptrdiff_t)width || gray_pitch < width) return PLF_E_BADARG;
    int rc = frame_dev_ok(device);
    if (rc != PLF_OK) return rc;
    hipLaunchKernelGGL(k_rgb_to_gray, dim3((width + 255) / 256, height, n_frames), dim3(256), 0, (hipStream_t)stream, rgb, pitch, frame_stride,
                       bgr_order, gray, gray_pitch, gray_frame_stride, width);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_depth_to_float(const uint16_t *depth, int32_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch_elems,
                                  ptrdiff_t frame_stride_elems, float factor, float *out, int32_t device, void *stream)
{
    if (!depth || !out || n_frames < 1 || width < 1 || height < 1 || pitch_elems < width) return PLF_E_BADARG;
    int rc = frame_dev_ok(device);
    if (rc != PLF_OK) return rc;
    hipLaunchKernelGGL(k_depth_to_float, dim3((width + 255) / 256, height, n_frames), dim3(256), 0, (hipStream_t)stream, depth, pitch_elems,
                       frame_stride_elems, factor, out, width, height);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_frame_tail(const plf_keypoint *keys, const int32_t *n_device, int32_t n_host, int32_t n_frames, int32_t kp_stride,
                              const float *depth, int32_t width, int32_t height, const plf_camera *cam, plf_keypoint *keys_un, float *uright,
                              float *kp_depth, int32_t device, void *stream)
{
    if (!keys || !keys_un || !cam || n_frames < 1 || kp_stride < 1 || (!n_device && (n_host < 0 || n_host > kp_stride))) return PLF_E_BADARG;
    int rc = frame_dev_ok(device);
    if (rc != PLF_OK) return rc;
    hipLaunchKernelGGL(k_frame_tail, dim3((kp_stride + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream, keys, n_device, n_host, kp_stride,
                       depth, width, height, *cam, keys_un, uright, kp_depth);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

// Frame::isInFrustum(MapLine*, viewingCosLimit) (include/Frame.h:107; tracking fields include/MapLine.h:113-122): declared without a body in the
// snapshot, so it is the MapPoint routine above applied to a segment -- both end points projected with the same statements (and the same FMA
// contractions) and both required in front of the camera and inside the image bounds, distance / viewing cosine / PredictScale taken at the
// segment's midpoint 0.5 * (S + E).
__device__ __forceinline__ bool frustum_project(const plf_frustum_pose &P, const plf_camera &cam, const float4 &bounds, float p0, float p1, float p2,
                                                float &u, float &v, float &ur)
{
    const float PcX = P.Rcw[0] * p0 + P.Rcw[1] * p1 + P.Rcw[2] * p2 + P.tcw[0];
    const float PcY = P.Rcw[3] * p0 + P.Rcw[4] * p1 + P.Rcw[5] * p2 + P.tcw[1];
    const float PcZ = P.Rcw[6] * p0 + P.Rcw[7] * p1 + P.Rcw[8] * p2 + P.tcw[2];
    if (PcZ < 0.0f) return false;
    const float invz = 1.0f / PcZ;
    u = fmaf(cam.fx * PcX, invz, cam.cx);
    v = fmaf(cam.fy * PcY, invz, cam.cy);
    if (u < bounds.x || u > bounds.z) return false;
    if (v < bounds.y || v > bounds.w) return false;
    ur = fmaf(-cam.bf, invz, u);
    return true;
}

__global__ void __launch_bounds__(256) k_frustum_lines(const float *__restrict__ xw, const float *__restrict__ normal,
                                                       const float *__restrict__ min_dist, const float *__restrict__ max_dist, int m,
                                                       plf_frustum_pose P, plf_camera cam, float4 bounds, float log_scale_factor, int nlevels,
                                                       float cos_limit, float *__restrict__ x1, float *__restrict__ y1, float *__restrict__ x1r,
                                                       float *__restrict__ x2, float *__restrict__ y2, float *__restrict__ x2r,
                                                       int *__restrict__ level, float *__restrict__ view_cos, uint8_t *__restrict__ in_view)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    in_view[i] = 0;
    const float *p = xw + 6 * (size_t)i;
    const float s0 = p[0], s1 = p[1], s2 = p[2], e0 = p[3], e1 = p[4], e2 = p[5];
    float u1, v1, r1, u2, v2, r2;
    if (!frustum_project(P, cam, bounds, s0, s1, s2, u1, v1, r1)) return;
    if (!frustum_project(P, cam, bounds, e0, e1, e2, u2, v2, r2)) return;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    const float PO0 = 0.5f * (s0 + e0) - P.Ow[0], PO1 = 0.5f * (s1 + e1) - P.Ow[1], PO2 = 0.5f * (s2 + e2) - P.Ow[2];
    double s = 0;
    s += (double)PO0 * (double)PO0; s += (double)PO1 * (double)PO1; s += (double)PO2 * (double)PO2;
    const float dist = (float)sqrt(s);
    if (dist < minDistance || dist > maxDistance) return;
    double dot = 0;
    dot += (double)PO0 * (double)normal[3 * (size_t)i]; dot += (double)PO1 * (double)normal[3 * (size_t)i + 1];
    dot += (double)PO2 * (double)normal[3 * (size_t)i + 2];
    const float viewCos = (float)(dot / (double)dist);
    if (viewCos < cos_limit) return;
    const float ratio = max_dist[i] / dist;
    int nScale = (int)ceilf((float)log((double)ratio) / log_scale_factor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    in_view[i] = 1;
    x1[i] = u1; y1[i] = v1; x2[i] = u2; y2[i] = v2; level[i] = nScale; view_cos[i] = viewCos;
    if (x1r) x1r[i] = r1;
    if (x2r) x2r[i] = r2;
}

extern "C" int plf_frame_line_tail(const plf_keyline *lines, const int32_t *n_device, int32_t n_host, int32_t n_frames, int32_t line_stride,
                                   const float *depth, int32_t width, int32_t height, const plf_camera *cam, plf_keyline *lines_un,
                                   float *uright_start, float *uright_end, float *depth_start, float *depth_end, int32_t device, void *stream)
{
    if (!lines || !lines_un || !cam || n_frames < 1 || line_stride < 1 || (!n_device && (n_host < 0 || n_host > line_stride))) return PLF_E_BADARG;
    if (depth && (width < 1 || height < 1)) return PLF_E_BADARG;
    int rc = frame_dev_ok(device);
    if (rc != PLF_OK) return rc;
    hipLaunchKernelGGL(k_frame_line_tail, dim3((line_stride + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream, lines, n_device, n_host,
                       line_stride, depth, width, height, *cam, lines_un, uright_start, uright_end, depth_start, depth_end);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_frustum_points(const float *world_pos, const float *normal, const float *min_distance, const float *max_distance, int32_t m,
                                  const plf_frustum_pose *pose, const plf_camera *cam, float min_x, float min_y, float max_x, float max_y,
                                  float log_scale_factor, int32_t nlevels, float viewing_cos_limit, float *proj_x, float *proj_y, float *proj_xr,
                                  int32_t *level, float *view_cos, uint8_t *in_view, int32_t device, void *stream)
{
    if (!world_pos || !normal || !min_distance || !max_distance || !pose || !cam || m < 1 || !proj_x || !proj_y || !proj_xr || !level ||
        !view_cos || !in_view || nlevels < 1)
        return PLF_E_BADARG;
    int rc = frame_dev_ok(device);
    if (rc != PLF_OK) return rc;
    hipLaunchKernelGGL(k_frustum_points, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, world_pos, normal, min_distance, max_distance,
                       m, *pose, *cam, make_float4(min_x, min_y, max_x, max_y), log_scale_factor, nlevels, viewing_cos_limit, proj_x, proj_y,
                       proj_xr, level, view_cos, in_view);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}

extern "C" int plf_frustum_lines(const float *world_pos, const float *normal, const float *min_distance, const float *max_distance, int32_t m,
                                 const plf_frustum_pose *pose, const plf_camera *cam, float min_x, float min_y, float max_x, float max_y,
                                 float log_scale_factor, int32_t nlevels, float viewing_cos_limit, float *x1, float *y1, float *x1r, float *x2,
                                 float *y2, float *x2r, int32_t *level, float *view_cos, uint8_t *in_view, int32_t device, void *stream)
{
    if (!world_pos || !normal || !min_distance || !max_distance || !pose || !cam || m < 1 || !x1 || !y1 || !x2 || !y2 || !level || !view_cos ||
        !in_view || nlevels < 1)
        return PLF_E_BADARG;
    int rc = frame_dev_ok(device);
    if (rc != PLF_OK) return rc;
    hipLaunchKernelGGL(k_frustum_lines, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, world_pos, normal, min_distance, max_distance, m,
                       *pose, *cam, make_float4(min_x, min_y, max_x, max_y), log_scale_factor, nlevels, viewing_cos_limit, x1, y1, x1r, x2, y2, x2r,
                       level, view_cos, in_view);
    PLF_HIP_TRY(hipGetLastError());
    return PLF_OK;
}
