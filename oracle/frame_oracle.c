/*
 * frame_oracle.c -- CPU restatement of the Frame "tail" stages next to the extractor and of the projection that
 * feeds the matcher.  TEST INFRASTRUCTURE ONLY (same rules as orb_oracle.c).
 *
 *   orc_rgb_to_gray            Tracking::GrabImageRGBD cv::cvtColor(RGB|BGR -> GRAY)      so@0x522e6  [UPSTREAM OpenCV 3.3 color.cpp, 8U fixed point]
 *   orc_depth_to_float         imDepth.convertTo(CV_32F, mDepthMapFactor)                 so@0x5206d  [UPSTREAM cvtScale 16u->32f, float arithmetic]
 *   orc_undistort_keypoints    Frame::UndistortKeyPoints                                  so@0xf8630  [UPSTREAM cv::undistortPoints, 5 iterations, double]
 *   orc_stereo_from_rgbd       Frame::ComputeStereoFromRGBD                               so@0xf6860
 *   orc_is_in_frustum          Frame::isInFrustum(MapPoint*, float) include/Frame.h:104   so@0xf5190, MapPoint::PredictScale so@0x8fc20 (logf, ceilf)
 * The ORB-SLAM glue follows the disassembly sites above; the OpenCV internals are "parity unpinned".
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include "oracle.h"

/* cv::cvtColor 8U: Y = (R*4899 + G*9617 + B*1868 + 8192) >> 14 (yuv_shift = 14) */
void orc_rgb_to_gray(const uint8_t *rgb, int w, int h, ptrdiff_t pitch, int bgr_order, uint8_t *gray, ptrdiff_t gpitch)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = rgb + (ptrdiff_t)y * pitch + 3 * x;
            const int r = bgr_order ? p[2] : p[0], g = p[1], b = bgr_order ? p[0] : p[2];
            gray[(ptrdiff_t)y * gpitch + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
        }
}

void orc_depth_to_float(const uint16_t *d, int w, int h, ptrdiff_t pitch_elems, float factor, float *out)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = (float)d[(ptrdiff_t)y * pitch_elems + x] * factor + 0.0f;
}

/* cam: fx, fy, cx, cy, k1, k2, p1, p2, k3 (floats as stored in mK / mDistCoef) */
void orc_undistort_keypoints(const orc_keypoint *keys, int n, const float *cam, orc_keypoint *keys_un)
{
    const float k1 = cam[4];
    if (k1 == 0.0f) { /* mDistCoef.at<float>(0)==0.0: mvKeysUn = mvKeys */
        for (int i = 0; i < n; i++) keys_un[i] = keys[i];
        return;
    }
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k[5] = {cam[4], cam[5], cam[6], cam[7], cam[8]};
    for (int i = 0; i < n; i++) {
        double x = keys[i].x, y = keys[i].y;
        double x0 = x = (x - cx) * ifx;
        double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0 * r2 + 0 * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* P = K, R = I: RR = K */
        const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
        keys_un[i] = keys[i];
        keys_un[i].x = (float)(xx * ww);
        keys_un[i].y = (float)(yy * ww);
    }
}

void orc_stereo_from_rgbd(const orc_keypoint *keys, const orc_keypoint *keys_un, int n, const float *depth, int w, int h, float bf,
                          float *uright, float *kdepth)
{
    for (int i = 0; i < n; i++) {
        uright[i] = -1.f; kdepth[i] = -1.f;
        const int v = (int)keys[i].y, u = (int)keys[i].x; /* at<float>(float, float): truncation */
        if (u < 0 || v < 0 || u >= w || v >= h) continue;   /* reference reads out of bounds here */
        const float d = depth[(size_t)v * w + u];
        if (d > 0) { kdepth[i] = d; uright[i] = keys_un[i].x - bf / d; }
    }
}

/* Line half of the Frame tail (include/Frame.h:207-211, :267): no body in the fork snapshot -- defined as the two routines above applied to both end
 * points of every KeyLine (so the arithmetic is the one pinned for key points); all other KeyLine fields are copied. */
void orc_line_tail(const orc_keyline *kl, int n, const float *cam, const float *depth, int w, int h, float bf, orc_keyline *kl_un,
                   float *ur_s, float *ur_e, float *d_s, float *d_e)
{
    for (int i = 0; i < n; i++) {
        orc_keypoint in[2] = {{0}}, un[2];
        in[0].x = kl[i].startPointX; in[0].y = kl[i].startPointY;
        in[1].x = kl[i].endPointX; in[1].y = kl[i].endPointY;
        orc_undistort_keypoints(in, 2, cam, un);
        kl_un[i] = kl[i];
        kl_un[i].startPointX = un[0].x; kl_un[i].startPointY = un[0].y;
        kl_un[i].endPointX = un[1].x; kl_un[i].endPointY = un[1].y;
        float ur[2] = {-1.f, -1.f}, dd[2] = {-1.f, -1.f};
        if (depth) orc_stereo_from_rgbd(in, un, 2, depth, w, h, bf, ur, dd);
        ur_s[i] = ur[0]; ur_e[i] = ur[1]; d_s[i] = dd[0]; d_e[i] = dd[1];
    }
}

/* pose: Rcw (9, row-major), tcw (3), Ow (3); cam: fx, fy, cx, cy; bounds: minx, miny, maxx, maxy */
void orc_is_in_frustum(const float *xw, const float *normal, const float *min_dist, const float *max_dist, int m, const float *Rcw,
                       const float *tcw, const float *Ow, const float *cam, const float *bounds, float bf, float log_scale_factor, int nlevels,
                       float cos_limit, float *proj_x, float *proj_y, float *proj_xr, int32_t *level, float *view_cos, uint8_t *in_view)
{
    for (int i = 0; i < m; i++) {
        in_view[i] = 0;
        const float *P = xw + 3 * (size_t)i;
        const float PcX = Rcw[0] * P[0] + Rcw[1] * P[1] + Rcw[2] * P[2] + tcw[0];
        const float PcY = Rcw[3] * P[0] + Rcw[4] * P[1] + Rcw[5] * P[2] + tcw[1];
        const float PcZ = Rcw[6] * P[0] + Rcw[7] * P[1] + Rcw[8] * P[2] + tcw[2];
        if (PcZ < 0.0f) continue;
        const float invz = 1.0f / PcZ;
        /* the reference binary contracts both projections into FMAs (so@0xf5772, so@0xf57c0: vfmadd213ss) */
        const float u = fmaf(cam[0] * PcX, invz, cam[2]);
        const float v = fmaf(cam[1] * PcY, invz, cam[3]);
        if (u < bounds[0] || u > bounds[2]) continue;
        if (v < bounds[1] || v > bounds[3]) continue;
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)PO[k] * (double)PO[k];   /* cv::norm(CV_32F): double accumulation */
        const float dist = (float)sqrt(s);
        if (dist < minDistance || dist > maxDistance) continue;
        const float *Pn = normal + 3 * (size_t)i;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn[k]; /* Mat::dot(CV_32F): double accumulation */
        const float viewCos = (float)(dot / (double)dist);
        if (viewCos < cos_limit) continue;
        const float ratio = max_dist[i] / dist;
        int nScale = (int)ceilf(logf(ratio) / log_scale_factor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= nlevels) nScale = nlevels - 1;
        in_view[i] = 1;
        /* mTrackProjXR = u - mbf*invz is contracted too (so@0xf5dec: vfnmadd132ss) */
        proj_x[i] = u; proj_xr[i] = fmaf(-bf, invz, u); proj_y[i] = v; level[i] = nScale; view_cos[i] = viewCos;
    }
}

/* Frame::isInFrustum(MapLine*, viewingCosLimit) (include/Frame.h:107): no body in the snapshot -- the point routine above on both end points of
 * the segment (xw: m x 6, start then end), distance / viewing cosine / PredictScale at the midpoint.  out: x1 y1 x1r x2 y2 x2r. */
static int frustum_project1(const float *P, const float *Rcw, const float *tcw, const float *cam, const float *bounds, float bf, float *u, float *v,
                            float *ur)
{
    const float PcX = Rcw[0] * P[0] + Rcw[1] * P[1] + Rcw[2] * P[2] + tcw[0];
    const float PcY = Rcw[3] * P[0] + Rcw[4] * P[1] + Rcw[5] * P[2] + tcw[1];
    const float PcZ = Rcw[6] * P[0] + Rcw[7] * P[1] + Rcw[8] * P[2] + tcw[2];
    if (PcZ < 0.0f) return 0;
    const float invz = 1.0f / PcZ;
    *u = fmaf(cam[0] * PcX, invz, cam[2]);
    *v = fmaf(cam[1] * PcY, invz, cam[3]);
    if (*u < bounds[0] || *u > bounds[2]) return 0;
    if (*v < bounds[1] || *v > bounds[3]) return 0;
    *ur = fmaf(-bf, invz, *u);
    return 1;
}

void orc_is_in_frustum_line(const float *xw, const float *normal, const float *min_dist, const float *max_dist, int m, const float *Rcw,
                            const float *tcw, const float *Ow, const float *cam, const float *bounds, float bf, float log_scale_factor,
                            int nlevels, float cos_limit, float *out6, int32_t *level, float *view_cos, uint8_t *in_view)
{
    for (int i = 0; i < m; i++) {
        in_view[i] = 0;
        const float *S = xw + 6 * (size_t)i, *E = S + 3;
        float u1, v1, r1, u2, v2, r2;
        if (!frustum_project1(S, Rcw, tcw, cam, bounds, bf, &u1, &v1, &r1)) continue;
        if (!frustum_project1(E, Rcw, tcw, cam, bounds, bf, &u2, &v2, &r2)) continue;
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
        float PO[3];
        for (int k = 0; k < 3; k++) PO[k] = 0.5f * (S[k] + E[k]) - Ow[k];
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)PO[k] * (double)PO[k];
        const float dist = (float)sqrt(s);
        if (dist < minDistance || dist > maxDistance) continue;
        const float *Pn = normal + 3 * (size_t)i;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn[k];
        const float viewCos = (float)(dot / (double)dist);
        if (viewCos < cos_limit) continue;
        const float ratio = max_dist[i] / dist;
        int nScale = (int)ceilf(logf(ratio) / log_scale_factor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= nlevels) nScale = nlevels - 1;
        in_view[i] = 1;
        float *o = out6 + 6 * (size_t)i;
        o[0] = u1; o[1] = v1; o[2] = r1; o[3] = u2; o[4] = v2; o[5] = r2;
        level[i] = nScale; view_cos[i] = viewCos;
    }
}
