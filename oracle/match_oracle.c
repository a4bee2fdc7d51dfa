/*
 * match_oracle.c -- CPU restatement of the reference Hamming matchers.  TEST INFRASTRUCTURE ONLY
 * (same rules as orb_oracle.c: only tests/, smoke() and bench.py's cpu_baseline may use it).
 *
 *   orc_hamming256                 ORBmatcher::DescriptorDistance  include/ORBmatcher.h:44, so@0x79d20,
 *                                  source form Thirdparty/DBoW2/DBoW2/FORB.cpp:82-102; LSDmatcher.h:43   (8a-9)
 *   orc_assign_grid / orc_features_in_area
 *                                  Frame::AssignFeaturesToGrid so@0xf9120, Frame::GetFeaturesInArea
 *                                  include/Frame.h:113, so@0xfbc60                                       (8a-11)
 *   orc_search_by_projection_map   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)
 *                                  include/ORBmatcher.h:61, so@0x79f10                                   (8a-10)
 *   orc_search_by_projection_last  ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)
 *                                  include/ORBmatcher.h:78, so@0x80d00                                   (8a-12)
 *   orc_radius_by_viewing_cos      ORBmatcher::RadiusByViewingCos so@0x79b60
 *   orc_three_maxima               ORBmatcher::ComputeThreeMaxima so@0x823eb
 *   orc_knn2_hamming               cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) [UPSTREAM: OpenCV 3.3 batchDistance]
 *   orc_line_mad / orc_match_lines_knn   Frame::lineDescriptorMAD include/Frame.h:75 + LSDmatcher::SearchByProjection
 *                                  (Frame&, const Frame&) include/LSDmatcher.h:32 [UPSTREAM body, PL-SLAM family] (8a-13)
 *   orc_lines_in_area / orc_search_by_projection_lines
 *                                  Frame::GetLinesInArea include/Frame.h:116 + LSDmatcher::SearchByProjection
 *                                  (Frame&, vector<MapLine*>&, th) include/LSDmatcher.h:40 [UPSTREAM body]   (8a-14)
 *
 * PINNED against the reference binary (tests/golden/ref_matcher.json): Hamming, RadiusByViewingCos,
 * ComputeThreeMaxima, TH_LOW/TH_HIGH/HISTO_LENGTH.  The SearchByProjection bodies follow the full
 * disassembly listing summarised in SURVEY.md 8a-10/12; the line matchers are "parity unpinned".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define GRID_COLS 64
#define GRID_ROWS 48
#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30

int orc_hamming256(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

float orc_radius_by_viewing_cos(float viewCos) { return ((double)viewCos > 0.998) ? 2.5f : 4.0f; }

/* ---------------------------------------------------------------- grid */
typedef struct { int *start; int *idx; } grid_t; /* CSR over cell = ix*GRID_ROWS+iy */

static void grid_build(const orc_frame *F, grid_t *g)
{
    const int ncell = GRID_COLS * GRID_ROWS;
    int *cell = (int *)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    g->start = (int *)calloc(ncell + 1, sizeof(int));
    g->idx = (int *)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    for (int i = 0; i < F->n; i++) {
        int px = (int)roundf((F->ux[i] - F->minx) * F->grid_inv_w);
        int py = (int)roundf((F->uy[i] - F->miny) * F->grid_inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) { cell[i] = -1; continue; }
        cell[i] = px * GRID_ROWS + py;
        g->start[cell[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) g->start[c + 1] += g->start[c];
    int *fill = (int *)calloc(ncell, sizeof(int));
    for (int i = 0; i < F->n; i++) /* insertion order inside a cell = keypoint order */
        if (cell[i] >= 0) g->idx[g->start[cell[i]] + fill[cell[i]]++] = i;
    free(fill); free(cell);
}
static void grid_free(grid_t *g) { free(g->start); free(g->idx); }

/* exported CSR builder so tests can hand the same grid to the product */
void orc_assign_grid(const orc_frame *F, int32_t *cell_start /*64*48+1*/, int32_t *cell_idx /*n*/)
{
    grid_t g; grid_build(F, &g);
    memcpy(cell_start, g.start, sizeof(int) * (GRID_COLS * GRID_ROWS + 1));
    memcpy(cell_idx, g.idx, sizeof(int) * (size_t)g.start[GRID_COLS * GRID_ROWS]);
    grid_free(&g);
}

static int features_in_area(const orc_frame *F, const grid_t *g, float x, float y, float r, int minLevel,
                            int maxLevel, int *out, int cap)
{
    int n = 0;
    int nMinCellX = (int)floorf((x - F->minx - r) * F->grid_inv_w);
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - F->minx + r) * F->grid_inv_w);
    if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - F->miny - r) * F->grid_inv_h);
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - F->miny + r) * F->grid_inv_h);
    if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            int c = ix * GRID_ROWS + iy;
            for (int j = g->start[c]; j < g->start[c + 1]; j++) {
                int k = g->idx[j];
                if (bCheckLevels) {
                    if (F->octave[k] < minLevel) continue;
                    if (maxLevel >= 0 && F->octave[k] > maxLevel) continue;
                }
                const float distx = F->ux[k] - x, disty = F->uy[k] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) { if (n < cap) out[n] = k; n++; }
            }
        }
    return n;
}

int orc_features_in_area(const orc_frame *F, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap)
{
    grid_t g; grid_build(F, &g);
    int n = features_in_area(F, &g, x, y, r, minLevel, maxLevel, out, cap);
    grid_free(&g);
    return n;
}

/* ---------------------------------------------------------------- 8a-10 */
/* match_of_kp[i] on entry: -1 = keypoint free (no MapPoint, or one with Observations()==0),
 *                          -2 = keypoint holds a MapPoint with Observations()>0 (never overwritten).
 * on exit: >= 0 = index of the map point written by this call.  MP->obs_positive[m] tells whether map
 * point m has Observations()>0 (NULL = all do), which decides if a later point may overwrite it. */
int orc_search_by_projection_map(const orc_frame *F, const orc_mappoints *MP, float th, float nnratio,
                                 int32_t *match_of_kp)
{
    int nmatches = 0;
    const int bFactor = th != 1.0f;
    grid_t g; grid_build(F, &g);
    int *cand = (int *)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    for (int m = 0; m < MP->m; m++) {
        if (!MP->in_view[m]) continue;
        const int lvl = MP->level[m];
        float r = orc_radius_by_viewing_cos(MP->view_cos[m]);
        if (bFactor) r *= th;
        const float rad = r * F->scale_factors[lvl];
        int nc = features_in_area(F, &g, MP->proj_x[m], MP->proj_y[m], rad, lvl - 1, lvl, cand, F->n);
        if (nc == 0) continue;
        const uint8_t *d = MP->desc + (size_t)m * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int cur = match_of_kp[idx];
            if (cur == -2) continue;
            if (cur >= 0 && (!MP->obs_positive || MP->obs_positive[cur])) continue;
            if (F->uright[idx] > 0) {
                const float er = fabsf(MP->proj_xr[m] - F->uright[idx]);
                if (er > rad) continue;
            }
            const int dist = orc_hamming256(d, F->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = F->octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F->octave[idx];
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match_of_kp[bestIdx] = m;
            nmatches++;
        }
    }
    free(cand); grid_free(&g);
    return nmatches;
}

/* ---------------------------------------------------------------- ComputeThreeMaxima */
void orc_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ---------------------------------------------------------------- 8a-12 */
/* Last-frame tracking.  Rcw/tcw = current pose (row-major 3x3, 3), twc = -Rcw^T tcw is computed
 * here; Rlw/tlw = last pose.  last_* arrays describe LastFrame: world position of its map points
 * (has_mp[i] && !outlier[i] are projected), octave and angle of its keypoints.  fx..cy, bf, b.
 * match_of_kp: same protocol as orc_search_by_projection_map (values >= 0 are LAST-FRAME indices). */
int orc_search_by_projection_last(const orc_frame *Cur, const orc_lastframe *Last, const float *Rcw,
                                  const float *tcw, const float *Rlw, const float *tlw, float fx, float fy,
                                  float cx, float cy, float bf, float b, float th, int bMono, int checkOri,
                                  int32_t *match_of_kp)
{
    int nmatches = 0;
    const int HS = (Cur->n > Last->n ? Cur->n : Last->n) > 0 ? (Cur->n > Last->n ? Cur->n : Last->n) : 1;   /* one entry per ASSIGNMENT: <= Last->n */
    int *rotHist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (size_t)HS);
    int histN[HISTO_LENGTH];
    memset(histN, 0, sizeof(histN));
    /* twc = -Rcw^T * tcw ; tlc = Rlw*twc + tlw */
    float twc[3], tlc[3];
    /* cv::gemm: the transposed product goes through the generic double-accumulating kernel, the
     * plain 3x3*3x1 products through the float small-matrix path [UPSTREAM OpenCV 3.3 matmul.cpp] */
    for (int i = 0; i < 3; i++)
        twc[i] = (float)(-((double)Rcw[0 * 3 + i] * tcw[0] + (double)Rcw[1 * 3 + i] * tcw[1] + (double)Rcw[2 * 3 + i] * tcw[2]));
    for (int i = 0; i < 3; i++) tlc[i] = Rlw[i * 3 + 0] * twc[0] + Rlw[i * 3 + 1] * twc[1] + Rlw[i * 3 + 2] * twc[2] + tlw[i];
    const int bForward = tlc[2] > b && !bMono;
    const int bBackward = -tlc[2] > b && !bMono;
    grid_t g; grid_build(Cur, &g);
    int *cand = (int *)malloc(sizeof(int) * (Cur->n > 0 ? Cur->n : 1));
    for (int i = 0; i < Last->n; i++) {
        if (!Last->has_mp[i] || Last->outlier[i]) continue;
        const float *xw = Last->xw + 3 * (size_t)i;
        const float xc = Rcw[0] * xw[0] + Rcw[1] * xw[1] + Rcw[2] * xw[2] + tcw[0];
        const float yc = Rcw[3] * xw[0] + Rcw[4] * xw[1] + Rcw[5] * xw[2] + tcw[1];
        const float zc = Rcw[6] * xw[0] + Rcw[7] * xw[1] + Rcw[8] * xw[2] + tcw[2];
        const float invzc = (float)(1.0 / (double)zc);
        if (invzc < 0) continue;
        /* the binary contracts these three expressions into FMAs (so@0x81cba vfmadd213ss, 0x81cd9 vfmadd213ss, 0x81eb5 vfnmadd132ss) */
        const float u = fmaf(fx * xc, invzc, cx);
        const float v = fmaf(fy * yc, invzc, cy);
        if (u < Cur->minx || u > Cur->maxx) continue;
        if (v < Cur->miny || v > Cur->maxy) continue;
        const int nLastOctave = Last->octave[i];
        const float radius = th * Cur->scale_factors[nLastOctave];
        int nc;
        if (bForward) nc = features_in_area(Cur, &g, u, v, radius, nLastOctave, -1, cand, Cur->n);
        else if (bBackward) nc = features_in_area(Cur, &g, u, v, radius, 0, nLastOctave, cand, Cur->n);
        else nc = features_in_area(Cur, &g, u, v, radius, nLastOctave - 1, nLastOctave + 1, cand, Cur->n);
        if (nc == 0) continue;
        const uint8_t *dMP = Last->mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int cur = match_of_kp[i2];
            /* CurrentFrame.mvpMapPoints[i2] && ->Observations() > 0 (so@0x81e3d): a key point given to a last-frame point WITHOUT observations
             * earlier in this loop stays available, is overwritten, and counts (nmatches, rotation histogram) once per assignment */
            if (cur == -2 || (cur >= 0 && (!Last->obs_positive || Last->obs_positive[cur]))) continue;
            if (Cur->uright[i2] > 0) {
                const float ur = fmaf(-bf, invzc, u);
                const float er = fabsf(ur - Cur->uright[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_hamming256(dMP, Cur->desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            match_of_kp[bestIdx2] = i;
            nmatches++;
            if (checkOri) {
                float rot = Last->angle[i] - Cur->angle[bestIdx2];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * (1.0f / 12.0f)); /* this binary: factor = HISTO_LENGTH/360 */
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin * HS + histN[bin]++] = bestIdx2;
            }
        }
    }
    if (checkOri) {
        int i1, i2, i3;
        orc_three_maxima(histN, HISTO_LENGTH, &i1, &i2, &i3);
        for (int bnum = 0; bnum < HISTO_LENGTH; bnum++) {
            if (bnum == i1 || bnum == i2 || bnum == i3) continue;
            for (int j = 0; j < histN[bnum]; j++) {
                match_of_kp[rotHist[bnum * HS + j]] = -1;
                nmatches--;
            }
        }
    }
    free(cand); free(rotHist); grid_free(&g);
    return nmatches;
}

/* ---------------------------------------------------------------- relocalisation SearchByProjection */
/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * include/ORBmatcher.h:82, so@0x7e8c0 (listing read; executed from the binary for tests/golden/ref_glue_search_reloc.json).
 * valid[i] = pKF->GetMapPointMatches()[i] != NULL && !isBad() && !sAlreadyFound.count(pMP).  Per valid point:
 * x3Dc = Rcw*x3Dw + tcw; invzc = 1/zc (no sign test in this overload); u, v with the binary's FMA contraction
 * (so@0x7f4dc, so@0x7f4fb); image bounds; PO = x3Dw - Ow with Ow = -Rcw^T tcw; dist3D = float(cv::norm(PO)) (double
 * accumulation); reject outside [0.8*mfMinDistance, 1.2*mfMaxDistance] (MapPoint::Get{Min,Max}DistanceInvariance,
 * so@0x8fa40 / 0x8fad0); level = MapPoint::PredictScale(dist3D, &CurrentFrame) (so@0x8fc20: ceilf(logf(max/dist) /
 * mfLogScaleFactor), clamped to [0, mnScaleLevels-1]); radius = th * mvScaleFactors[level]; candidates of
 * GetFeaturesInArea(u, v, radius, level-1, level+1) that hold NO map point at all (match_of_kp == -1; every occupant
 * must be passed as -2); best Hamming distance <= ORBdist; rotation histogram over pKF->mvKeysUn[i].angle -
 * CurrentFrame.mvKeysUn[k].angle, everything outside the three dominant bins removed.
 * match_of_kp values >= 0 are keyframe feature indices.  Last->octave and Last->outlier are not read. */
int orc_search_by_projection_reloc(const orc_frame *Cur, const orc_lastframe *KF, const float *min_dist, const float *max_dist,
                                   const float *Rcw, const float *tcw, float fx, float fy, float cx, float cy, float log_scale_factor,
                                   float th, int ORBdist, int checkOri, int32_t *match_of_kp)
{
    int nmatches = 0;
    int *rotHist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (Cur->n > 0 ? Cur->n : 1));
    int histN[HISTO_LENGTH];
    memset(histN, 0, sizeof(histN));
    float Ow[3];
    for (int i = 0; i < 3; i++)
        Ow[i] = (float)(-((double)Rcw[0 * 3 + i] * tcw[0] + (double)Rcw[1 * 3 + i] * tcw[1] + (double)Rcw[2 * 3 + i] * tcw[2]));
    grid_t g; grid_build(Cur, &g);
    int *cand = (int *)malloc(sizeof(int) * (Cur->n > 0 ? Cur->n : 1));
    for (int i = 0; i < KF->n; i++) {
        if (!KF->has_mp[i]) continue;
        const float *xw = KF->xw + 3 * (size_t)i;
        const float xc = Rcw[0] * xw[0] + Rcw[1] * xw[1] + Rcw[2] * xw[2] + tcw[0];
        const float yc = Rcw[3] * xw[0] + Rcw[4] * xw[1] + Rcw[5] * xw[2] + tcw[1];
        const float zc = Rcw[6] * xw[0] + Rcw[7] * xw[1] + Rcw[8] * xw[2] + tcw[2];
        const float invzc = (float)(1.0 / (double)zc);
        const float u = fmaf(fx * xc, invzc, cx);
        const float v = fmaf(fy * yc, invzc, cy);
        if (u < Cur->minx || u > Cur->maxx) continue;
        if (v < Cur->miny || v > Cur->maxy) continue;
        const float PO[3] = {xw[0] - Ow[0], xw[1] - Ow[1], xw[2] - Ow[2]};
        double s2 = 0;
        for (int k = 0; k < 3; k++) s2 += (double)PO[k] * (double)PO[k];
        const float dist3D = (float)sqrt(s2);
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float ratio = max_dist[i] / dist3D;
        int lvl = (int)ceilf(logf(ratio) / log_scale_factor);
        if (lvl < 0) lvl = 0;
        else if (lvl >= Cur->nlevels) lvl = Cur->nlevels - 1;
        const float radius = th * Cur->scale_factors[lvl];
        const int nc = features_in_area(Cur, &g, u, v, radius, lvl - 1, lvl + 1, cand, Cur->n);
        if (nc == 0) continue;
        const uint8_t *dMP = KF->mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (match_of_kp[i2] != -1) continue;
            const int dist = orc_hamming256(dMP, Cur->desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            match_of_kp[bestIdx2] = i;
            nmatches++;
            if (checkOri) {
                float rot = KF->angle[i] - Cur->angle[bestIdx2];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * (1.0f / 12.0f));
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin * Cur->n + histN[bin]++] = bestIdx2;
            }
        }
    }
    if (checkOri) {
        int i1, i2, i3;
        orc_three_maxima(histN, HISTO_LENGTH, &i1, &i2, &i3);
        for (int bnum = 0; bnum < HISTO_LENGTH; bnum++) {
            if (bnum == i1 || bnum == i2 || bnum == i3) continue;
            for (int j = 0; j < histN[bnum]; j++) {
                match_of_kp[rotHist[bnum * Cur->n + j]] = -1;
                nmatches--;
            }
        }
    }
    free(cand); free(rotHist); grid_free(&g);
    return nmatches;
}

/* ---------------------------------------------------------------- SearchByBoW */
/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  include/ORBmatcher.h:104,
 * so@0x80150 (listing read; executed from the binary for tests/golden/ref_glue_bow.json).
 * The two DBoW2 feature vectors (std::map<NodeId, vector<unsigned>>) arrive flattened in key order:
 * node ids (ascending, unique), CSR starts, feature indices.  Walk of the binary: nodes present in both maps are
 * visited in key order (the lower_bound jumps only skip unmatched keys); for every keyframe feature of the node that
 * holds a good map point (kf_has_mp = vpMapPointsKF[i] != NULL && !isBad()): best / second-best Hamming distance
 * over the node's frame features that are still unmatched (running scan: ties keep the first index, the second-best
 * takes equal values), accept if best <= TH_LOW (so@0x808e0: cmpl $0x32) and float(best) < mfNNratio * float(second)
 * (so@0x808fc-0x80914), rotation histogram bin = roundf(rot * (1/12)) with rot = kf_angle - f_angle (+360 if
 * negative), bin 30 -> 0 (so@0x80980-0x809ad), then every match outside the three dominant bins is removed.
 * match_of_f[j] = keyframe feature index whose map point frame feature j received, -1 otherwise (all entries are
 * reset first, as the reference re-creates vpMapPointMatches).  Returns nmatches. */
int orc_search_by_bow(int n_kf, int n_f, const uint8_t *kf_desc, const uint8_t *f_desc, const float *kf_angle, const float *f_angle,
                      const uint8_t *kf_has_mp, int kf_nodes, const uint32_t *kf_node_id, const int32_t *kf_node_start,
                      const int32_t *kf_feat, int f_nodes, const uint32_t *f_node_id, const int32_t *f_node_start,
                      const int32_t *f_feat, float nnratio, int checkOri, int32_t *match_of_f)
{
    (void)n_kf;
    int nmatches = 0;
    int *rotHist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n_f > 0 ? n_f : 1));
    int histN[HISTO_LENGTH];
    memset(histN, 0, sizeof(histN));
    for (int j = 0; j < n_f; j++) match_of_f[j] = -1;
    int a = 0, b = 0;
    while (a < kf_nodes && b < f_nodes) {
        if (kf_node_id[a] < f_node_id[b]) { a++; continue; }
        if (kf_node_id[a] > f_node_id[b]) { b++; continue; }
        for (int p = kf_node_start[a]; p < kf_node_start[a + 1]; p++) {
            const int ikf = kf_feat[p];
            if (!kf_has_mp[ikf]) continue;
            const uint8_t *dKF = kf_desc + 32 * (size_t)ikf;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            for (int q = f_node_start[b]; q < f_node_start[b + 1]; q++) {
                const int jf = f_feat[q];
                if (match_of_f[jf] >= 0) continue;
                const int dist = orc_hamming256(dKF, f_desc + 32 * (size_t)jf);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = jf; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 <= TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                match_of_f[bestIdxF] = ikf;
                if (checkOri) {
                    float rot = kf_angle[ikf] - f_angle[bestIdxF];
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * (1.0f / 12.0f));
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin * n_f + histN[bin]++] = bestIdxF;
                }
                nmatches++;
            }
        }
        a++; b++;
    }
    if (checkOri) {
        int i1, i2, i3;
        orc_three_maxima(histN, HISTO_LENGTH, &i1, &i2, &i3);
        for (int bnum = 0; bnum < HISTO_LENGTH; bnum++) {
            if (bnum == i1 || bnum == i2 || bnum == i3) continue;
            for (int j = 0; j < histN[bnum]; j++) {
                match_of_f[rotHist[bnum * n_f + j]] = -1;
                nmatches--;
            }
        }
    }
    free(rotHist);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)  include/ORBmatcher.h:105,
 * so@0x82cc0 (loop closing; executed from the binary for tests/golden/ref_glue_bow_kf.json).  Same node walk as above
 * with these differences: both sides need a good map point (has_mp1 / has_mp2), occupancy is tracked on the KF2
 * features (vbMatched2), the acceptance threshold is STRICT (bestDist1 < TH_LOW, so@0x83490: cmpl $0x31), the output is
 * indexed by KF1 features: match12[i1] = KF2 feature whose map point KF1 feature i1 received, -1 otherwise. */
int orc_search_by_bow_kf(int n1, int n2, const uint8_t *desc1, const uint8_t *desc2, const float *angle1, const float *angle2,
                         const uint8_t *has_mp1, const uint8_t *has_mp2, int nodes1, const uint32_t *node_id1, const int32_t *node_start1,
                         const int32_t *feat1, int nodes2, const uint32_t *node_id2, const int32_t *node_start2, const int32_t *feat2,
                         float nnratio, int checkOri, int32_t *match12)
{
    int nmatches = 0;
    int *rotHist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n1 > 0 ? n1 : 1));
    uint8_t *matched2 = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1);
    int histN[HISTO_LENGTH];
    memset(histN, 0, sizeof(histN));
    for (int i = 0; i < n1; i++) match12[i] = -1;
    int a = 0, b = 0;
    while (a < nodes1 && b < nodes2) {
        if (node_id1[a] < node_id2[b]) { a++; continue; }
        if (node_id1[a] > node_id2[b]) { b++; continue; }
        for (int p = node_start1[a]; p < node_start1[a + 1]; p++) {
            const int i1 = feat1[p];
            if (!has_mp1[i1]) continue;
            const uint8_t *d1 = desc1 + 32 * (size_t)i1;
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int q = node_start2[b]; q < node_start2[b + 1]; q++) {
                const int i2 = feat2[q];
                if (matched2[i2] || !has_mp2[i2]) continue;
                const int dist = orc_hamming256(d1, desc2 + 32 * (size_t)i2);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 < TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                match12[i1] = bestIdx2;
                matched2[bestIdx2] = 1;
                if (checkOri) {
                    float rot = angle1[i1] - angle2[bestIdx2];
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * (1.0f / 12.0f));
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin * n1 + histN[bin]++] = i1;
                }
                nmatches++;
            }
        }
        a++; b++;
    }
    if (checkOri) {
        int i1, i2, i3;
        orc_three_maxima(histN, HISTO_LENGTH, &i1, &i2, &i3);
        for (int bnum = 0; bnum < HISTO_LENGTH; bnum++) {
            if (bnum == i1 || bnum == i2 || bnum == i3) continue;
            for (int j = 0; j < histN[bnum]; j++) {
                match12[rotHist[bnum * n1 + j]] = -1;
                nmatches--;
            }
        }
    }
    free(rotHist); free(matched2);
    return nmatches;
}

/* ---------------------------------------------------------------- Fuse */
/* MapPoint::PredictScale(const float&, KeyFrame*)  so@0x8fb60: ceilf(logf(mfMaxDistance / dist) / mfLogScaleFactor) clamped */
static int predict_scale(float max_dist, float dist, float log_scale_factor, int nlevels)
{
    const float ratio = max_dist / dist;
    int lvl = (int)ceilf(logf(ratio) / log_scale_factor);
    if (lvl < 0) lvl = 0;
    else if (lvl >= nlevels) lvl = nlevels - 1;
    return lvl;
}

/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th)  include/ORBmatcher.h:119, so@0x7a500
 * (listing read; executed from the binary for tests/golden/ref_glue_fuse.json).  Per map point, in list order:
 *   p3Dc = Rcw*p3Dw + tcw (cv::gemm small-matrix float path); skip if z < 0; invz = 1/z; x = X*invz, y = Y*invz;
 *   u = fmaf(x, fx, cx), v = fmaf(fy, y, cy) -- contracted in the binary (so@0x7abc8, so@0x7abf0);
 *   KeyFrame::IsInImage (so@0x97480: u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY);
 *   dist3D = float(cv::norm(p3Dw - Ow)); skip unless 0.8f*mfMinDistance <= dist3D <= 1.2f*mfMaxDistance (so@0x8fa40/0x8fad0);
 *   skip if PO.dot(Pn) < 0.5*dist3D (doubles, so@0x7b4b4-0x7b4d2); level = PredictScale; radius = th*mvScaleFactors[level];
 *   candidates = KeyFrame::GetFeaturesInArea(u, v, radius) (so@0x96fe0: the Frame cell walk without a level filter);
 *   keep key points with level-1 <= octave <= level; reprojection test with e = (u - kx, v - ky[, ur - kur]),
 *   ur = fmaf(-bf, invz, u) (so@0x7b63e), e2 = fmaf(er, er, fmaf(ex, ex, ey*ey)) (so@0x7b649-0x7b652),
 *   skip if double(e2 * mvInvLevelSigma2[octave]) > 7.8 (stereo key point, mvuRight >= 0) / 5.99 (mono);
 *   best Hamming distance, first index wins ties; a point is fused when best <= TH_LOW.
 * The map mutation that follows (Replace / AddObservation / AddMapPoint) reads nothing this search depends on and
 * stays on the host: best_idx[i] = key point of pKF for map point i, -1 = none.  P->valid[i] = pMP != NULL &&
 * !pMP->isBad() && !pMP->IsInKeyFrame(pKF).  Returns nFused. */
int orc_fuse(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, float th, int32_t *best_idx, int32_t *best_dist)
{
    int nFused = 0;
    grid_t g; grid_build(KF, &g);
    int *cand = (int *)malloc(sizeof(int) * (KF->n > 0 ? KF->n : 1));
    for (int i = 0; i < P->m; i++) {
        best_idx[i] = -1;
        if (best_dist) best_dist[i] = 256;
        if (!P->valid[i]) continue;
        const float *xw = P->xw + 3 * (size_t)i;
        const float X = C->Rcw[0] * xw[0] + C->Rcw[1] * xw[1] + C->Rcw[2] * xw[2] + C->tcw[0];
        const float Y = C->Rcw[3] * xw[0] + C->Rcw[4] * xw[1] + C->Rcw[5] * xw[2] + C->tcw[1];
        const float Z = C->Rcw[6] * xw[0] + C->Rcw[7] * xw[1] + C->Rcw[8] * xw[2] + C->tcw[2];
        if (Z < 0.0f) continue;
        const float invz = 1.0f / Z;
        const float x = X * invz, y = Y * invz;
        const float u = fmaf(x, C->fx, C->cx), v = fmaf(C->fy, y, C->cy);
        if (!(u >= KF->minx && u < KF->maxx && v >= KF->miny && v < KF->maxy)) continue;
        const float PO[3] = {xw[0] - C->Ow[0], xw[1] - C->Ow[1], xw[2] - C->Ow[2]};
        double s2 = 0;
        for (int k = 0; k < 3; k++) s2 += (double)PO[k] * (double)PO[k];
        const float dist3D = (float)sqrt(s2);
        const float maxDistance = 1.2f * P->max_dist[i], minDistance = 0.8f * P->min_dist[i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float *Pn = P->normal + 3 * (size_t)i;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn[k];
        if (dot < 0.5 * (double)dist3D) continue;
        const int lvl = predict_scale(P->max_dist[i], dist3D, C->log_scale_factor, KF->nlevels);
        const float radius = th * KF->scale_factors[lvl];
        const int nc = features_in_area(KF, &g, u, v, radius, -1, -1, cand, KF->n);
        if (nc == 0) continue;
        const uint8_t *dMP = P->desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int kpLevel = KF->octave[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            const float ey = v - KF->uy[idx], ex = u - KF->ux[idx];
            if (KF->uright && KF->uright[idx] >= 0.0f) {
                const float ur = fmaf(-C->bf, invz, u);
                const float er = ur - KF->uright[idx];
                const float e2 = fmaf(er, er, fmaf(ex, ex, ey * ey));
                if ((double)(e2 * C->inv_level_sigma2[kpLevel]) > 7.8) continue;
            } else {
                const float e2 = fmaf(ex, ex, ey * ey);
                if ((double)(e2 * C->inv_level_sigma2[kpLevel]) > 5.99) continue;
            }
            const int dist = orc_hamming256(dMP, KF->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
        if (best_dist) best_dist[i] = bestDist;
    }
    free(cand); grid_free(&g);
    return nFused;
}

/* Decomposition of a Sim3 matrix at the top of Fuse(KeyFrame*, Scw, ...) (so@0x7bb20) and SearchByProjection(KeyFrame*, Scw, ...) (so@0x880f0):
 *   sRcw = Scw.rowRange(0,3).colRange(0,3); scw = float(sqrt(sRcw.row(0).dot(sRcw.row(0))))   (Mat::dot accumulates in double; so@0x7bda0-0x7bdc2)
 *   Rcw = sRcw / scw; tcw = Scw.rowRange(0,3).col(3) / scw     cv::operator/(Mat, double) = a scale expression with alpha = 1.0 / double(scw), assigned by
 *                                                              Mat::convertTo: OpenCV 3.3 cvtScale32f works in float, d = s * float(alpha) (+ 0)
 *   Ow = -Rcw.t() * tcw                                        cv::gemm transposed path: float(double-sum * -1)
 * OpenCV 3.3.0 (the reference's pinned dependency) is not in the image: this follows its published convert.cpp / matmul.cpp. */
void orc_sim3_decompose(const float *Scw /*4x4 row-major*/, float *Rcw, float *tcw, float *Ow)
{
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)sqrt(dot);
    const float alpha = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Scw[r * 4 + c] * alpha;
        tcw[r] = Scw[r * 4 + 3] * alpha;
    }
    for (int i = 0; i < 3; i++)
        Ow[i] = (float)(((double)Rcw[0 * 3 + i] * tcw[0] + (double)Rcw[1 * 3 + i] * tcw[1] + (double)Rcw[2 * 3 + i] * tcw[2]) * -1.0);
}

/* shared gates of the two Scw overloads: camera point, image test, invariance range, viewing angle, predicted level.
 * u = fmaf(x, fx, cx), v = fmaf(y, fy, cy) (so@0x7caac / 0x7cabe, so@0x8914e / 0x89160). */
static int sim3_gates(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, int i, float *u, float *v, int *lvl)
{
    const float *xw = P->xw + 3 * (size_t)i;
    const float X = C->Rcw[0] * xw[0] + C->Rcw[1] * xw[1] + C->Rcw[2] * xw[2] + C->tcw[0];
    const float Y = C->Rcw[3] * xw[0] + C->Rcw[4] * xw[1] + C->Rcw[5] * xw[2] + C->tcw[1];
    const float Z = C->Rcw[6] * xw[0] + C->Rcw[7] * xw[1] + C->Rcw[8] * xw[2] + C->tcw[2];
    if (Z < 0.0f) return 0;
    const float invz = 1.0f / Z;
    const float x = X * invz, y = Y * invz;
    *u = fmaf(x, C->fx, C->cx); *v = fmaf(y, C->fy, C->cy);
    if (!(*u >= KF->minx && *u < KF->maxx && *v >= KF->miny && *v < KF->maxy)) return 0;
    const float PO[3] = {xw[0] - C->Ow[0], xw[1] - C->Ow[1], xw[2] - C->Ow[2]};
    double s2 = 0;
    for (int k = 0; k < 3; k++) s2 += (double)PO[k] * (double)PO[k];
    const float dist3D = (float)sqrt(s2);
    const float maxDistance = 1.2f * P->max_dist[i], minDistance = 0.8f * P->min_dist[i];
    if (dist3D < minDistance || dist3D > maxDistance) return 0;
    const float *Pn = P->normal + 3 * (size_t)i;
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn[k];
    if (dot < 0.5 * (double)dist3D) return 0;
    *lvl = predict_scale(P->max_dist[i], dist3D, C->log_scale_factor, KF->nlevels);
    return 1;
}

/* ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint)
 * include/ORBmatcher.h:122, so@0x7bb20 (loop closing).  C holds Rcw / tcw / Ow from orc_sim3_decompose and the keyframe intrinsics
 * (bf and inv_level_sigma2 unread: this overload has no reprojection test).  P->valid[i] = !isBad() && !pKF->GetMapPoints().count(pMP).
 * best_idx[i] = key point of pKF with the lowest distance among level-1 <= octave <= level, kept if <= TH_LOW.  The reference then
 * either records vpReplacePoint[i] = pKF->GetMapPoint(best) or adds the observation: host work.  Returns nFused. */
int orc_fuse_sim3(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, float th, int32_t *best_idx)
{
    int nFused = 0;
    grid_t g; grid_build(KF, &g);
    int *cand = (int *)malloc(sizeof(int) * (KF->n > 0 ? KF->n : 1));
    for (int i = 0; i < P->m; i++) {
        best_idx[i] = -1;
        if (!P->valid[i]) continue;
        float u, v; int lvl;
        if (!sim3_gates(KF, C, P, i, &u, &v, &lvl)) continue;
        const float radius = th * KF->scale_factors[lvl];
        const int nc = features_in_area(KF, &g, u, v, radius, -1, -1, cand, KF->n);
        const uint8_t *dMP = P->desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (KF->octave[idx] < lvl - 1 || KF->octave[idx] > lvl) continue;
            const int dist = orc_hamming256(dMP, KF->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    free(cand); grid_free(&g);
    return nFused;
}

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th)
 * include/ORBmatcher.h:86, so@0x880f0 (loop closing).  P->valid[i] = !isBad() && pMP not among vpMatched on entry.
 * match_of_kp[k]: in  -1 = vpMatched[k] == NULL, -2 = occupied; out >= 0 = index of the map point stored by this call
 * (key points taken earlier in the loop are skipped by later points: the loop is greedy in list order).  radius = float(th) *
 * mvScaleFactors[level] (so@0x89b22).  Returns nmatches. */
int orc_search_by_projection_sim3(const orc_frame *KF, const orc_kf_pose *C, const orc_points3d *P, int th, int32_t *match_of_kp)
{
    int nmatches = 0;
    grid_t g; grid_build(KF, &g);
    int *cand = (int *)malloc(sizeof(int) * (KF->n > 0 ? KF->n : 1));
    for (int i = 0; i < P->m; i++) {
        if (!P->valid[i]) continue;
        float u, v; int lvl;
        if (!sim3_gates(KF, C, P, i, &u, &v, &lvl)) continue;
        const float radius = (float)th * KF->scale_factors[lvl];
        const int nc = features_in_area(KF, &g, u, v, radius, -1, -1, cand, KF->n);
        const uint8_t *dMP = P->desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (match_of_kp[idx] != -1) continue;
            if (KF->octave[idx] < lvl - 1 || KF->octave[idx] > lvl) continue;
            const int dist = orc_hamming256(dMP, KF->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { match_of_kp[bestIdx] = i; nmatches++; }
    }
    free(cand); grid_free(&g);
    return nmatches;
}

/* one direction of SearchBySim3: the points of the source keyframe into the target keyframe.  cam point = R2 * (R * xw + t) + t2 (two
 * cv::gemm small-matrix float products), skip z < 0, u = fmaf(x, fx, cx), v = fmaf(y, fy, cy) (so@0x84af1 / 0x84b03, so@0x8589f / 0x858b1),
 * KeyFrame::IsInImage, dist3D = float(cv::norm(cam point)) within [0.8 mfMinDistance, 1.2 mfMaxDistance], PredictScale on the target,
 * radius = th * mvScaleFactors[level], best distance among level-1 <= octave <= level, kept if <= TH_HIGH (so@0x86692 / 0x866e9: cmpl $0x64). */
static void sim3_direction(const orc_frame *T, const orc_kf_pose *Cs /*source pose*/, const orc_kf_pose *Ct /*target: scale pyramid*/,
                           const orc_kf_pose *Ci /*intrinsics: ALWAYS keyframe 1's, so@0x84acc / 0x8586c load fx..cy from pKF1 in both directions*/, const float *R2,
                           const float *t2, const orc_points3d *P, float th, int32_t *vnMatch)
{
    grid_t g; grid_build(T, &g);
    int *cand = (int *)malloc(sizeof(int) * (T->n > 0 ? T->n : 1));
    for (int i = 0; i < P->m; i++) {
        vnMatch[i] = -1;
        if (!P->valid[i]) continue;
        const float *xw = P->xw + 3 * (size_t)i;
        const float a = Cs->Rcw[0] * xw[0] + Cs->Rcw[1] * xw[1] + Cs->Rcw[2] * xw[2] + Cs->tcw[0];
        const float b = Cs->Rcw[3] * xw[0] + Cs->Rcw[4] * xw[1] + Cs->Rcw[5] * xw[2] + Cs->tcw[1];
        const float c = Cs->Rcw[6] * xw[0] + Cs->Rcw[7] * xw[1] + Cs->Rcw[8] * xw[2] + Cs->tcw[2];
        const float X = R2[0] * a + R2[1] * b + R2[2] * c + t2[0];
        const float Y = R2[3] * a + R2[4] * b + R2[5] * c + t2[1];
        const float Z = R2[6] * a + R2[7] * b + R2[8] * c + t2[2];
        if (Z < 0.0f) continue;
        const float invz = 1.0f / Z;
        const float x = X * invz, y = Y * invz;
        const float u = fmaf(x, Ci->fx, Ci->cx), v = fmaf(y, Ci->fy, Ci->cy);
        if (!(u >= T->minx && u < T->maxx && v >= T->miny && v < T->maxy)) continue;
        const double s2 = (double)X * (double)X + (double)Y * (double)Y + (double)Z * (double)Z;
        const float dist3D = (float)sqrt(s2);
        const float maxDistance = 1.2f * P->max_dist[i], minDistance = 0.8f * P->min_dist[i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int lvl = predict_scale(P->max_dist[i], dist3D, Ct->log_scale_factor, T->nlevels);
        const float radius = th * T->scale_factors[lvl];
        const int nc = features_in_area(T, &g, u, v, radius, -1, -1, cand, T->n);
        const uint8_t *dMP = P->desc + 32 * (size_t)i;
        int bestDist = 0x7fffffff, bestIdx = -1;
        for (int q = 0; q < nc; q++) {
            const int idx = cand[q];
            if (T->octave[idx] < lvl - 1 || T->octave[idx] > lvl) continue;
            const int dist = orc_hamming256(dMP, T->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
    }
    free(cand); grid_free(&g);
}

/* ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
 *                          const cv::Mat& t12, const float th)   include/ORBmatcher.h:116, so@0x838b0 (loop closing).
 * sR12 = s12 * R12 and sR21 = (1.0 / s12) * R12.t() are scale expressions assigned through Mat::convertTo (float work type in OpenCV 3.3:
 * element * float(alpha)); t21 = -sR21 * t12 is cv::gemm with alpha = -1 (small-matrix float path).  P1 / P2 = the map points of the two
 * keyframes (one per key point, m = n): valid = pointer non-null, !vbAlreadyMatched, !isBad().  A pair is accepted when both directions
 * agree.  match12[i1] = key point of KF2 (the caller stores vpMapPoints2[match12[i1]]), -1 otherwise.  Returns nFound. */
int orc_search_by_sim3(const orc_frame *KF1, const orc_frame *KF2, const orc_kf_pose *C1, const orc_kf_pose *C2, float s12, const float *R12,
                       const float *t12, float th, const orc_points3d *P1, const orc_points3d *P2, int32_t *match12)
{
    float sR12[9], sR21[9], t21[3];
    const float a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sR12[r * 3 + c] = R12[r * 3 + c] * s12; sR21[r * 3 + c] = R12[c * 3 + r] * a21; }
    for (int r = 0; r < 3; r++) {
        const float t = sR21[r * 3] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
        t21[r] = (float)((double)t * -1.0);
    }
    int32_t *vn1 = (int32_t *)malloc(sizeof(int32_t) * (P1->m > 0 ? P1->m : 1)), *vn2 = (int32_t *)malloc(sizeof(int32_t) * (P2->m > 0 ? P2->m : 1));
    sim3_direction(KF2, C1, C2, C1, sR21, t21, P1, th, vn1);
    sim3_direction(KF1, C2, C1, C1, sR12, t12, P2, th, vn2);
    int nFound = 0;
    for (int i1 = 0; i1 < P1->m; i1++) {
        match12[i1] = -1;
        const int idx2 = vn1[i1];
        if (idx2 >= 0 && idx2 < P2->m && vn2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    free(vn1); free(vn2);
    return nFound;
}

/* ---------------------------------------------------------------- SearchForTriangulation */
/* ORBmatcher::CheckDistEpipolarLine(kp1, kp2, F12, pKF2)  include/ORBmatcher.h (protected), so@0x79b90.  The binary's contractions:
 *   a = fmaf(x1, F00, y1*F10) + F20;  b = fmaf(x1, F01, y1*F11) + F21;  c = fmaf(y1, F12, x1*F02) + F22;
 *   den = fmaf(a, a, b*b) (== 0 -> false);  num = c + fmaf(b, y2, a*x2);  dsqr = num*num/den;
 *   accept when 3.84 * double(mvLevelSigma2[kp2.octave]) > double(dsqr). */
int orc_check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float *F12 /*3x3 row-major*/, float level_sigma2)
{
    const float b = fmaf(x1, F12[1], y1 * F12[4]) + F12[7];
    const float a = fmaf(x1, F12[0], y1 * F12[3]) + F12[6];
    const float den = fmaf(a, a, b * b);
    if (den == 0.0f) return 0;
    const float c = fmaf(y1, F12[5], x1 * F12[2]) + F12[8];
    const float num = c + fmaf(b, y2, a * x2);
    const float dsqr = num * num / den;
    return 3.84 * (double)level_sigma2 > (double)dsqr;
}

/* Epipole of keyframe 1 in keyframe 2 (so@0x86b9c-0x86f95): C2 = R2w*Cw + t2w (cv::gemm small-matrix float path), invz = 1/C2z,
 * ex = fmaf(fx*C2x, invz, cx), ey = fmaf(fy*C2y, invz, cy). */
void orc_epipole(const float *Cw, const float *R2w, const float *t2w, float fx, float fy, float cx, float cy, float *ex, float *ey)
{
    float C2[3];
    for (int r = 0; r < 3; r++) {
        const float t = R2w[r * 3] * Cw[0] + R2w[r * 3 + 1] * Cw[1] + R2w[r * 3 + 2] * Cw[2];
        C2[r] = (float)((double)t + (double)t2w[r]);
    }
    const float invz = 1.0f / C2[2];
    *ex = fmaf(fx * C2[0], invz, cx);
    *ey = fmaf(fy * C2[1], invz, cy);
}

/* ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vector<pair<size_t,size_t>>& vMatchedPairs,
 *                                    const bool bOnlyStereo)   include/ORBmatcher.h:111, so@0x86b30 (LocalMapping::CreateNewMapPoints).
 * Walk over the common DBoW2 nodes in key order (feature vectors flattened as for orc_search_by_bow_kf).  Per keyframe-1 feature WITHOUT a
 * map point (has_mp1 = pKF1->GetMapPoint(i) != NULL -> skipped), optionally stereo only: over the node's keyframe-2 features that are
 * unmatched (vbMatched2 -- this binary sets it, so@0x87bc9) and hold no map point: dist <= TH_LOW and dist <= bestDist (so@0x87a01-0x87a13:
 * a later equal distance replaces the earlier one), for two mono key points the epipole test fmaf(dx, dx, dy*dy) < 100 * mvScaleFactors2[octave]
 * rejects (so@0x87a52-0x87a91), then CheckDistEpipolarLine.  Orientation histogram: bin = roundf(rot * (1/12)) (so@0x87c07), 30 -> 0.
 * match12[i1] = keyframe-2 feature, -1 none (the reference emits the pairs (i1, match12[i1]) in i1 order).  Returns nmatches. */
int orc_search_for_triangulation(int n1, int n2, const float *x1, const float *y1, const float *angle1, const float *uright1, const uint8_t *desc1,
                                 const uint8_t *has_mp1, const float *x2, const float *y2, const float *angle2, const int32_t *octave2,
                                 const float *uright2, const uint8_t *desc2, const uint8_t *has_mp2, int nodes1, const uint32_t *node_id1,
                                 const int32_t *node_start1, const int32_t *feat1, int nodes2, const uint32_t *node_id2, const int32_t *node_start2,
                                 const int32_t *feat2, const float *F12, float ex, float ey, const float *scale_factors2, const float *level_sigma2_2,
                                 int bOnlyStereo, int checkOri, int32_t *match12)
{
    int nmatches = 0;
    int *rotHist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n1 > 0 ? n1 : 1));
    uint8_t *matched2 = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1);
    int histN[HISTO_LENGTH];
    memset(histN, 0, sizeof(histN));
    for (int i = 0; i < n1; i++) match12[i] = -1;
    int a = 0, b = 0;
    while (a < nodes1 && b < nodes2) {
        if (node_id1[a] < node_id2[b]) { a++; continue; }
        if (node_id1[a] > node_id2[b]) { b++; continue; }
        for (int p = node_start1[a]; p < node_start1[a + 1]; p++) {
            const int idx1 = feat1[p];
            if (has_mp1[idx1]) continue;
            const int bStereo1 = uright1[idx1] >= 0.0f;
            if (bOnlyStereo && !bStereo1) continue;
            const uint8_t *d1 = desc1 + 32 * (size_t)idx1;
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int q = node_start2[b]; q < node_start2[b + 1]; q++) {
                const int idx2 = feat2[q];
                if (matched2[idx2] || has_mp2[idx2]) continue;
                const int bStereo2 = uright2[idx2] >= 0.0f;
                if (bOnlyStereo && !bStereo2) continue;
                const int dist = orc_hamming256(d1, desc2 + 32 * (size_t)idx2);
                if (dist > TH_LOW || dist > bestDist) continue;
                if (!bStereo1 && !bStereo2) {
                    const float distex = ex - x2[idx2], distey = ey - y2[idx2];
                    if (fmaf(distex, distex, distey * distey) < 100.0f * scale_factors2[octave2[idx2]]) continue;
                }
                if (orc_check_dist_epipolar_line(x1[idx1], y1[idx1], x2[idx2], y2[idx2], F12, level_sigma2_2[octave2[idx2]])) { bestIdx2 = idx2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                match12[idx1] = bestIdx2;
                matched2[bestIdx2] = 1;
                nmatches++;
                if (checkOri) {
                    float rot = angle1[idx1] - angle2[bestIdx2];
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * (1.0f / 12.0f));
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin * n1 + histN[bin]++] = idx1;
                }
            }
        }
        a++; b++;
    }
    if (checkOri) {
        int i1, i2, i3;
        orc_three_maxima(histN, HISTO_LENGTH, &i1, &i2, &i3);
        for (int bnum = 0; bnum < HISTO_LENGTH; bnum++) {
            if (bnum == i1 || bnum == i2 || bnum == i3) continue;
            for (int j = 0; j < histN[bnum]; j++) {
                match12[rotHist[bnum * n1 + j]] = -1;
                nmatches--;
            }
        }
    }
    free(rotHist); free(matched2);
    return nmatches;
}

/* ---------------------------------------------------------------- BF kNN (k=2), cv::batchDistance semantics */
int orc_knn2_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    for (int i = 0; i < nq; i++) {
        int d0 = 0x7fffffff, d1 = 0x7fffffff, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_hamming256(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < d1) {
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
    return nq;
}

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* Frame::lineDescriptorMAD: medians are order statistics, so sort tie order is irrelevant */
void orc_line_mad(const int32_t *dist /*n x 2*/, int n, double *nn_mad, double *nn12_mad)
{
    *nn_mad = *nn12_mad = 0;
    if (n <= 0) return;
    float *v = (float *)malloc(sizeof(float) * n);
    for (int i = 0; i < n; i++) v[i] = (float)dist[2 * i];
    qsort(v, n, sizeof(float), cmp_float);
    double med = v[n / 2];
    for (int i = 0; i < n; i++) v[i] = fabsf((float)((double)(float)dist[2 * i] - med));
    qsort(v, n, sizeof(float), cmp_float);
    *nn_mad = 1.4826 * v[n / 2];
    for (int i = 0; i < n; i++) v[i] = (float)dist[2 * i + 1] - (float)dist[2 * i];
    qsort(v, n, sizeof(float), cmp_float);
    med = v[n / 2];
    for (int i = 0; i < n; i++) v[i] = fabsf((float)((double)((float)dist[2 * i + 1] - (float)dist[2 * i]) - med));
    qsort(v, n, sizeof(float), cmp_float);
    *nn12_mad = 1.4826 * v[n / 2];
    free(v);
}

/* double LineSegment::LineSegmentOverlap(spl_obs, epl_obs, spl_proj, epl_proj)  include/ExtractLineSegment.h:47 -- declared without a body in the
 * snapshot; restated from the PL-SLAM family's lineSegmentOverlap (PARITY UNPINNED): both intervals ordered, length = max(obs) - min(proj),
 * overlap = 0 when disjoint, the observed extent when the projection covers it, else min(ends) - max(starts); divided by length when
 * length > 0.01f, else 0. */
double orc_line_segment_overlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj)
{
    double sln = fmin(spl_obs, epl_obs), eln = fmax(spl_obs, epl_obs);
    double spn = fmin(spl_proj, epl_proj), epn = fmax(spl_proj, epl_proj);
    double length = eln - spn;
    double overlap;
    if ((epn < sln) || (spn > eln)) overlap = 0.f;
    else {
        if ((epn > eln) && (spn < sln)) overlap = eln - sln;
        else overlap = fmin(eln, epn) - fmax(sln, spn);
    }
    if (length > 0.01f) overlap = overlap / length;
    else overlap = 0.f;
    return overlap;
}

/* LSDmatcher::SearchByProjection(CurrentFrame, LastFrame): query = last-frame lines, train = current.
 * last_has_mapline[q] says whether LastFrame.mvpMapLines[q] != NULL.  match_of_line[t] = q written. */
int orc_match_lines_knn(const uint8_t *last_desc, int nlast, const uint8_t *cur_desc, int ncur,
                        const uint8_t *last_has_mapline, int32_t *match_of_line)
{
    if (nlast <= 0 || ncur < 2) return 0;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * nlast), *dist = (int32_t *)malloc(sizeof(int32_t) * 2 * nlast);
    orc_knn2_hamming(last_desc, nlast, cur_desc, ncur, idx, dist);
    double nn_mad, nn12_mad;
    orc_line_mad(dist, nlast, &nn_mad, &nn12_mad);
    const double th12 = nn12_mad * 0.5;
    int n = 0;
    for (int q = 0; q < nlast; q++) { /* sorted by queryIdx == natural order */
        double d12 = (double)((float)dist[2 * q + 1] - (float)dist[2 * q]);
        if (d12 > th12 && last_has_mapline[q]) { match_of_line[idx[2 * q]] = q; n++; }
    }
    free(idx); free(dist);
    return n;
}

/* ---------------------------------------------------------------- the two remaining LSDmatcher overloads (bodies absent: PARITY UNPINNED)
 * int LSDmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo)
 * include/LSDmatcher.h:54 (LocalMapping::CreateNewMapLines).  Restated from the PL-SLAM family this fork derives from -- the same brute-force rule
 * as LSDmatcher::SearchByProjection(Cur, Last): BinaryDescriptorMatcher::knnMatch(pKF1->mLineDescriptors, pKF2->mLineDescriptors, k = 2),
 * KeyFrame::lineDescriptorMAD (include/KeyFrame.h:159), nn12 threshold = mad_factor * nn12_mad (0.1 upstream), matches walked in queryIdx order,
 * a pair (q, t) kept when d2 - d1 exceeds the threshold and NEITHER line holds a MapLine yet (the point overload: both pMP1 and pMP2 must be
 * NULL, ORBmatcher.h:111); bOnlyStereo -- this fork's added argument -- applies the point overload's rule to lines: both lines need stereo data
 * (end-point depths, Frame.h:208-211).  match12[q] = t or -1; returns the number of pairs. */
int orc_lines_search_for_triangulation(const uint8_t *desc1, int n1, const uint8_t *desc2, int n2, const uint8_t *has_ml1, const uint8_t *has_ml2,
                                       const uint8_t *stereo1, const uint8_t *stereo2, int only_stereo, double mad_factor, int32_t *match12)
{
    for (int q = 0; q < n1; q++) match12[q] = -1;
    if (n1 <= 0 || n2 < 2) return 0;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * n1), *dist = (int32_t *)malloc(sizeof(int32_t) * 2 * n1);
    orc_knn2_hamming(desc1, n1, desc2, n2, idx, dist);
    double nn_mad, nn12_mad;
    orc_line_mad(dist, n1, &nn_mad, &nn12_mad);
    const double th12 = nn12_mad * mad_factor;
    int n = 0;
    for (int q = 0; q < n1; q++) {
        const int t = idx[2 * q];
        const double d12 = (double)((float)dist[2 * q + 1] - (float)dist[2 * q]);
        if (!(d12 > th12)) continue;
        if (has_ml1[q] || has_ml2[t]) continue;
        if (only_stereo && (!stereo1[q] || !stereo2[t])) continue;
        match12[q] = t; n++;
    }
    free(idx); free(dist);
    return n;
}

/* int LSDmatcher::Fuse(KeyFrame *pKF, const vector<MapLine*> &vpMapLines)   include/LSDmatcher.h:58 (LocalMapping::SearchInNeighbors) -- the search
 * half.  PL-SLAM family: no projection (the signature has no th); for every map line that is non-NULL, not bad and not already in the keyframe
 * (valid[i]) the keyframe line with the smallest Hamming distance to MapLine::mLDescriptor over ALL of pKF->mLineDescriptors (first minimum),
 * fused when that distance is <= TH_LOW.  best_idx[i] = keyframe line or -1; returns the number fused.  The map mutation (Replace / AddObservation /
 * AddMapLine, decided by pKF->GetMapLine(best_idx)) is the caller's, in list order, as for ORBmatcher::Fuse. */
int orc_lines_fuse(const uint8_t *kf_desc, int n_kf, const uint8_t *ml_desc, const uint8_t *valid, int m, int32_t *best_idx)
{
    int nfused = 0;
    for (int i = 0; i < m; i++) {
        best_idx[i] = -1;
        if (!valid[i]) continue;
        int bestDist = 256, bestIdx = -1;
        for (int j = 0; j < n_kf; j++) {
            const int d = orc_hamming256(ml_desc + 32 * (size_t)i, kf_desc + 32 * (size_t)j);
            if (d < bestDist) { bestDist = d; bestIdx = j; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nfused++; }
    }
    return nfused;
}

/* Frame::GetLinesInArea [UPSTREAM] */
int orc_lines_in_area(const orc_lineframe *F, float x1, float y1, float x2, float y2, float r, int minLevel,
                      int maxLevel, int *out, int cap)
{
    int n = 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel > 0);
    for (int i = 0; i < F->n; i++) {
        double dx = 0.5 * (x1 + x2) - F->pt_x[i], dy = 0.5 * (y1 + y2) - F->pt_y[i];
        double distance = dx * dx + dy * dy;
        if (distance > r * r) continue;
        float slope = (y1 - y2) / (x1 - x2) - F->angle[i];
        if (slope > r * 0.01) continue;
        if (bCheckLevels) {
            if (F->octave[i] < minLevel) continue;
            if (maxLevel >= 0 && F->octave[i] > maxLevel) continue;
        }
        if (n < cap) out[n] = i;
        n++;
    }
    return n;
}

int orc_search_by_projection_lines(const orc_lineframe *F, const orc_maplines *ML, float th, float nnratio,
                                   int32_t *match_of_line)
{
    int nmatches = 0;
    const int bFactor = th != 1.0f;
    int *cand = (int *)malloc(sizeof(int) * (F->n > 0 ? F->n : 1));
    for (int m = 0; m < ML->m; m++) {
        if (!ML->in_view[m]) continue;
        const int lvl = ML->level[m];
        float r = orc_radius_by_viewing_cos(ML->view_cos[m]);
        if (bFactor) r *= th;
        int nc = orc_lines_in_area(F, ML->x1[m], ML->y1[m], ML->x2[m], ML->y2[m], r * F->scale_factors[lvl], lvl - 1, lvl,
                                   cand, F->n);
        if (nc == 0) continue;
        if (nc > F->n) nc = F->n;
        const uint8_t *d = ML->desc + 32 * (size_t)m;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (match_of_line[idx] == -2 || match_of_line[idx] >= 0) continue;
            const int dist = orc_hamming256(d, F->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F->octave[idx]; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = F->octave[idx]; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match_of_line[bestIdx] = m;
            nmatches++;
        }
    }
    free(cand);
    return nmatches;
}
