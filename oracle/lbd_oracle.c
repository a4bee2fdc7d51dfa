/*
 * lbd_oracle.c -- CPU restatement of the LBD line descriptor and of LineSegment::ExtractLineSegment.
 * TEST INFRASTRUCTURE ONLY (same rules as orb_oracle.c).
 *
 * Reference entry: LineSegment::ExtractLineSegment(img, keylines, ldesc, lineFunctions, scale, numOctaves)
 * include/ExtractLineSegment.h:38; comparator sort_lines_by_response include/auxiliar.h:67-72.  The
 * body is absent from /root/reference; it is restated from the PL-SLAM family this fork derives from
 * (SURVEY.md 8a-8): LSDDetector::detect -> keep the N strongest by response -> BinaryDescriptor::
 * compute -> normalised line equations.  The descriptor itself is opencv_contrib 3.3
 * line_descriptor/src/binary_descriptor.cpp (Zhang & Koch, "An efficient and robust line segment
 * matching approach based on LBD descriptor and pairwise geometric consistency", JVCI 2013), not
 * vendored -> PARITY UNPINNED.
 * Tie order of std::sort(sort_lines_by_response) is unspecified upstream; this restatement uses the
 * stable order (response descending, then detection index), one of the valid outcomes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define NUM_OF_BANDS 9
#define WIDTH_OF_BAND 7

static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

/* cv::Sobel(img, d, CV_16S, dx, dy, 3), BORDER_REFLECT_101 -- exact integer arithmetic */
void orc_sobel3_16s(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int16_t *dxImg, int16_t *dyImg)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = gray + (ptrdiff_t)reflect101(y - 1, h) * pitch;
        const uint8_t *r1 = gray + (ptrdiff_t)y * pitch;
        const uint8_t *r2 = gray + (ptrdiff_t)reflect101(y + 1, h) * pitch;
        for (int x = 0; x < w; x++) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            int gx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
            int gy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
            dxImg[(size_t)y * w + x] = (int16_t)gx;
            dyImg[(size_t)y * w + x] = (int16_t)gy;
        }
    }
}

static const int combinations[32][2] = {
    {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
    {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

static uint8_t binary_conversion(const float *f1, const float *f2)
{
    uint8_t result = 0;
    for (int i = 0; i < 8; i++)
        if (f1[i] > f2[i]) result = (uint8_t)(result + (uint8_t)(8 * (8 - i - 1))); /* upstream quirk: 8*(7-i), not 1<<(7-i) */
    return result;
}

void orc_lbd_gauss_coefs(double *gaussCoefL /*21*/, double *gaussCoefG /*63*/)
{
    double u = (WIDTH_OF_BAND * 3 - 1) / 2;         /* integer division: 10 */
    double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;     /* integer division: 7 */
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < WIDTH_OF_BAND * 3; i++) { double dis = i - u; gaussCoefL[i] = exp(dis * dis * invsigma2); }
    u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;     /* 31 */
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; i++) { double dis = i - u; gaussCoefG[i] = exp(dis * dis * invsigma2); }
}

/* BinaryDescriptor::computeLBD for one line (octave 0).  fdesc: 72 floats. */
static void lbd_one(const int16_t *pdxImg, const int16_t *pdyImg, int realWidth, int realHeight, const orc_keyline *kl,
                    const double *gaussCoefL, const double *gaussCoefG, float *desVec)
{
    float dL[2], dO[2];
    const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS);
    const short descriptor_size = NUM_OF_BANDS * 8;
    float pgdLRowSum, ngdLRowSum, pgdL2RowSum, ngdL2RowSum, pgdORowSum, ngdORowSum, pgdO2RowSum, ngdO2RowSum;
    float pgdLBandSum[NUM_OF_BANDS] = {0}, ngdLBandSum[NUM_OF_BANDS] = {0}, pgdL2BandSum[NUM_OF_BANDS] = {0},
          ngdL2BandSum[NUM_OF_BANDS] = {0}, pgdOBandSum[NUM_OF_BANDS] = {0}, ngdOBandSum[NUM_OF_BANDS] = {0},
          pgdO2BandSum[NUM_OF_BANDS] = {0}, ngdO2BandSum[NUM_OF_BANDS] = {0};
    const short halfHeight = (short)((heightOfLSP - 1) / 2);
    const short imageWidth = (short)(realWidth - 1), imageHeight = (short)(realHeight - 1);
    const short lengthOfLSP = (short)kl->numOfPixels;
    const short halfWidth = (short)((lengthOfLSP - 1) / 2);
    const float lineMiddlePointX = (float)(0.5 * (kl->sPointInOctaveX + kl->ePointInOctaveX));
    const float lineMiddlePointY = (float)(0.5 * (kl->sPointInOctaveY + kl->ePointInOctaveY));
    dL[0] = (float)cos((double)kl->angle); /* ::cos(double) under GCC 5.4; osl.direction = kl.angle */
    dL[1] = (float)sin((double)kl->angle);
    dO[0] = -dL[1];
    dO[1] = dL[0];
    float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
    float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
    for (short hID = 0; hID < heightOfLSP; hID++) {
        float sCorX = sCorX0, sCorY = sCorY0;
        pgdLRowSum = 0; ngdLRowSum = 0; pgdORowSum = 0; ngdORowSum = 0;
        for (short wID = 0; wID < lengthOfLSP; wID++) {
            short tempCor = (short)roundf(sCorX);
            short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
            tempCor = (short)roundf(sCorY);
            short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
            short dx = pdxImg[yCor * realWidth + xCor];
            short dy = pdyImg[yCor * realWidth + xCor];
            float gDL = dx * dL[0] + dy * dL[1];
            float gDO = dx * dO[0] + dy * dO[1];
            if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
            if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
            sCorX += dL[0];
            sCorY += dL[1];
        }
        sCorX0 -= dL[1];
        sCorY0 += dL[0];
        float coefInGaussion = (float)gaussCoefG[hID];
        pgdLRowSum = coefInGaussion * pgdLRowSum;
        ngdLRowSum = coefInGaussion * ngdLRowSum;
        pgdL2RowSum = pgdLRowSum * pgdLRowSum;
        ngdL2RowSum = ngdLRowSum * ngdLRowSum;
        pgdORowSum = coefInGaussion * pgdORowSum;
        ngdORowSum = coefInGaussion * ngdORowSum;
        pgdO2RowSum = pgdORowSum * pgdORowSum;
        ngdO2RowSum = ngdORowSum * ngdORowSum;
        short bandID = (short)(hID / WIDTH_OF_BAND);
        for (int pass = 0; pass < 3; pass++) {
            short b; int ci;
            if (pass == 0) { b = bandID; ci = hID % WIDTH_OF_BAND + WIDTH_OF_BAND; }
            else if (pass == 1) { b = (short)(bandID - 1); ci = hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND; if (b < 0) continue; }
            else { b = (short)(bandID + 1); ci = hID % WIDTH_OF_BAND; if (b >= NUM_OF_BANDS) continue; }
            coefInGaussion = (float)gaussCoefL[ci];
            pgdLBandSum[b] += coefInGaussion * pgdLRowSum;
            ngdLBandSum[b] += coefInGaussion * ngdLRowSum;
            pgdL2BandSum[b] += coefInGaussion * coefInGaussion * pgdL2RowSum;
            ngdL2BandSum[b] += coefInGaussion * coefInGaussion * ngdL2RowSum;
            pgdOBandSum[b] += coefInGaussion * pgdORowSum;
            ngdOBandSum[b] += coefInGaussion * ngdORowSum;
            pgdO2BandSum[b] += coefInGaussion * coefInGaussion * pgdO2RowSum;
            ngdO2BandSum[b] += coefInGaussion * coefInGaussion * ngdO2RowSum;
        }
    }
    const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
    for (short bandID = 0; bandID < NUM_OF_BANDS; bandID++) {
        float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
        short desID = (short)(bandID * 8);
        float temp = pgdLBandSum[bandID] * invN;
        desVec[desID] = temp;
        desVec[desID + 4] = sqrtf(pgdL2BandSum[bandID] * invN - temp * temp);
        temp = ngdLBandSum[bandID] * invN;
        desVec[desID + 1] = temp;
        desVec[desID + 5] = sqrtf(ngdL2BandSum[bandID] * invN - temp * temp);
        temp = pgdOBandSum[bandID] * invN;
        desVec[desID + 2] = temp;
        desVec[desID + 6] = sqrtf(pgdO2BandSum[bandID] * invN - temp * temp);
        temp = ngdOBandSum[bandID] * invN;
        desVec[desID + 3] = temp;
        desVec[desID + 7] = sqrtf(ngdO2BandSum[bandID] * invN - temp * temp);
    }
    float tempM = 0, tempS = 0;
    for (int base = 0; base < NUM_OF_BANDS; base++) {
        const float *d = desVec + 8 * base;
        tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
        tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
    }
    tempM = 1 / sqrtf(tempM);
    tempS = 1 / sqrtf(tempS);
    for (int base = 0; base < NUM_OF_BANDS; base++) {
        float *d = desVec + 8 * base;
        d[0] = d[0] * tempM; d[1] = d[1] * tempM; d[2] = d[2] * tempM; d[3] = d[3] * tempM;
        d[4] = d[4] * tempS; d[5] = d[5] * tempS; d[6] = d[6] * tempS; d[7] = d[7] * tempS;
    }
    for (short i = 0; i < descriptor_size; i++)
        if ((double)desVec[i] > 0.4) desVec[i] = (float)0.4;
    float temp = 0;
    for (short i = 0; i < descriptor_size; i++) temp += desVec[i] * desVec[i];
    temp = 1 / sqrtf(temp);
    for (short i = 0; i < descriptor_size; i++) desVec[i] = desVec[i] * temp;
}

/* BinaryDescriptor::compute: desc n x 32 (row = keyline order), fdesc optional n x 72 */
/* sobel_input: ORC_LBD_BLURRED = BinaryDescriptor::computeGaussianPyramid's octave 0, i.e. cv::GaussianBlur(image.clone(), Size(5, 5), 1)
 * (8U: 8-bit fixed-point separable path, taps 14 63 103 63 14) before the two cv::Sobel calls -- believed to be what opencv_contrib 3.3
 * does (the same blur line is commented out in LSDDetector's pyramid, which is why the DETECTOR sees the raw image);
 * ORC_LBD_RAW = Sobel on the image as handed in (round-1 behaviour of this repo). */
void orc_lbd_compute_ex(const uint8_t *gray, int w, int h, ptrdiff_t pitch, const orc_keyline *kl, int n, uint8_t *desc,
                        float *fdesc, int sobel_input)
{
    if (n <= 0) return;
    int16_t *dxImg = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h), *dyImg = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
    if (sobel_input == ORC_LBD_BLURRED) {
        uint8_t *bl = (uint8_t *)malloc((size_t)w * h);
        orc_gaussian_blur_8u(gray, pitch, bl, w, w, h, 5, 1.0);
        orc_sobel3_16s(bl, w, h, w, dxImg, dyImg);
        free(bl);
    } else
        orc_sobel3_16s(gray, w, h, pitch, dxImg, dyImg);
    double gL[WIDTH_OF_BAND * 3], gG[NUM_OF_BANDS * WIDTH_OF_BAND];
    orc_lbd_gauss_coefs(gL, gG);
    for (int k = 0; k < n; k++) {
        float dv[NUM_OF_BANDS * 8];
        lbd_one(dxImg, dyImg, w, h, &kl[k], gL, gG, dv);
        if (fdesc) memcpy(fdesc + (size_t)k * 72, dv, sizeof(dv));
        for (int c = 0; c < 32; c++)
            desc[(size_t)k * 32 + c] = binary_conversion(&dv[8 * combinations[c][0]], &dv[8 * combinations[c][1]]);
    }
    free(dxImg); free(dyImg);
}

void orc_lbd_compute(const uint8_t *gray, int w, int h, ptrdiff_t pitch, const orc_keyline *kl, int n, uint8_t *desc, float *fdesc)
{
    orc_lbd_compute_ex(gray, w, h, pitch, kl, n, desc, fdesc, ORC_LBD_BLURRED);
}

/* LineSegment::ExtractLineSegment.  nkeep = number of lines kept (lsdNFeatures of the fork; not
 * configurable in the reference YAML -- BASELINE configs use 100/200/400).  Returns line count. */
int orc_line_extract_ex(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nkeep, int seed_order, orc_keyline *out,
                        uint8_t *desc, double *lineeq, int cap, int *ndetected, int sobel_input)
{
    int segcap = 1 << 16;
    float *segs = (float *)malloc(sizeof(float) * 4 * segcap);
    ORC_T0(t_lsd);
    int ns = orc_lsd_detect(gray, w, h, pitch, seed_order, segs, segcap, NULL);
    ORC_T1(t_lsd, ORC_ST_LSD);
    if (ns > segcap) ns = segcap;
    if (ndetected) *ndetected = ns;
    orc_keyline *kl = (orc_keyline *)malloc(sizeof(orc_keyline) * (ns > 0 ? ns : 1));
    orc_keylines_from_segments(segs, ns, w, h, kl);
    free(segs);
    int n = ns;
    if (ns > nkeep) {
        /* stable selection sort by response descending (ties: detection order) */
        int *order = (int *)malloc(sizeof(int) * ns);
        for (int i = 0; i < ns; i++) order[i] = i;
        /* insertion-based stable sort (ns is a few hundred) */
        for (int i = 1; i < ns; i++) {
            int t = order[i], j = i - 1;
            while (j >= 0 && kl[order[j]].response < kl[t].response) { order[j + 1] = order[j]; j--; }
            order[j + 1] = t;
        }
        orc_keyline *s = (orc_keyline *)malloc(sizeof(orc_keyline) * nkeep);
        for (int i = 0; i < nkeep; i++) { s[i] = kl[order[i]]; s[i].class_id = i; }
        memcpy(kl, s, sizeof(orc_keyline) * nkeep);
        free(s); free(order);
        n = nkeep;
    }
    if (n > cap) n = cap;
    ORC_T0(t_lbd);
    orc_lbd_compute_ex(gray, w, h, pitch, kl, n, desc, NULL, sobel_input);
    ORC_T1(t_lbd, ORC_ST_LBD);
    for (int i = 0; i < n; i++) {
        out[i] = kl[i];
        double sx = kl[i].startPointX, sy = kl[i].startPointY, ex = kl[i].endPointX, ey = kl[i].endPointY;
        double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex; /* sp.cross(ep) */
        double nrm = sqrt(l0 * l0 + l1 * l1);
        lineeq[3 * i] = l0 / nrm; lineeq[3 * i + 1] = l1 / nrm; lineeq[3 * i + 2] = l2 / nrm;
    }
    free(kl);
    return n;
}

int orc_line_extract(const uint8_t *gray, int w, int h, ptrdiff_t pitch, int nkeep, int seed_order, orc_keyline *out,
                     uint8_t *desc, double *lineeq, int cap, int *ndetected)
{
    return orc_line_extract_ex(gray, w, h, pitch, nkeep, seed_order, out, desc, lineeq, cap, ndetected, ORC_LBD_BLURRED);
}
