// lsd_geom.h -- geometry / constants of the line pipeline shared by host code and kernels.
#pragma once
#include <stdint.h>

struct LsdRect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

struct LsdTaps { double k[7]; };  // cv::getGaussianKernel(7, 0.75, CV_64F)

struct LbdCoefs { float gL[21]; float gG[63]; };  // (float) of the double LBD band / global Gaussian weights

struct LsdGeom {
    int w, h;             // input image
    int sw, sh;           // 0.8x scaled image
    int xmax;             // first dx whose source column is clamped (cv::resize HResize split)
    uint32_t full_stride; // elements per frame of the full-resolution double maps
    uint32_t s_stride;    // elements per frame of the scaled maps
    double rho, prec, p, log_nt;
    int min_reg_size;
    int rcap;             // region-list entries kept in LDS (one spare word follows)
    int rect_cap;         // rectangles / segments per frame
    int nkeep;            // lines kept after the response sort
    int sort_cap;         // power of two >= rect_cap
};
