// orb_geom.h -- per-handle geometry of the ORB pipeline shared by host code and kernels.
#pragma once
#include <stdint.h>

#define PLF_MAX_LEVELS 12
#define PLF_EDGE 19        // EDGE_THRESHOLD of the reference (so@0x7609c: 19 - 3 = 16)
#define PLF_MINB 16        // minBorderX/Y
#define PLF_HALF_PATCH 15
#define PLF_PATCH 31
#ifndef PLF_ORB_LEVEL_THREADS
#define PLF_ORB_LEVEL_THREADS 256   // workgroup size of k_orb_level (one tile of 2 x 2 cells): 4 waves x 72 VGPRs still fit a CU next to 16 region-growing waves
#endif

struct OrbLevel {
    int w, h;            // level image (interior) size
    int ppitch;          // padded plane pitch = w + 38
    uint32_t plane_off;  // byte offset of the padded plane inside one frame's pyramid block
    int bpitch;          // pitch of the blurred / score planes (multiple of 64)
    uint32_t blur_off;   // byte offset inside one frame's blur block (score block uses the same offsets)
    int ncells;          // FAST cells of this level
    int cell_base;       // index of the first cell in the cell table
    int wCell, hCell;
    int quota;           // mnFeaturesPerLevel[l]
    float scale;         // mvScaleFactor[l]
    int size_i;          // int(31 * scale), so@0x76996
    uint32_t pool_off;   // candidate pool: first entry inside one frame's pool block
    uint32_t pool_cap;   // entries
    uint32_t sel_off;    // selected keypoints: first entry inside one frame's sel block
    uint32_t sel_cap;    // quota + 8
    uint32_t tabx_off, taby_off;  // resize coefficient tables (levels >= 1)
    // k_orb_level: the effective cell grid (cells skipped by the reference's loop are at the row / column ends), tiles of 2 x 2 cells, and the
    // end of the last cell's computed region (cell sub-image minus its 3-pixel frame; the regions of neighbouring cells abut)
    int ncx, ncy, tcx, tcy, rex, rey;
};

struct OrbGeom {
    int nlevels;
    int iniTh, minTh;
    int in_w, in_h;
    int cells_total;
    int maxsel;            // max over levels of sel_cap
    uint32_t pyr_stride;   // bytes per frame
    uint32_t blur_stride;  // bytes per frame
    uint32_t pool_stride;  // entries per frame
    uint32_t sel_stride;   // entries per frame
    // LDS layout of k_orb_level (bytes; maxima over the levels): pixel tile pitch, source tile pitch, score tile pitch, byte offsets
    int lds_pw, lds_spw, lds_sp, lds_eh, lds_off_a, lds_off_s, lds_off_list, lds_off_tab, lds_total;
    int lds_list_cap;    // entries of the FAST survivor list: half the pixels of the largest computed region of a tile (k_orb_level redoes denser passes in two row halves)
    int lds_parts;       // the source rows of a tile are staged in this many passes (rows of the tile split evenly): bounds the staging buffer
    OrbLevel lv[PLF_MAX_LEVELS];
};

// Source rows (of level l - 1) behind part `pi` of `parts` of a tile's rows [ey0, ey0 + EH) of level l: rows of the tile are split evenly, rows outside
// the level are REFLECT_101 mirrors (they fall inside the part's clamped range: a tile keeps >= 4 rows inside).  yofs: the level's row table.
// Used by the host (LDS sizing) and by k_orb_level (staging) -- the same arithmetic on both sides.
#ifdef __HIPCC__
__host__ __device__
#endif
static inline void orb_part_rows(int ey0, int EH, int parts, int pi, int H, int srcH, const int *yofs, int *s_lo, int *s_hi)
{
    const int e0 = pi * EH / parts, e1 = (pi + 1) * EH / parts;
    const int a = ey0 + e0, b = ey0 + e1 - 1;
    const int ra = a < 0 ? -a : (a >= H ? 2 * (H - 1) - a : a), rb = b < 0 ? -b : (b >= H ? 2 * (H - 1) - b : b);
    int lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
    if (a < 0 && b >= 0) lo = 0;
    if (a <= H - 1 && b > H - 1) hi = H - 1;
    int sl = yofs[lo], sh = yofs[hi] + 1;
    sl = sl < 0 ? 0 : (sl > srcH - 1 ? srcH - 1 : sl);
    sh = sh < 0 ? 0 : (sh > srcH - 1 ? srcH - 1 : sh);
    *s_lo = sl; *s_hi = sh;
}
