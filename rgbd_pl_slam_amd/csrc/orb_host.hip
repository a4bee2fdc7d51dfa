// orb_host.hip -- host side of the ORB extractor: handle, geometry, tables, launch sequence, C ABI.
// Mirrors ORB_SLAM2::ORBextractor (include/ORBextractor.h:44-112): ctor tables (so@0x73050),
// ComputePyramid geometry (so@0x70430), cell tiling of ComputeKeyPointsOctTree (so@0x75fa0).
#ifndef PLF_ORB_LDS_PAD
#define PLF_ORB_LDS_PAD 0   // experiment: extra dynamic LDS per tile (occupancy sensitivity)
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "plf_common.h"
#include "orb_geom.h"

// kernels (orb_kernels.hip, orb_octree.hip)
void plf_orb_upload_constants(const int *umax16);
__global__ void k_orb_level(const uint8_t *, ptrdiff_t, ptrdiff_t, uint8_t *, uint8_t *, int, const int *, const short2 *, const int *, const short2 *, const int4 *,
                            int2 *, uint2 *, int *, int *, OrbGeom, int4);
__global__ void k_octree(const int2 *, const uint2 *, int *, uint2 *, int *, uint8_t *, uint2 *, int *, int *, int *, OrbGeom, int, int);
__global__ void k_orient_brief(const uint8_t *, const uint8_t *, const uint2 *, const int *, plf_keypoint *, uint8_t *, int *, int,
                               int *, OrbGeom, int);

struct plf_orb {
    plf_orb_params prm;
    int device;
    OrbGeom g;           // geometry of the current input size
    OrbGeom alloc;       // geometry the buffers were sized for (max_width x max_height)
    uint32_t alloc_tx, alloc_ty;
    int cur_w, cur_h;    // size the geometry / tables are currently built for
    float scale[PLF_MAX_LEVELS], inv[PLF_MAX_LEVELS], sigma2[PLF_MAX_LEVELS], invsigma2[PLF_MAX_LEVELS];
    int per_level[PLF_MAX_LEVELS];
    int umax[16];
    int capacity;
    int cap_nodes, cap_sort;
    size_t octree_lds;
    int4 taps;
    hipStream_t stream;
    // device buffers
    uint8_t *d_pyr, *d_blur, *d_quad, *d_in;
    uint2 *d_pool, *d_keys, *d_sel;
    int *d_nodeof, *d_celloff, *d_counters;  // counters: poolcnt[B*nl], selcnt[B*nl], ncand[B*nl], status[1]
    int2 *d_cellinfo;
    int4 *d_cells;
    int *d_xofs, *d_yofs;
    short2 *d_xa, *d_yb;
    plf_keypoint *d_kps;
    uint8_t *d_desc;
    int *d_nout;
    size_t in_cap;
    int last_frames;
    hipStream_t last_stream;   // stream of the most recent call
    bool last_stream_set;
    PlfStreamOrder order;
};

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

// ---- ctor tables (so@0x73050)
static void orb_tables(plf_orb *h)
{
    const int nlevels = h->prm.nlevels, nfeatures = h->prm.nfeatures;
    const double scaleFactor = (double)h->prm.scale_factor;  // member is double, ctor argument float
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        h->scale[i] = (float)((double)h->scale[i - 1] * scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < nlevels; i++) { h->inv[i] = 1.0f / h->scale[i]; h->invsigma2[i] = 1.0f / h->sigma2[i]; }
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        h->per_level[l] = cv_round_f(nDesired);
        sum += h->per_level[l];
        nDesired *= factor;
    }
    h->per_level[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    int v, v0;
    const int vmax = (int)floor(PLF_HALF_PATCH * sqrt(2.f) / 2 + 1), vmin = (int)ceil(PLF_HALF_PATCH * sqrt(2.f) / 2);
    const double hp2 = PLF_HALF_PATCH * PLF_HALF_PATCH;
    for (v = 0; v < 16; v++) h->umax[v] = 0;
    for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = PLF_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
        h->umax[v] = v0;
        ++v0;
    }
    // GaussianBlur(7x7, sigma 2) 8-bit fixed-point taps: float kernel * 256, rounded (cv::getGaussianKernel + convertTo)
    float cf[7];
    double s = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(-0.5 / (2.0 * 2.0) * x * x); s += cf[i]; }
    s = 1. / s;
    int k[7];
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * s); k[i] = cv_round_d((double)cf[i] * 256.0); }
    h->taps = make_int4(k[0], k[1], k[2], k[3]);
}

// ---- geometry for an input size; returns PLF_OK or PLF_E_BADARG
static int orb_geometry(plf_orb *h, int w, int h_, OrbGeom *g, std::vector<int4> *cells)
{
    memset(g, 0, sizeof(*g));
    g->nlevels = h->prm.nlevels; g->iniTh = h->prm.ini_th_fast; g->minTh = h->prm.min_th_fast;
    g->in_w = w; g->in_h = h_;
    size_t pyr = 0, blur = 0;
    uint32_t pool = 0, sel = 0, tabx = 0, taby = 0;
    int cellbase = 0, maxsel = 0;
    for (int l = 0; l < g->nlevels; l++) {
        OrbLevel &L = g->lv[l];
        L.w = cv_round_f((float)w * h->inv[l]);  // so@0x7051e: float multiply, cvtss2si
        L.h = cv_round_f((float)h_ * h->inv[l]);
        // cell tiling needs at least one 30-px cell in each direction; octree roots need W/H to round to >= 1
        const int minB = PLF_MINB, maxBX = L.w - PLF_MINB, maxBY = L.h - PLF_MINB;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        if (nCols < 1 || nRows < 1) return PLF_E_BADARG;
        const int nIni = (int)roundf(width / height);
        if (nIni < 1 || nIni > 8) return PLF_E_BADARG;
        if (L.w + 38 > 65000 || L.h + 38 > 65000) return PLF_E_BADARG;
        L.ppitch = L.w + 2 * PLF_EDGE;
        L.plane_off = (uint32_t)pyr;
        pyr += plf_align_up((size_t)L.ppitch * (L.h + 2 * PLF_EDGE), 256);
        L.bpitch = (int)plf_align_up(L.w, 64);
        L.blur_off = (uint32_t)blur;
        blur += plf_align_up((size_t)L.bpitch * L.h, 256);
        L.wCell = (int)ceilf(width / nCols); L.hCell = (int)ceilf(height / nRows);
        L.cell_base = cellbase;
        int nc = 0;
        uint32_t poolcap = 0;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minB + i * L.hCell);
            float maxY = iniY + L.hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minB + j * L.wCell);
                float maxX = iniX + L.wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                const int cw = (int)maxX - (int)iniX, ch = (int)maxY - (int)iniY;
                if (cw - 6 > 64 || ch - 6 > 64) return PLF_E_BADARG;  // one wave per cell row
                if (cells) cells->push_back(make_int4((int)iniX, (int)iniY, cw, ch));
                // strict 8-neighbour maxima are at most ceil(w/2)*ceil(h/2) per computed region
                const int rw = cw - 6 > 0 ? cw - 6 : 0, rh = ch - 6 > 0 ? ch - 6 : 0;
                poolcap += (uint32_t)(((rw + 1) / 2) * ((rh + 1) / 2));
                nc++;
            }
        }
        L.ncells = nc;
        cellbase += nc;
        // effective cell grid (the loop above skips cells only at the row / column ends) and the end of the last computed region
        L.ncx = 0; L.ncy = 0;
        for (int j = 0; j < nCols; j++) if (!((float)(minB + j * L.wCell) >= maxBX - 6)) L.ncx++;
        for (int i = 0; i < nRows; i++) if (!((float)(minB + i * L.hCell) >= maxBY - 3)) L.ncy++;
        if (L.ncx < 1 || L.ncy < 1 || L.ncx * L.ncy != nc) return PLF_E_BADARG;
        {
            float mx = (float)(minB + (L.ncx - 1) * L.wCell) + L.wCell + 6, my = (float)(minB + (L.ncy - 1) * L.hCell) + L.hCell + 6;
            if (mx > maxBX) mx = (float)maxBX;
            if (my > maxBY) my = (float)maxBY;
            L.rex = (int)mx - 3; L.rey = (int)my - 3;
        }
        L.tcx = (L.ncx + 1) / 2; L.tcy = (L.ncy + 1) / 2;
        if (2 * L.wCell > 250 || 2 * L.hCell > 250) return PLF_E_BADARG;   // tile-relative coordinates are packed in 8 bits
        L.quota = h->per_level[l];
        L.scale = h->scale[l];
        L.size_i = (int)(PLF_PATCH * h->scale[l]);
        L.pool_off = pool; L.pool_cap = poolcap;
        pool += (uint32_t)plf_align_up(poolcap, 64);
        L.sel_off = sel; L.sel_cap = (uint32_t)(L.quota + 32);
        sel += (uint32_t)plf_align_up(L.sel_cap, 16);
        if ((int)L.sel_cap > maxsel) maxsel = (int)L.sel_cap;
        L.tabx_off = tabx; L.taby_off = taby;
        tabx += (uint32_t)L.w; taby += (uint32_t)L.h;
    }
    g->cells_total = cellbase; g->maxsel = maxsel;
    g->pyr_stride = (uint32_t)pyr; g->blur_stride = (uint32_t)blur; g->pool_stride = pool; g->sel_stride = sel;
    return PLF_OK;
}

// cv::resize INTER_LINEAR coefficient tables for level l (source = level l-1), OpenCV 3.3 imgproc
static void resize_tables(int sw, int sh, int dw, int dh, int *xofs, short2 *xa, int *yofs, short2 *yb)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[dx].x = (short)cv_round_f((1.f - fx) * 2048.f);
        xa[dx].y = (short)cv_round_f(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        yb[dy].x = (short)cv_round_f((1.f - fy) * 2048.f);
        yb[dy].y = (short)cv_round_f(fy * 2048.f);
    }
}

static void orb_free(plf_orb *h)
{
    void *ptrs[] = {h->d_pyr, h->d_blur, h->d_quad, h->d_in, h->d_pool, h->d_keys, h->d_sel, h->d_nodeof, h->d_celloff,
                    h->d_counters, h->d_cellinfo, h->d_cells, h->d_xofs, h->d_yofs, h->d_xa, h->d_yb, h->d_kps, h->d_desc, h->d_nout};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    plf_order_free(h->order);
}

// (re)build geometry, tables and buffers for an input size
static int orb_configure(plf_orb *h, int w, int hh)
{
    if (h->cur_w == w && h->cur_h == hh) return PLF_OK;
    OrbGeom g;
    std::vector<int4> cells;
    int rc = orb_geometry(h, w, hh, &g, &cells);
    if (rc != PLF_OK) return rc;
    // buffers were sized for max_width x max_height: a smaller image must not need more
    const OrbGeom &big = h->alloc;
    if (g.pyr_stride > big.pyr_stride || g.blur_stride > big.blur_stride || g.pool_stride > big.pool_stride ||
        g.cells_total > big.cells_total || g.sel_stride > big.sel_stride)
        return PLF_E_BADARG;
    std::vector<int> xofs, yofs;
    std::vector<short2> xa, yb;
    uint32_t tx = 0, ty = 0;
    for (int l = 0; l < g.nlevels; l++) { tx += g.lv[l].w; ty += g.lv[l].h; }
    if (tx > h->alloc_tx || ty > h->alloc_ty) return PLF_E_BADARG;
    xofs.resize(tx); xa.resize(tx); yofs.resize(ty); yb.resize(ty);
    // kernels of a previous call (any stream: the caller's streams do not synchronise with the null stream) may still read the tables
    if (h->cur_w >= 0) PLF_HIP_TRY(hipDeviceSynchronize());
    for (int l = 1; l < g.nlevels; l++)
        resize_tables(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, &xofs[g.lv[l].tabx_off], &xa[g.lv[l].tabx_off],
                      &yofs[g.lv[l].taby_off], &yb[g.lv[l].taby_off]);
    PLF_HIP_TRY(hipMemcpy(h->d_xofs, xofs.data(), sizeof(int) * tx, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_xa, xa.data(), sizeof(short2) * tx, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_yofs, yofs.data(), sizeof(int) * ty, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_yb, yb.data(), sizeof(short2) * ty, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_cells, cells.data(), sizeof(int4) * cells.size(), hipMemcpyHostToDevice));
    // LDS layout of k_orb_level: maxima over all tiles of all levels.  The kernel is latency-bound per tile, so its run time is inversely
    // proportional to the tiles a CU holds (measured: 52 / 37 / 30 ms per 4096 frames for 2 / 3 / 4 resident tiles): the source rows of a tile are
    // staged in `parts` passes and the score tile only spans the cells' computed columns, so that 5 tiles fit (38.5 -> 30.8 KB at VGA).
    {
        int maxEW = 0, maxEH = 0, maxSW = 0, maxRW = 0, maxRH = 0, maxOW = 0;
        int maxSHp[5] = {0, 0, 0, 0, 0};   // [parts]: tallest staged source block when the tile rows are split into 1..4 parts
        for (int l = 0; l < g.nlevels; l++) {
            const OrbLevel &L = g.lv[l];
            for (int ty = 0; ty < L.tcy; ty++)
                for (int txi = 0; txi < L.tcx; txi++) {
                    const int cx0 = 2 * txi, cx1 = std::min(cx0 + 2, L.ncx), cy0 = 2 * ty, cy1 = std::min(cy0 + 2, L.ncy);
                    const int rx0 = PLF_EDGE + cx0 * L.wCell, rx1 = cx1 == L.ncx ? L.rex : PLF_EDGE + cx1 * L.wCell;
                    const int ry0 = PLF_EDGE + cy0 * L.hCell, ry1 = cy1 == L.ncy ? L.rey : PLF_EDGE + cy1 * L.hCell;
                    const int xs = txi == 0 ? 0 : rx0, xe = txi == L.tcx - 1 ? L.w : rx1, ys = ty == 0 ? 0 : ry0, ye = ty == L.tcy - 1 ? L.h : ry1;
                    const int ex0 = (xs & ~3) - 4, EW = ((xe - 1) & ~3) + 8 - ex0, EH = ye - ys + 6, ey0 = ys - 3;
                    if (rx1 <= rx0 || ry1 <= ry0 || xe - xs < 4 || ye - ys < 4) return PLF_E_BADARG;
                    maxEW = std::max(maxEW, EW); maxEH = std::max(maxEH, EH); maxOW = std::max(maxOW, xe - xs);
                    maxRW = std::max(maxRW, rx1 - rx0); maxRH = std::max(maxRH, ry1 - ry0);
                    if (l > 0) {
                        const OrbLevel &S = g.lv[l - 1];
                        const int lx_lo = std::max(ex0, 0), lx_hi = std::min(ex0 + EW - 1, L.w - 1);
                        const int sx_lo = xofs[L.tabx_off + lx_lo] & ~3, sx_hi = std::min(xofs[L.tabx_off + lx_hi] + 1, S.w - 1);
                        maxSW = std::max(maxSW, sx_hi - sx_lo + 1);
                        for (int parts = 1; parts <= 4; parts++)
                            for (int pi = 0; pi < parts; pi++) {
                                int s_lo, s_hi;
                                orb_part_rows(ey0, EH, parts, pi, L.h, S.h, &yofs[L.taby_off], &s_lo, &s_hi);
                                maxSHp[parts] = std::max(maxSHp[parts], s_hi - s_lo + 1);
                            }
                    }
                }
        }
        if (maxEW > 255 || maxRH > 255) return PLF_E_BADARG;   // survivors are packed as (tile column | row << 8)
        g.lds_pw = (maxEW + 3) & ~3;
        g.lds_spw = ((maxSW + 8 + 15) & ~15) + 16;   // (rows are filled 16 bytes at a time)
        g.lds_sp = (maxRW + 6 + 3) & ~3;              // score tile: columns (rx0 - ex0) & ~3 .. of the tile, i.e. the computed regions + alignment slack
        g.lds_eh = maxEH;
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t sz_p = (size_t)g.lds_pw * (maxEH + 1) + 16;
        const size_t sz_s = (size_t)g.lds_sp * maxRH + 16;
        // resize tables; the non-maximum suppression re-uses the area for its per-wave corner queues (128 uint16 each: k_orb_level phase 5)
        const size_t sz_t = std::max((size_t)(g.lds_pw + maxEH) * 8 + (size_t)(g.lds_pw / 4 + 1) * 2 + 16, (size_t)(PLF_ORB_LEVEL_THREADS / 64) * 256 + 16);
#ifndef PLF_ORB_TILES_TARGET
#define PLF_ORB_TILES_TARGET 6   // resident tiles per CU the staging passes are chosen for (round 6: 5 -> 6 with the halved survivor list)
#endif
        // FAST survivor list (aliases the staged source): half the pixels of the largest computed region, rounded up to whole rows + one group -- any half of a tile's
        // rows fits; a pass with more survivors is redone in two row halves (k_orb_level: fast_pass)
        g.lds_list_cap = maxRW * ((maxRH + 1) / 2) + 4;
        const size_t sz_list = (size_t)g.lds_list_cap * 2 + 16;
        size_t sz_a = 0;
        for (g.lds_parts = 1; g.lds_parts <= 4; g.lds_parts++) {
            sz_a = std::max((size_t)g.lds_spw * (maxSHp[g.lds_parts] + 1) + 16, sz_list);
            // five tiles per CU: 160 KB / 5 minus the 2064 static bytes (the NMS masks, round 6: one set for both passes); stop splitting when the survivor list is what is left
            if (up16(sz_p) + up16(sz_a) + up16(sz_s) + up16(sz_t) <= 160 * 1024 / PLF_ORB_TILES_TARGET - 2064 - 64 || sz_a == sz_list || g.lds_parts == 4) break;
        }
        g.lds_off_a = (int)up16(sz_p);
        g.lds_off_s = g.lds_off_a + (int)up16(sz_a);
        g.lds_off_list = g.lds_off_a;
        g.lds_off_tab = g.lds_off_s + (int)up16(sz_s);
        g.lds_total = g.lds_off_tab + (int)up16(sz_t);
        if (g.lds_total > 150 * 1024) return PLF_E_BADARG;
        if (getenv("PLF_ORB_DEBUG_LDS"))
            fprintf(stderr, "[plf] k_orb_level LDS: P %zu, SRC/LIST %zu (%d parts), S %zu, tables %zu -> %d dynamic + 2064 static (pw %d, spw %d, sp %d, eh %d, maxRW %d, maxRH %d)\n",
                    sz_p, sz_a, g.lds_parts, sz_s, sz_t, g.lds_total, g.lds_pw, g.lds_spw, g.lds_sp, g.lds_eh, maxRW, maxRH);
    }
    // keep the per-frame strides of the allocation (max size) so that buffer sizes stay valid
    g.pyr_stride = big.pyr_stride; g.blur_stride = big.blur_stride; g.pool_stride = big.pool_stride; g.sel_stride = big.sel_stride;
    const int cells_alloc = big.cells_total;
    h->g = g;
    // per-frame cell arrays (cellinfo, celloff) are indexed with the ALLOCATED cell count as frame stride
    h->g.cells_total = g.cells_total;
    (void)cells_alloc;
    h->cur_w = w; h->cur_h = hh;
    return PLF_OK;
}

extern "C" int plf_orb_create(const plf_orb_params *p, plf_orb **out)
{
    if (!p || !out) return PLF_E_BADARG;
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > PLF_MAX_LEVELS || p->nfeatures < 1 || !(p->scale_factor > 1.0f) || p->max_batch < 1 ||
        p->max_width < 1 || p->max_height < 1 || p->ini_th_fast < 1 || p->min_th_fast < 1 || p->min_th_fast > p->ini_th_fast ||
        p->ini_th_fast > 255)
        return PLF_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[plf] no HIP device available: the ORB extractor has no CPU path\n");
        return PLF_E_HIP;
    }
    if (p->device < 0 || p->device >= ndev) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(p->device));
    plf_orb *h = (plf_orb *)calloc(1, sizeof(plf_orb));
    if (!h) return PLF_E_NOMEM;
    h->prm = *p; h->device = p->device;
    orb_tables(h);
    int rc = orb_geometry(h, p->max_width, p->max_height, &h->alloc, nullptr);
    if (rc != PLF_OK) { free(h); return rc; }
    h->g = h->alloc;
    h->capacity = p->nfeatures + 4 * p->nlevels;
    int maxq = 0;
    for (int l = 0; l < p->nlevels; l++) maxq = h->per_level[l] > maxq ? h->per_level[l] : maxq;
    h->cap_nodes = maxq + 32;
    if (h->cap_nodes > 4000) { free(h); return PLF_E_BADARG; }  // node positions are packed in 12 bits
    h->cap_sort = 1;
    while (h->cap_sort < h->cap_nodes) h->cap_sort <<= 1;
    h->octree_lds = sizeof(unsigned long long) * h->cap_sort + (size_t)h->cap_nodes * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 4 + 16 + 4 + 4 + 4) + 4 * 260;
    if (h->octree_lds > 160 * 1024) { free(h); return PLF_E_BADARG; }
    const size_t B = (size_t)p->max_batch;
    const OrbGeom &g = h->g;
    uint32_t tx = 0, ty = 0;
    for (int l = 0; l < g.nlevels; l++) { tx += g.lv[l].w; ty += g.lv[l].h; }
    h->alloc_tx = tx; h->alloc_ty = ty;
#define ALLOC(ptr, bytes)                                                             \
    do {                                                                              \
        if (hipMalloc((void **)&(ptr), (bytes) > 0 ? (bytes) : 256) != hipSuccess) { \
            orb_free(h); free(h); return PLF_E_NOMEM;                                 \
        }                                                                             \
    } while (0)
    ALLOC(h->d_pyr, B * g.pyr_stride);
    ALLOC(h->d_blur, B * g.blur_stride);
    ALLOC(h->d_pool, B * g.pool_stride * sizeof(uint2));
    ALLOC(h->d_keys, B * g.pool_stride * sizeof(uint2));
    ALLOC(h->d_nodeof, B * g.pool_stride * sizeof(int));
    ALLOC(h->d_quad, B * g.pool_stride);
    ALLOC(h->d_sel, B * g.sel_stride * sizeof(uint2));
    ALLOC(h->d_cellinfo, B * g.cells_total * sizeof(int2));
    ALLOC(h->d_celloff, B * g.cells_total * sizeof(int));
    ALLOC(h->d_cells, (size_t)g.cells_total * sizeof(int4));
    ALLOC(h->d_counters, (B * g.nlevels * 3 + 16) * sizeof(int));
    ALLOC(h->d_xofs, tx * sizeof(int)); ALLOC(h->d_xa, tx * sizeof(short2));
    ALLOC(h->d_yofs, ty * sizeof(int)); ALLOC(h->d_yb, ty * sizeof(short2));
    ALLOC(h->d_kps, B * h->capacity * sizeof(plf_keypoint));
    ALLOC(h->d_desc, B * h->capacity * 32);
    ALLOC(h->d_nout, B * sizeof(int));
    h->in_cap = B * (size_t)p->max_width * p->max_height;
    ALLOC(h->d_in, h->in_cap);
#undef ALLOC
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { orb_free(h); free(h); return PLF_E_HIP; }
    plf_orb_upload_constants(h->umax);
    (void)hipFuncSetAttribute((const void *)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->octree_lds);
    (void)hipFuncSetAttribute((const void *)k_orb_level, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();
    h->cur_w = -1; h->cur_h = -1;
    rc = orb_configure(h, p->max_width, p->max_height);
    if (rc != PLF_OK) { orb_free(h); free(h); return rc; }
    if (hipDeviceSynchronize() != hipSuccess) { orb_free(h); free(h); return PLF_E_HIP; }
    *out = h;
    return PLF_OK;
}

extern "C" void plf_orb_destroy(plf_orb *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    orb_free(h);
    free(h);
}

extern "C" int plf_orb_get_tables(const plf_orb *h, int32_t *nlevels, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                                  int32_t *per_level)
{
    if (!h) return PLF_E_BADARG;
    const int n = h->prm.nlevels;
    if (nlevels) *nlevels = n;
    for (int i = 0; i < n; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->invsigma2[i];
        if (per_level) per_level[i] = h->per_level[i];
    }
    return PLF_OK;
}

extern "C" int plf_orb_capacity(const plf_orb *h) { return h ? h->capacity : PLF_E_BADARG; }

// enqueue the whole extractor on `s` for frames already resident at d_gray
static int orb_enqueue(plf_orb *h, const uint8_t *d_gray, int n_frames, ptrdiff_t pitch, ptrdiff_t fstride, plf_keypoint *d_kps,
                       uint8_t *d_desc, int *d_nout, int capacity, hipStream_t s)
{
    const OrbGeom &g = h->g;
    const int B = n_frames, nl = g.nlevels;
    int *poolcnt = h->d_counters, *selcnt = h->d_counters + (size_t)h->prm.max_batch * nl,
        *ncand = h->d_counters + 2 * (size_t)h->prm.max_batch * nl, *status = h->d_counters + 3 * (size_t)h->prm.max_batch * nl;
    PLF_HIP_TRY(hipMemsetAsync(h->d_counters, 0, (3 * (size_t)h->prm.max_batch * nl + 16) * sizeof(int), s));
    // one fused launch per level (k_orb_level: pyramid plane + blur + FAST score / NMS / cell candidates from one LDS tile); level l is
    // resized from level l-1, so the launches are stream-ordered
    for (int l = 0; l < nl; l++) {
        const OrbLevel &L = g.lv[l];
        hipLaunchKernelGGL(k_orb_level, dim3(L.tcx * L.tcy, B), dim3(PLF_ORB_LEVEL_THREADS), (size_t)g.lds_total + PLF_ORB_LDS_PAD, s, d_gray, pitch, fstride, h->d_pyr, h->d_blur, l, h->d_xofs, h->d_xa,
                           h->d_yofs, h->d_yb, h->d_cells, h->d_cellinfo, h->d_pool, poolcnt, status, g, h->taps);
    }
// k_octree is a chain of LDS sweeps and barriers over <= 250 nodes: latency-bound per workgroup.  128 threads: twice as many independent workgroups per wave slot
    // (4.9 -> 3.0 ms per 4096 frames solo, and two of them fit in the one-wave-per-SIMD room next to the region-growing kernel)
#ifndef PLF_OCTREE_THREADS
#define PLF_OCTREE_THREADS 128
#endif
    hipLaunchKernelGGL(k_octree, dim3(nl, B), dim3(PLF_OCTREE_THREADS), h->octree_lds, s, h->d_cellinfo, h->d_pool, h->d_celloff, h->d_keys,
                       h->d_nodeof, h->d_quad, h->d_sel, selcnt, ncand, status, g, h->cap_nodes, h->cap_sort);
    int slots = 0;   // at most sum of the per-level selection caps, and never more than the caller can take
    for (int l = 0; l < nl; l++) slots += (int)g.lv[l].sel_cap;
    if (slots > capacity) slots = capacity;
    hipLaunchKernelGGL(k_orient_brief, dim3(8 * slots, (B + 7) / 8), dim3(64), 0, s, h->d_pyr, h->d_blur, h->d_sel, selcnt, d_kps, d_desc,
                       d_nout, capacity, status, g, B);
    PLF_HIP_TRY(hipGetLastError());
    h->last_frames = B;
    return PLF_OK;
}

extern "C" int plf_orb_extract_batch(plf_orb *h, const uint8_t *gray, int32_t in_mem, int32_t n_frames, int32_t width, int32_t height,
                                     ptrdiff_t pitch, ptrdiff_t frame_stride, plf_keypoint *kps, uint8_t *desc, int32_t *n_out,
                                     int32_t out_mem, int32_t capacity, void *stream)
{
    if (!h) return PLF_E_BADARG;
    if (!gray || width <= 0 || height <= 0 || n_frames <= 0) return PLF_E_EMPTY;  // reference: silent return, so@0x76dda
    if (n_frames > h->prm.max_batch || width > h->prm.max_width || height > h->prm.max_height || pitch < width || !kps || !desc ||
        !n_out || capacity < 1)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    int rc = orb_configure(h, width, height);
    if (rc != PLF_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    // handle-owned scratch is ordered by the stream of the previous call: a call on another stream waits for it first (include/plf.h, "Streams")
    plf_order_begin(h->order, s);
    PlfOrderGuard order_guard_{h->order, s};
    h->last_stream = s; h->last_stream_set = true;
    const uint8_t *d_gray = gray;
    ptrdiff_t dpitch = pitch, dfstride = frame_stride;
    if (in_mem == PLF_MEM_HOST) {
        // pack rows tightly into the staging buffer
        dpitch = width; dfstride = (ptrdiff_t)width * height;
        for (int f = 0; f < n_frames; f++)
            PLF_HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)f * dfstride, dpitch, gray + (size_t)f * frame_stride, pitch, width, height,
                                         hipMemcpyHostToDevice, s));
        d_gray = h->d_in;
    }
    const bool host_out = out_mem == PLF_MEM_HOST;
    const int cap_dev = host_out ? h->capacity : capacity;
    plf_keypoint *d_kps = host_out ? h->d_kps : kps;
    uint8_t *d_desc = host_out ? h->d_desc : desc;
    int *d_nout = host_out ? h->d_nout : n_out;
    rc = orb_enqueue(h, d_gray, n_frames, dpitch, dfstride, d_kps, d_desc, d_nout, cap_dev, s);
    if (rc != PLF_OK) return rc;
    if (!host_out && in_mem == PLF_MEM_DEVICE) return PLF_OK;  // fully asynchronous
    int status = 0;
    PLF_HIP_TRY(hipMemcpyAsync(&status, h->d_counters + 3 * (size_t)h->prm.max_batch * h->g.nlevels, sizeof(int), hipMemcpyDeviceToHost, s));
    if (host_out) {
        std::vector<int> cnt(n_frames);
        PLF_HIP_TRY(hipMemcpyAsync(cnt.data(), d_nout, sizeof(int) * n_frames, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
        int ret = PLF_OK;
        for (int f = 0; f < n_frames; f++) {
            int n = cnt[f];
            if (n > capacity) { n = capacity; ret = PLF_E_CAPACITY; }
            n_out[f] = n;
            if (n > 0) {
                PLF_HIP_TRY(hipMemcpyAsync(kps + (size_t)f * capacity, d_kps + (size_t)f * cap_dev, sizeof(plf_keypoint) * n, hipMemcpyDeviceToHost, s));
                PLF_HIP_TRY(hipMemcpyAsync(desc + (size_t)f * capacity * 32, d_desc + (size_t)f * cap_dev * 32, (size_t)32 * n, hipMemcpyDeviceToHost, s));
            }
        }
        PLF_HIP_TRY(hipStreamSynchronize(s));
        if (status & 5) return PLF_E_HIP;  // internal pool/selection overflow: cannot happen with the sizing rules
        if (status & 2) ret = PLF_E_CAPACITY;
        return ret;
    }
    PLF_HIP_TRY(hipStreamSynchronize(s));
    return (status & 5) ? PLF_E_HIP : ((status & 2) ? PLF_E_CAPACITY : PLF_OK);
}

// batch driver (batch_host.hip): the status word of the batch just enqueued on `s`, copied to pinned host memory in stream order
int plf_orb_status_async(plf_orb *h, int32_t *host_dst, hipStream_t s)
{
    PLF_HIP_TRY(hipMemcpyAsync(host_dst, h->d_counters + 3 * (size_t)h->prm.max_batch * h->g.nlevels, sizeof(int), hipMemcpyDeviceToHost, s));
    return PLF_OK;
}

extern "C" int plf_orb_extract(plf_orb *h, const uint8_t *gray, int32_t width, int32_t height, ptrdiff_t pitch, plf_keypoint *kps,
                               uint8_t *desc, int32_t capacity, int32_t *n_out)
{
    return plf_orb_extract_batch(h, gray, PLF_MEM_HOST, 1, width, height, pitch, (ptrdiff_t)pitch * height, kps, desc, n_out,
                                 PLF_MEM_HOST, capacity, nullptr);
}

extern "C" int plf_orb_get_pyramid_level(plf_orb *h, int32_t frame, int32_t level, uint8_t *dst, int32_t *level_w, int32_t *level_h)
{
    if (!h || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->g.nlevels) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    const OrbLevel &L = h->g.lv[level];
    if (level_w) *level_w = L.w;
    if (level_h) *level_h = L.h;
    if (dst) {
        PLF_HIP_TRY(hipDeviceSynchronize());
        PLF_HIP_TRY(hipMemcpy(dst, h->d_pyr + (size_t)frame * h->g.pyr_stride + L.plane_off, (size_t)L.ppitch * (L.h + 2 * PLF_EDGE),
                              hipMemcpyDeviceToHost));
    }
    return PLF_OK;
}

extern "C" int plf_orb_get_blurred_level(plf_orb *h, int32_t frame, int32_t level, uint8_t *dst)
{
    if (!h || !dst || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->g.nlevels) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    const OrbLevel &L = h->g.lv[level];
    PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy2D(dst, L.w, h->d_blur + (size_t)frame * h->g.blur_stride + L.blur_off, L.bpitch, L.w, L.h, hipMemcpyDeviceToHost));
    return PLF_OK;
}

extern "C" int plf_orb_get_candidates(plf_orb *h, int32_t frame, int32_t level, float *xyr, int32_t capacity, int32_t *n_out)
{
    if (!h || !n_out || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->g.nlevels) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    PLF_HIP_TRY(hipDeviceSynchronize());
    const OrbLevel &L = h->g.lv[level];
    int n = 0;
    PLF_HIP_TRY(hipMemcpy(&n, h->d_counters + 2 * (size_t)h->prm.max_batch * h->g.nlevels + (size_t)frame * h->g.nlevels + level, sizeof(int),
                          hipMemcpyDeviceToHost));
    *n_out = n;
    if (xyr && n > 0) {
        const int m = n < capacity ? n : capacity;
        std::vector<uint2> k(m);
        PLF_HIP_TRY(hipMemcpy(k.data(), h->d_keys + (size_t)frame * h->g.pool_stride + L.pool_off, sizeof(uint2) * m, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) {
            xyr[3 * i] = (float)(k[i].x & 0xFFFF); xyr[3 * i + 1] = (float)(k[i].x >> 16); xyr[3 * i + 2] = (float)k[i].y;
        }
    }
    return PLF_OK;
}
