// lsd_geom.h -- geometry / constants of the line pipeline shared by host code and kernels.
#pragma once
#include <cstddef>
#include <stdint.h>

struct LsdRect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

struct LsdTaps { double k[7]; };  // cv::getGaussianKernel(7, 0.75, CV_64F)

struct LbdCoefs { float gL[21]; float gG[63]; };  // (float) of the double LBD band / global Gaussian weights

// LDS of one wave of the large-batch region kernel (k_lsd_regions2): the first words of the region list (rcap + the mailbox word) and 1 KB for the parked seed chunk
// and 768 bytes of staging for region2rect's ordered sums (3072 + 1024 + 768 bytes: 32 waves = 152 KB per CU -- eight region waves per SIMD, see k_lsd_regions2)
#ifndef PLF_LSD_WAVE_LIST
#define PLF_LSD_WAVE_LIST 3072
#endif
#define PLF_LSD_WAVE_LDS (PLF_LSD_WAVE_LIST + 1024 + 768)

// k_lsd_pre: a workgroup of PRE_NT threads produces a PRE_TW x PRE_TH tile of the 0.8x scaled image; the tile (+1 column / row for the 2x2 gradient) needs at most
// PRE_SC x PRE_SR blurred source pixels (checked on the host for the actual geometry) and 6 more rows of the row pass
#define PRE_TW 64
#ifndef PRE_TH
#define PRE_TH 24   // 16 / 24 rows: 16.9 / 15.6 ms per 8192 frames (the row pass reads 6 extra rows per tile, the column pass works in bands of 8 source rows); 32 rows x 1024 threads: 21.5
#endif
#define PRE_SC 88
#ifndef PRE_SR
#define PRE_SR (PRE_TH * 5 / 4 + 6)
#endif
#ifndef PRE_NT
#define PRE_NT 512   // 8 waves share the tile's LDS (4 tiles per CU = 8 waves per SIMD); 256 / 384 / 448 / 512 / 1024 threads: 16.3 / 14.7 / 15.1 / 13.5 / 18.9 ms per 4096 frames (round 2)
#endif

// k_blur5_sobel3: 256 threads produce a BS_TW x BS_TH tile of (dx, dy)
#define BS_TW 64
#ifndef BS_TH
#define BS_TH 32
#endif

struct LsdGeom {
    int w, h;             // input image
    int sw, sh;           // 0.8x scaled image
    int xmax;             // first dx whose source column is clamped (cv::resize HResize split)
    uint32_t full_stride; // elements per frame of the full-resolution double maps
    uint32_t s_stride;    // elements per frame of the scaled maps
    double rho, prec, p, log_nt;
    double rho_q;         // largest double q with sqrt(q) <= rho: a gradient has a level-line angle iff its squared norm / 4 exceeds it (k_lsd_pre)
    int min_reg_size;
    int rcap;             // region-list entries kept in LDS (one spare word follows)
    int rect_cap;         // rectangles / segments per frame: sw * sh / min_reg_size (every region owns >= min_reg_size pixels), at most 16384
    int nkeep;            // lines kept after the response sort
    int sort_cap;         // power of two >= rect_cap
    int sort_lds;         // sort keys k_lsd_finalize keeps in LDS (frames with more segments sort in a global scratch row)
    int nfa_pool;         // rectangles of the whole batch the NFA stage buffers hold (entries are compacted over the batch)
    unsigned budget_ticks; // plf_line_params.max_ms in ticks of the 100 MHz wall clock (k_*_budget kernels only)
    uint32_t *sgl;        // [frame][s_stride / 32] bitmap of the static singles (k_lsd_pre; a device buffer of the handle, carried here because every kernel of the line
                          // pipeline gets the geometry): pixels with an angle none of whose neighbours can pass the first alignment test of a region seeded there
};

// banded speculative region growing: one record per effective seed of a band wave, and the buffers of both phases
struct SpecRec { int seed, t0, nt, has_rect; int bx0, by0, bx1, by1; LsdRect rec; };   // b*: bounding box of the accepted pixels, dilated by one
// (the band waves and the validation rounds write and read a record's header as two int4 words -- seed, t0, nt, has_rect | bx0, by0, bx1, by1 -- in front of rec)
static_assert(offsetof(SpecRec, seed) == 0 && offsetof(SpecRec, t0) == 4 && offsetof(SpecRec, nt) == 8 && offsetof(SpecRec, has_rect) == 12, "SpecRec header word 0");
static_assert(offsetof(SpecRec, bx0) == 16 && offsetof(SpecRec, by0) == 20 && offsetof(SpecRec, bx1) == 24 && offsetof(SpecRec, by1) == 28, "SpecRec header word 1");
static_assert(offsetof(SpecRec, rec) >= 32 && sizeof(SpecRec) % 16 == 0 && alignof(SpecRec) <= 16, "SpecRec: the header is 32 bytes, records stay 16-byte aligned in their arrays");
struct SpecBufs {
    uint32_t *rxy;      // [frame][band][s_stride] list overflow of the band waves
    uint32_t *tl;       // [frame][band][tcap] accepted pixels (bit 30: still marked at the end of the seed)
    SpecRec *recs;      // [frame][band][rcap_rec]
    int *cnt;           // [frame][band][4]: records, accepted pixels, overflow, heartbeat / end time stamp
    int spin_bound;     // polls (~3.4 us each) the commit wave waits for a band wave WITHOUT a heartbeat before it gives up (status bit 4)
    uint32_t *seedmap;  // [frame][bm_words]: seeds that own a record
    uint32_t *defmap;   // [frame][bm_words]: pixels with a defined level-line angle (k_lsd_spec_bands)
    uint32_t *tl2;      // [frame][2 * s_stride]: accepted pixels of a seed regrown by the commit kernel
    int *band_y;        // [frame][nbands + 1]: first row of every band (shares of the frame's defined pixels)
    int *done;          // [frame][band]: set (release) when the band wave has written its log; the commit wave waits for it (acquire)
    float stagger;      // band b gets a share proportional to 1 + stagger * b: early bands finish early, the commit wave follows them
    uint32_t *sglob;    // [frame][bm_words]: S of the commit wave when it does not fit the LDS next to T (s_global)
    uint32_t *halo;     // [frame][band][bm_words]: the band's speculative flags after its warm-up rows = its initial state S
    int halo_rows;      // rows above a band that its wave grows first, unrecorded (model: orc_lsd_band_speculation_halo)
    int halo_clip;      // the warm-up regions may not grow below the band's last row + halo_clip (< 0: unbounded)
    int fill_rows;      // > 0: no warm-up growth; the band's first fill_rows rows are presumed taken where they hang on the pixels above through chains of aligned neighbours
    float fill_tol_deg; // largest difference of level-line angles (degrees) between neighbours of such a chain
    int s_global;
    int tcap, rcap_rec, nbands, bm_words;
    // parallel validation rounds (k_lsd_spec_prefix / _validate / _assemble): every band keeps its log consistent with what the bands before it mark
    uint32_t *out;      // [frame][band][bm_words]: pixels marked by the band's own records (current log)
    uint32_t *pre;      // [frame][band][bm_words]: union of out[] of the bands before it
    uint32_t *tl_alt;   // second side of the logs: a validation reads one side and writes the other
    SpecRec *recs_alt;
    int *cnt_alt;
    int *side;          // [frame][band]: side that holds the band's current log (0: tl / recs / cnt)
    int *nrects;        // [frame][band]: rectangles in the band's current log
    int *reach;         // [frame][band][2] = first / last row the band's log can depend on (its own rows and the dilated boxes of its records): a validation round
                        // whose changes lie outside leaves the log as it is.  A buffer of its own (2 ints per slot of the allocation): its base must not depend on the
                        // band count of the CALL, which may exceed the allocation's
    int *round_state;   // [frame][4]: bands whose marks changed in the even / odd rounds, converged, fall back to the serial commit; behind the frames_cap frames: [frame] band workgroups through the current round
    int frames_cap;     // frames round_state was allocated for
    uint32_t *tl2b;     // [frame][band][2 * s_stride]: accepted pixels of a seed regrown by a validation
    int *round_log;     // [frame][band][16 rounds][4] (diagnostics, -DPLF_ROUND_LOG builds with PLF_LSD_ROUND_LOG set; else null): ticks of the band's validation, seeds regrown, pixels regrown, records that stood
    int *band_ticks;    // [frame][band][2]: run time of the band wave in 100 MHz ticks, accepted pixels it logged (diagnostics: how well the band shares are balanced)
};

// NFA table (k_nfa_table): rectangles of fewer than NFA_TAB_N pixels, p = 1/8 * 2^-j for j < NFA_TAB_P
#ifndef NFA_TAB_N
#define NFA_TAB_N 512
#endif
#define NFA_TAB_P 11
