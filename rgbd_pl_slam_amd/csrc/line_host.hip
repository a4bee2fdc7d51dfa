// line_host.hip -- host side of the line extractor: handle, geometry, launch sequence, C ABI.
// Mirrors ORB_SLAM2::LineSegment::ExtractLineSegment (include/ExtractLineSegment.h:38):
//   LSDDetector::detect(img, keylines, scale = int(1.2) = 1, numOctaves = 1)  ->  keep the N best by response
//   -> BinaryDescriptor::compute -> normalised line equations.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <chrono>
#include <hipcub/hipcub.hpp>
#include "plf_common.h"
#include "lsd_geom.h"

__global__ void k_lsd_pre(const uint8_t *, ptrdiff_t, ptrdiff_t, float *, double *, double2 *, float2 *, LsdGeom, LsdTaps, const int *, const float2 *,
                          const int *, const float2 *, int *);
__global__ void k_lsd_balance(const int *, int *, int, int);
__global__ void k_lsd_regions2(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, const uint32_t *, int, const int *);
__global__ void k_lsd_regions_lat(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, const uint32_t *, int *);
// the same kernels with the time budget of plf_line_params.max_ms compiled in (separate instances: the default ones read no clock)
__global__ void k_lsd_regions2_budget(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, const uint32_t *, int, const int *);
__global__ void k_lsd_regions_lat_budget(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, const uint32_t *, int *);
__global__ void k_lsd_spec_fused_budget(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, SpecBufs, int *, int);
__global__ void k_lsd_spec_grow_budget(float *, const double *, const double2 *, const float2 *, LsdGeom, SpecBufs);
__global__ void k_lsd_spec_commit_budget(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, SpecBufs, int *);
__global__ void k_lsd_maxgrad(const float *, const double *, double *, LsdGeom);
__global__ void k_lsd_seedkeys(const float *, const double *, const double *, uint32_t *, LsdGeom);
__global__ void k_lsd_lgamma_table(double *);
__global__ void k_lsd_count_used(const float *, int *, LsdGeom);
struct NfaEntry { LsdRect r; int frame, nprec, pad0, pad1; };
struct NfaCounts { int total, alg[6], pad; };
struct NfaState { LsdRect rec; double log_nfa; int frame, rect; };
__global__ void k_nfa_init(const LsdRect *, const int *, uint8_t *, NfaEntry *, NfaState *, int *, int *, LsdGeom);
__global__ void k_nfa_clamp(int *, int *, LsdGeom);
__global__ void k_nfa_count(const float *, const NfaEntry *, const int *, int, int, NfaCounts *, LsdGeom);
__global__ void k_nfa_count1(const float *, const NfaEntry *, const int *, int, int, NfaCounts *, LsdGeom);
__global__ void k_nfa_count_w(const float *, const NfaEntry *, const int *, int, int, NfaCounts *, LsdGeom);
__global__ void k_nfa_count1_w(const float *, const NfaEntry *, const int *, int, int, NfaCounts *, LsdGeom);
__global__ void k_nfa_eval(int, const double *, const double *, const NfaCounts *, const NfaEntry *, const int *, double *, LsdGeom);
__global__ void k_nfa_table(double *, const double *, double);
__global__ void k_nfa_small(const float *, const double *, const LsdRect *, const int *, uint8_t *, float4 *, NfaEntry *, NfaState *, int *, int *, LsdGeom, int, NfaState *, int *, int);
__global__ void k_nfa_small2(const float *, const double *, const LsdRect *, uint8_t *, float4 *, NfaEntry *, NfaState *, int *, int *, LsdGeom, int, const NfaState *, const int *, int);
__global__ void k_nfa_math(int, const double *, const NfaEntry *, const NfaState *, NfaState *, NfaEntry *, int *, float4 *, uint8_t *, LsdGeom);
__global__ void k_nfa_fused(const float *, const double *, const double *, const LsdRect *, const int *, uint8_t *, float4 *, LsdGeom);
__global__ void k_lsd_finalize(const float4 *, const uint8_t *, const int *, float4 *, int *, plf_keyline *, plf_keyline *, double *, int *,
                               int, int *, unsigned long long *, LsdGeom);
__global__ void k_sobel3(const uint8_t *, ptrdiff_t, ptrdiff_t, short2 *, LsdGeom);
__global__ void k_blur5_sobel3(const uint8_t *, ptrdiff_t, ptrdiff_t, short2 *, LsdGeom, int4);
__global__ void k_lbd(const short2 *, const plf_keyline *, const int *, uint8_t *, int, LsdGeom, const LbdCoefs *);

__global__ void k_lsd_spec_fused(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, SpecBufs, int *, int);
__global__ void k_lsd_spec_rows(const float *, LsdGeom, SpecBufs, int *);
__global__ void k_lsd_spec_bands(LsdGeom, SpecBufs, const int *, int *);
__global__ void k_lsd_spec_grow(float *, const double *, const double2 *, const float2 *, LsdGeom, SpecBufs);
__global__ void k_lsd_spec_commit(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, SpecBufs, int *);
__global__ void k_lsd_spec_commit_rest(float *, const double *, const double2 *, const float2 *, uint32_t *, LsdRect *, int *, int *, LsdGeom, SpecBufs, int *, int);
__global__ void k_lsd_spec_prefix(SpecBufs, int);
__global__ void k_lsd_spec_clear(SpecBufs, int *, int);
__global__ void k_lsd_spec_validate(float *, const double *, const double2 *, const float2 *, LsdGeom, SpecBufs, int);
__global__ void k_lsd_spec_assemble(LsdRect *, int *, int *, LsdGeom, SpecBufs, int);

// Schedule knobs of the line extractor.  Read ONCE, when the handle is created (environment: PLF_LSD_* / PLF_NFA_FUSED, for experiments and the test hooks that force a
// schedule), changed afterwards only through plf_line_tune(); a call never looks at the environment (round 3 did, ~30 getenv per call).  PLF_TUNE_AUTO = chosen per
// call from the number of frames in flight (the table in line_enqueue).
#define PLF_TUNE_AUTO (-0x7fffffff)
struct LineTune {
    int lat_max;          // PLF_LSD_LAT_MAX      frames in flight up to which the latency kernel (one frame per workgroup + L2 warm-up waves) is used when speculation is off
    int spec_bands;       // PLF_LSD_SPEC_BANDS   row bands of the speculative schedule (AUTO: 48 / 32 / 16 / 8 / 4 / 2 by frames in flight)
    int spec_max;         // PLF_LSD_SPEC_MAX     frames in flight up to which the speculative schedule is used (256; 0 = never)
    int spec_z;           // PLF_LSD_SPEC_Z       ... up to which the validation rounds replace the serial commit wave (256: round 6, mid-range batches)
    int spec_rounds;      // PLF_LSD_SPEC_ROUNDS  validation rounds enqueued (20; 12 until the real photographs of round 6: gravel, a coffee cup and one natural-image-like frame
                          //                      needed 13-14 and fell back to the serial commit of the rest, +1.5 ms; an idle round costs ~4 us)
    int spec_halo;        // PLF_LSD_SPEC_HALO    warm-up rows above a band (AUTO: 4 with validation rounds, 16 otherwise)
    int spec_fill;        // PLF_LSD_SPEC_FILL    rows of the no-growth guess instead of warm-up growth (0 = off)
    float spec_fill_tol;  // PLF_LSD_SPEC_FILL_TOL
    int spec_clip;        // PLF_LSD_SPEC_CLIP    rows below a band its warm-up regions may reach (AUTO: 16 from 4 frames in flight on, unbounded below; < 0 = unbounded)
    float spec_stagger;   // PLF_LSD_SPEC_STAGGER band shares of the one-launch schedule (0.2)
    int spec_nofuse;      // PLF_LSD_SPEC_NOFUSE  never the one-launch schedule
    int spec_spins;       // PLF_LSD_SPEC_SPINS   polls without a heartbeat before the commit wave gives up (test hook)
    int spec_reccap;      // PLF_LSD_SPEC_RECCAP  records per band log (test hook: forces the overflow path)
    int wpg;              // PLF_LSD_WPG          frames (= waves) per workgroup of the large-batch region kernel (AUTO: 2 up to 768 frames in flight, 4 up to 1536, else 8)
    int nfa_fused;        // PLF_NFA_FUSED        frames in flight up to which one wave per rectangle runs all NFA stages (64)
    int nfa_small;        // PLF_NFA_SMALL        2: rect_improve of the rectangles the table covers in one launch (k_nfa_small), 16 lanes per rectangle; 1: only for more
                          //                      than nfa_fused frames in flight (one frame: 4.41 ms with it, 4.63 ms with k_nfa_fused); 0: off
    int nfa_two_pass;     // PLF_NFA_TWO_PASS     1: above nfa_fused frames in flight k_nfa_small only runs stage 0 and queues the undecided rectangles for k_nfa_small2 (stages 1-4)
    int nfa_table;        // PLF_NFA_TABLE        1: NFA values of rectangles of fewer than 512 pixels come from the per-image-size table (k_nfa_table)
    int balance;          // PLF_LSD_BALANCE      1: large batches -- the frames are dealt to the waves of k_lsd_regions2 by chain length (k_lsd_balance); 0: in batch order
    float slow_factor;    // PLF_LSD_SLOW_FACTOR  PLF_W_SLOW: a host-output call that takes more than this many times the median per-frame time of the recent calls (10; 0: off)
    float slow_floor_ms;  // PLF_LSD_SLOW_FLOOR_MS  ... and more than this per frame (20 ms)
};
static int tune_env_i(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static float tune_env_f(const char *name, float dflt) { const char *e = getenv(name); return e ? (float)atof(e) : dflt; }
static void line_tune_init(LineTune *t)
{
    t->lat_max = tune_env_i("PLF_LSD_LAT_MAX", 8);
    t->spec_bands = tune_env_i("PLF_LSD_SPEC_BANDS", PLF_TUNE_AUTO);
    t->spec_max = tune_env_i("PLF_LSD_SPEC_MAX", 256);
    t->spec_z = tune_env_i("PLF_LSD_SPEC_Z", 256);
    t->spec_rounds = std::max(1, std::min(64, tune_env_i("PLF_LSD_SPEC_ROUNDS", 20)));
    t->spec_halo = tune_env_i("PLF_LSD_SPEC_HALO", PLF_TUNE_AUTO);
    t->spec_fill = tune_env_i("PLF_LSD_SPEC_FILL", 0);
    t->spec_fill_tol = tune_env_f("PLF_LSD_SPEC_FILL_TOL", 11.25f);
    t->spec_clip = tune_env_i("PLF_LSD_SPEC_CLIP", PLF_TUNE_AUTO);
    t->spec_stagger = tune_env_f("PLF_LSD_SPEC_STAGGER", 0.2f);
    t->spec_nofuse = getenv("PLF_LSD_SPEC_NOFUSE") ? 1 : 0;
    t->spec_spins = std::max(64, tune_env_i("PLF_LSD_SPEC_SPINS", 1 << 21));
    t->slow_factor = tune_env_f("PLF_LSD_SLOW_FACTOR", 10.f);
    t->slow_floor_ms = tune_env_f("PLF_LSD_SLOW_FLOOR_MS", 20.f);
    { const int v = tune_env_i("PLF_LSD_SPEC_RECCAP", 8192); t->spec_reccap = (v >= 1 && v <= 8192) ? v : 8192; }
    t->wpg = getenv("PLF_LSD_WPG") ? std::max(1, std::min(16, tune_env_i("PLF_LSD_WPG", 8))) : PLF_TUNE_AUTO;
    t->nfa_fused = tune_env_i("PLF_NFA_FUSED", 64);
    t->nfa_table = tune_env_i("PLF_NFA_TABLE", 1);
    t->nfa_small = tune_env_i("PLF_NFA_SMALL", 2);
    t->nfa_two_pass = tune_env_i("PLF_NFA_TWO_PASS", 1);
    t->balance = tune_env_i("PLF_LSD_BALANCE", 1);
}

struct plf_line {
    plf_line_params prm;
    LineTune tune;
    SpecBufs spec;            // banded speculative region growing (few frames in flight); allocated on first use
    int spec_frames;          // frames the buffers were sized for (0: not allocated, -1: allocation failed / disabled)
    size_t spec_slots;        // (frame, band) slots they hold = frames x bands at allocation time
    bool spec_sglob_per_band; // sglob was allocated with one bitmap per band (validation rounds)
    size_t fused_lds, fused_capacity;   // k_lsd_spec_fused: workgroups of that LDS size the GPU can hold at once (occupancy query)
    int n_cus;                // compute units of the device (queried on first use)
    int *d_spec_stats;
    int *d_spec_rowcnt;       // [frames][1024] defined pixels per row unit, then [frames][64][65] scratch of the band sweep
    int device;
    LsdGeom g;
    LsdTaps taps;
    LbdCoefs lbd;
    LbdCoefs *d_lbd;   // the same, in device memory (k_lbd indexes it per lane)
    int4 blur5;               // 8-bit fixed-point taps of GaussianBlur(5 x 5, sigma 1): k[0], k[1], k[2]
    int cur_w, cur_h;
    size_t alloc_full, alloc_scaled;  // elements per frame the buffers were sized for
    int alloc_rect_cap;
    int alloc_nfa_pool;
    size_t regions_lds, finalize_lds, nfa_lds;
    hipStream_t stream;
    uint8_t *d_in, *d_keep, *d_ldesc;
    double *d_modgrad, *d_lineeq, *d_lgam;
    double *d_nfa_tab;        // nfa(n, k, p) of small rectangles (k_nfa_table), valid for scaled images with LOG_NT == nfa_tab_log_nt
    double nfa_tab_log_nt;
    // seed_order = 1 only: per-frame max gradient, (bin, pixel) keys before / after the segmented sort, segment offsets, sort scratch
    double *d_maxgrad;
    uint32_t *d_keys[2];
    int *d_seg_off;
    void *d_sort_tmp;
    size_t sort_tmp_bytes;
    NfaEntry *d_ent[2];
    NfaState *d_st[2];
    NfaCounts *d_cnt;
    int *d_nfa_counters;
    int *d_nfa_fcnt;          // [frame]: rectangles queued for k_nfa_small2
    double *d_vals;
    unsigned long long *d_sort_scratch;   // [frame][sort_cap] when sort_cap > sort_lds
    double2 *d_cs;
    float2 *d_cs0;
    uint32_t *d_sgl;          // [max_batch][s_stride / 32] bitmap of the static singles, written by k_lsd_pre (LsdGeom::sgl)
    float *d_ang;
    uint32_t *d_rxy;
    LsdRect *d_rects;
    float4 *d_seg, *d_segs_out;
    short2 *d_grad;
    plf_keyline *d_kl_tmp, *d_lines;
    int *d_counters;  // nrect[B], nseg[B], nout[B], status
    int *d_xofs, *d_yofs;
    float2 *d_xa, *d_yb;
    int *d_balance;           // [2][max_batch + 16]: defined pixels per frame (k_lsd_pre), then the frame of every wave slot of k_lsd_regions2 (k_lsd_balance)
    int last_frames;
    hipStream_t last_stream;   // stream of the most recent call
    bool last_stream_set;
    PlfStreamOrder order;      // event recorded at the end of every call: a call on another stream waits for it on the device
    int prof_on, prof_n;
    hipEvent_t prof_ev[2 * 512];  // (start, stop) pairs of the region kernel
    hipEvent_t ev_front;          // recorded after the front stages of the last batch (plf_line_wait_front)
    bool ev_front_set;
    uint8_t *h_pin;        // pinned staging of the host-output path for a few frames in flight (results of <= PIN_FRAMES frames come back in one go)
    size_t h_pin_bytes;
    double prof_ms;
    int prof_launches;
    // a batch that was redone in halves (status bit 1): the per-frame "ran out of max_ms" flags and the status bits of ALL the pieces, accumulated on the host
    // (the device words only ever hold the last piece's)
    int32_t *retry_flags;      // [max_batch]
    int retry_status, retry_depth;
    bool retry_valid;
    // PLF_W_SLOW: per-frame wall time (ms) of the last host-output calls at the current image size (ring), and whether the last call was flagged
    float slow_hist[32];
    int slow_n, slow_w, slow_h;
    bool slow_last;
};

static void line_free(plf_line *h)
{
    void *ptrs[] = {h->d_in, h->d_keep, h->d_ldesc, h->d_modgrad, h->d_maxgrad, h->d_keys[0], h->d_keys[1], h->d_seg_off, h->d_sort_tmp, h->d_lineeq, h->d_cs,
                    h->d_ang, h->d_rxy, h->d_cs0, h->d_sgl, h->d_rects, h->d_seg, h->d_segs_out, h->d_grad, h->d_kl_tmp, h->d_lines, h->d_counters,
                    h->d_xofs, h->d_yofs, h->d_xa, h->d_yb, h->d_balance, h->d_lgam, h->d_nfa_tab, h->d_ent[0], h->d_ent[1], h->d_st[0], h->d_st[1], h->d_cnt, h->d_nfa_counters, h->d_nfa_fcnt, h->d_vals, h->d_sort_scratch, h->d_lbd};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    void *sp[] = {h->spec.rxy, h->spec.tl, h->spec.recs, h->spec.cnt, h->spec.seedmap, h->spec.defmap, h->spec.tl2, h->spec.band_y, h->spec.done, h->spec.sglob, h->spec.halo, h->d_spec_stats,
                  h->spec.out, h->spec.pre, h->spec.tl_alt, h->spec.recs_alt, h->spec.cnt_alt, h->spec.side, h->spec.nrects, h->spec.reach, h->spec.round_state, h->spec.tl2b, h->spec.band_ticks, h->spec.round_log, h->d_spec_rowcnt};
    for (void *p : sp) if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (int i = 0; i < 2 * 512; i++) if (h->prof_ev[i]) (void)hipEventDestroy(h->prof_ev[i]);
    if (h->ev_front) (void)hipEventDestroy(h->ev_front);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    free(h->retry_flags); h->retry_flags = nullptr;
    plf_order_free(h->order);
}

static int line_geometry(const plf_line *h, int w, int hh, LsdGeom *g)
{
    memset(g, 0, sizeof(*g));
    const double SCALE = 0.8, ANG_TH = 22.5, QUANT = 2.0;
    g->w = w; g->h = hh;
    g->sw = (int)lrint(w * SCALE); g->sh = (int)lrint(hh * SCALE);  // cv::resize: saturate_cast<int>(ssize * inv_scale)
    if (g->sw < 8 || g->sh < 8 || w > 32000 || hh > 32000) return PLF_E_BADARG;
    g->full_stride = (uint32_t)plf_align_up((size_t)w * hh, 64);
    g->s_stride = (uint32_t)plf_align_up((size_t)g->sw * g->sh, 64);
    g->prec = 3.1415926535897932384626433832795 * ANG_TH / 180;
    g->p = ANG_TH / 180;
    g->rho = QUANT / sin(g->prec);
    {   // ll_angle's test `norm <= rho` on the squared norm: IEEE sqrt is correctly rounded and monotone, so the pixels with sqrt(q) <= rho are exactly q <= rho_q
        double t = g->rho * g->rho;
        while (sqrt(t) > g->rho) t = nextafter(t, -INFINITY);
        while (sqrt(nextafter(t, INFINITY)) <= g->rho) t = nextafter(t, INFINITY);
        g->rho_q = t;
    }
    g->log_nt = 5 * (log10((double)g->sw) + log10((double)g->sh)) / 2 + log10(11.0);
    g->min_reg_size = (int)(-g->log_nt / log10(g->p));
    // LDS of k_lsd_regions = the first rcap entries of the region list (+1 mailbox word); longer regions spill to HBM.
    // The USED flags live in the angle map, so a workgroup needs ~6 KB and a CU hosts as many frames as it has wave
    // slots: the kernel is a latency-bound serial chain per frame and its throughput is the number of frames in flight.
    g->rcap = 1279;   // (1535 until the large-batch kernel parked its seed chunk in LDS: no measurable difference, 68.8 vs 68.8 ms per 4096 frames)
    if (const char *e = getenv("PLF_LSD_RCAP")) { if (atoi(e) >= 127 && atoi(e) <= 16384) g->rcap = atoi(e); }   // (>= 96: chunk_taken parks its flag words in list entries 32..95)
    // Rectangles per frame: regions are disjoint and one that yields a rectangle owns >= min_reg_size pixels, so sw * sh / min_reg_size bounds
    // their number for ANY image (13107 at VGA, 46k at 1280x960; real frames produce 500-1500): no frame can overflow its rows.  Only the NFA
    // stage buffers are pooled over the batch (line_nfa_pool).
    int rc = g->sw * g->sh / (g->min_reg_size > 0 ? g->min_reg_size : 1) + 1;
    rc = (rc + 255) & ~255;
    if (rc < 256) rc = 256;
    g->rect_cap = rc;
    g->sort_cap = 256;
    while (g->sort_cap < rc) g->sort_cap <<= 1;
    g->sort_lds = g->sort_cap < 4096 ? g->sort_cap : 4096;
    g->nkeep = h->prm.nlines;
    // time budget of the region stage (opt-in; the reference has none): milliseconds -> ticks of the 100 MHz wall clock the kernels read
    g->budget_ticks = h->prm.max_ms > 0.f ? (unsigned)std::min((double)h->prm.max_ms * 1.0e5, 4.0e9) : 0u;
    if (h->prm.max_ms > 0.f && g->budget_ticks == 0u) g->budget_ticks = 1u;
    return PLF_OK;
}

// NFA stage capacity in rectangles for a batch of B frames: the stage works on one list compacted over the batch, sized for 2048-8192 rectangles
// per frame on average (by image area) and for at least one worst-case frame
static size_t line_nfa_pool(const LsdGeom &g, size_t B)
{
    size_t per = 2048;
    while (per < (size_t)g.sw * g.sh / 48 && per < 8192) per <<= 1;
    size_t pool = B * per;
    if (pool < (size_t)g.rect_cap) pool = (size_t)g.rect_cap;
    return pool < 0x7fffffff / 8 ? pool : 0x7fffffff / 8;
}

static int line_configure(plf_line *h, int w, int hh)
{
    if (h->cur_w == w && h->cur_h == hh) return PLF_OK;
    LsdGeom g;
    int rc = line_geometry(h, w, hh, &g);
    if (rc != PLF_OK) return rc;
    if (g.full_stride > h->alloc_full || g.s_stride > h->alloc_scaled) return PLF_E_BADARG;
    // cv::resize(double image, fx = fy = 0.8, INTER_LINEAR): float coefficients, see oracle/lsd_oracle.c
    const double scale = 1. / 0.8;
    std::vector<int> xofs(g.sw), yofs(g.sh);
    std::vector<float2> xa(g.sw), yb(g.sh);
    int xmax = g.sw;
    for (int dx = 0; dx < g.sw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= w) { if (dx < xmax) xmax = dx; if (sx >= w - 1) { fx = 0; sx = w - 1; } }
        xofs[dx] = sx; xa[dx].x = 1.f - fx; xa[dx].y = fx;
    }
    for (int dy = 0; dy < g.sh; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy; yb[dy].x = 1.f - fy; yb[dy].y = fy;
    }
    g.xmax = xmax;
    // k_lsd_pre keeps the blurred source pixels of a 65 x 17 scaled tile in LDS: PRE_SC x PRE_SR
    for (int d0 = 0; d0 < g.sw; d0 += PRE_TW)
        if (std::min(xofs[std::min(d0 + PRE_TW, g.sw - 1)] + 1, w - 1) - xofs[d0] + 1 > PRE_SC) return PLF_E_BADARG;
    for (int d0 = 0; d0 < g.sh; d0 += PRE_TH) {
        const int lo = std::min(std::max(yofs[d0], 0), hh - 1), hi = std::min(std::max(yofs[std::min(d0 + PRE_TH, g.sh - 1)] + 1, 0), hh - 1);
        if (hi - lo + 1 > PRE_SR) return PLF_E_BADARG;
    }
    // keep the allocation strides so that per-frame offsets stay inside the buffers
    g.full_stride = (uint32_t)h->alloc_full; g.s_stride = (uint32_t)h->alloc_scaled; g.rect_cap = h->alloc_rect_cap;
    g.sort_cap = 256;
    while (g.sort_cap < g.rect_cap) g.sort_cap <<= 1;
    g.sort_lds = g.sort_cap < 4096 ? g.sort_cap : 4096;
    g.nfa_pool = h->alloc_nfa_pool;
    g.sgl = h->d_sgl;
    // kernels of a previous call (any stream: the caller's streams do not synchronise with the null stream) may still read the tables
    if (h->cur_w >= 0) PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy(h->d_xofs, xofs.data(), sizeof(int) * g.sw, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_xa, xa.data(), sizeof(float2) * g.sw, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_yofs, yofs.data(), sizeof(int) * g.sh, hipMemcpyHostToDevice));
    PLF_HIP_TRY(hipMemcpy(h->d_yb, yb.data(), sizeof(float2) * g.sh, hipMemcpyHostToDevice));
    h->g = g;
    h->regions_lds = (size_t)(g.rcap + 1) * 4 + 8 + 768 + 64;   // list + mailbox word, region2rect's staging (RegCtx::stg)
    h->finalize_lds = (size_t)g.sort_lds * 8 + 260 * 4;   // (the compaction flags alias the sort keys)
    h->nfa_lds = (size_t)g.sh * 2 * sizeof(int) + 64;
    h->cur_w = w; h->cur_h = hh;
    return PLF_OK;
}

extern "C" int plf_line_create(const plf_line_params *p, plf_line **out)
{
    if (!p || !out) return PLF_E_BADARG;
    *out = nullptr;
    if (p->nlines < 1 || p->max_batch < 1 || p->max_width < 16 || p->max_height < 16) return PLF_E_BADARG;
    if (p->seed_order != 0 && p->seed_order != 1) return PLF_E_BADARG;
    if (p->lbd_sobel_input != PLF_LBD_BLURRED && p->lbd_sobel_input != PLF_LBD_RAW) return PLF_E_BADARG;
    if (!(p->max_ms >= 0.f)) return PLF_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[plf] no HIP device available: the line extractor has no CPU path\n");
        return PLF_E_HIP;
    }
    if (p->device < 0 || p->device >= ndev) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(p->device));
    plf_line *h = (plf_line *)calloc(1, sizeof(plf_line));
    if (!h) return PLF_E_NOMEM;
    h->prm = *p; h->device = p->device;
    line_tune_init(&h->tune);
    h->retry_flags = (int32_t *)calloc((size_t)p->max_batch, sizeof(int32_t));
    if (!h->retry_flags) { free(h); return PLF_E_NOMEM; }
    LsdGeom g;
    int rc = line_geometry(h, p->max_width, p->max_height, &g);
    if (rc != PLF_OK) { free(h->retry_flags); free(h); return rc; }
    h->alloc_full = g.full_stride; h->alloc_scaled = g.s_stride; h->alloc_rect_cap = g.rect_cap;
    h->alloc_nfa_pool = (int)line_nfa_pool(g, (size_t)p->max_batch);
    // cv::getGaussianKernel(7, 0.6/0.8, CV_64F):  h = ceil(sigma * sqrt(2 * 3 * ln 10)) = 3 -> ksize 7
    {
        const double sigma = 0.6 / 0.8, scale2X = -0.5 / (sigma * sigma);
        double sum = 0;
        for (int i = 0; i < 7; i++) { const double x = i - 3.0; h->taps.k[i] = exp(scale2X * x * x); sum += h->taps.k[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) h->taps.k[i] *= sum;
    }
    // cv::getGaussianKernel(5, 1, CV_32F) converted to 8-bit fixed point (createSeparableLinearFilter, 8U smoothing kernels): 14 63 103 63 14
    {
        float cf[5];
        double sum = 0;
        for (int i = 0; i < 5; i++) { const double x = i - 2.0; cf[i] = (float)exp(-0.5 * x * x); sum += cf[i]; }
        sum = 1. / sum;
        int k[5];
        for (int i = 0; i < 5; i++) { cf[i] = (float)(cf[i] * sum); k[i] = (int)lrint((double)cf[i] * 256.0); }
        h->blur5 = make_int4(k[0], k[1], k[2], 0);
    }
    // BinaryDescriptor ctor: gaussCoefL_ (21 taps, centre 10, sigma 7), gaussCoefG_ (63 taps, centre 31, sigma 31)
    {
        double u = (7 * 3 - 1) / 2, sigma = (7 * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < 21; i++) { const double d = i - u; h->lbd.gL[i] = (float)exp(d * d * inv); }
        u = (9 * 7 - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < 63; i++) { const double d = i - u; h->lbd.gG[i] = (float)exp(d * d * inv); }
    }
    const size_t B = (size_t)p->max_batch, F = g.full_stride, S = g.s_stride, R = (size_t)g.rect_cap, NP = (size_t)h->alloc_nfa_pool;
    const int cap = p->nlines;
#define ALLOC(ptr, bytes)                                                             \
    do {                                                                              \
        if (hipMalloc((void **)&(ptr), (bytes) > 0 ? (bytes) : 256) != hipSuccess) { \
            line_free(h); free(h); return PLF_E_NOMEM;                                \
        }                                                                             \
    } while (0)
    ALLOC(h->d_in, B * (size_t)p->max_width * p->max_height);
    ALLOC(h->d_grad, B * F * sizeof(short2));
    ALLOC(h->d_modgrad, B * S * sizeof(double));
    ALLOC(h->d_cs, B * S * sizeof(double2));
    ALLOC(h->d_ang, B * S * sizeof(float));
    ALLOC(h->d_cs0, B * S * sizeof(float2));
    ALLOC(h->d_sgl, B * S / 8);
    ALLOC(h->d_rxy, B * S * sizeof(uint32_t));
    ALLOC(h->d_rects, B * R * sizeof(LsdRect));
    ALLOC(h->d_seg, B * R * sizeof(float4));
    ALLOC(h->d_segs_out, B * R * sizeof(float4));
    ALLOC(h->d_keep, B * R);
    ALLOC(h->d_kl_tmp, B * R * sizeof(plf_keyline));
    ALLOC(h->d_lines, B * (size_t)cap * sizeof(plf_keyline));
    ALLOC(h->d_ldesc, B * (size_t)cap * 32);
    ALLOC(h->d_lineeq, B * (size_t)cap * 3 * sizeof(double));
    ALLOC(h->d_balance, 2 * (B + 16) * sizeof(int));
    ALLOC(h->d_counters, (5 * B + 16) * sizeof(int));   // nrect[B], nseg[B], nout[B], status[16] + truncated[B], chain lengths[B]
    ALLOC(h->d_lgam, 65536 * sizeof(double));
    ALLOC(h->d_nfa_tab, (size_t)NFA_TAB_P * (NFA_TAB_N * (NFA_TAB_N + 1) / 2) * sizeof(double));   // (lsd_geom.h)
    h->nfa_tab_log_nt = -1.0;
    ALLOC(h->d_lbd, sizeof(LbdCoefs));
    ALLOC(h->d_ent[0], NP * 5 * sizeof(NfaEntry)); ALLOC(h->d_ent[1], NP * 5 * sizeof(NfaEntry));
    ALLOC(h->d_st[0], NP * sizeof(NfaState)); ALLOC(h->d_st[1], NP * sizeof(NfaState));
    ALLOC(h->d_cnt, NP * 5 * sizeof(NfaCounts));
    ALLOC(h->d_nfa_counters, 16 * sizeof(int));
    ALLOC(h->d_nfa_fcnt, B * sizeof(int));
    ALLOC(h->d_vals, NP * 6 * sizeof(double));
    if (g.sort_cap > g.sort_lds) ALLOC(h->d_sort_scratch, B * (size_t)g.sort_cap * sizeof(unsigned long long));
    if (p->seed_order == 1) {
        if ((size_t)g.sw * g.sh >= (1u << 20)) { line_free(h); free(h); return PLF_E_BADARG; }   // pixel index must fit the 20 low key bits
        ALLOC(h->d_maxgrad, B * sizeof(double));
        ALLOC(h->d_keys[0], B * S * sizeof(uint32_t)); ALLOC(h->d_keys[1], B * S * sizeof(uint32_t));
        ALLOC(h->d_seg_off, (2 * B) * sizeof(int));
        h->sort_tmp_bytes = 0;
        if (hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, h->sort_tmp_bytes, h->d_keys[0], h->d_keys[1], (int)(B * S), (int)B, h->d_seg_off,
                                                       h->d_seg_off + B, 0, 30, (hipStream_t)0) != hipSuccess) { line_free(h); free(h); return PLF_E_HIP; }
        ALLOC(h->d_sort_tmp, h->sort_tmp_bytes + 256);
    }
    ALLOC(h->d_xofs, sizeof(int) * (size_t)g.sw); ALLOC(h->d_xa, sizeof(float2) * (size_t)g.sw);
    ALLOC(h->d_yofs, sizeof(int) * (size_t)g.sh); ALLOC(h->d_yb, sizeof(float2) * (size_t)g.sh);
#undef ALLOC
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { line_free(h); free(h); return PLF_E_HIP; }
    (void)hipFuncSetAttribute((const void *)k_lsd_regions_lat, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_grow, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_commit, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_validate, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_commit_rest, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_finalize, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_regions_lat_budget, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_grow_budget, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_commit_budget, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void *)k_lsd_spec_fused_budget, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_lsd_lgamma_table, dim3(65536 / 256), dim3(256), 0, h->stream, h->d_lgam);
    if (hipMemcpy(h->d_lbd, &h->lbd, sizeof(LbdCoefs), hipMemcpyHostToDevice) != hipSuccess) { line_free(h); free(h); return PLF_E_HIP; }
    h->cur_w = -1; h->cur_h = -1;
    rc = line_configure(h, p->max_width, p->max_height);
    if (rc != PLF_OK) { line_free(h); free(h); return rc; }
    if (hipDeviceSynchronize() != hipSuccess) { line_free(h); free(h); return PLF_E_HIP; }
    *out = h;
    return PLF_OK;
}

// Tuning / test hook: change one schedule knob of a handle (the names of struct LineTune, e.g. "nfa_fused", "spec_max"; value PLF_TUNE_AUTO = -2147483647 restores the
// per-call choice where there is one).  Takes effect from the next call on.
extern "C" int plf_line_tune(plf_line *h, const char *name, double value)
{
    if (!h || !name) return PLF_E_BADARG;
    LineTune &t = h->tune;
    const int v = (int)value;
    struct { const char *n; int *p; } ints[] = {{"lat_max", &t.lat_max}, {"spec_bands", &t.spec_bands}, {"spec_max", &t.spec_max}, {"spec_z", &t.spec_z},
        {"spec_rounds", &t.spec_rounds}, {"spec_halo", &t.spec_halo}, {"spec_fill", &t.spec_fill}, {"spec_clip", &t.spec_clip}, {"spec_nofuse", &t.spec_nofuse},
        {"spec_spins", &t.spec_spins}, {"spec_reccap", &t.spec_reccap}, {"wpg", &t.wpg}, {"nfa_fused", &t.nfa_fused}, {"nfa_table", &t.nfa_table}, {"nfa_small", &t.nfa_small}, {"nfa_two_pass", &t.nfa_two_pass}, {"balance", &t.balance}};
    for (auto &e : ints)
        if (!strcmp(name, e.n)) {
            if (!strcmp(name, "spec_rounds") && (v < 1 || v > 64)) return PLF_E_BADARG;
            if (!strcmp(name, "spec_reccap") && (v < 1 || v > 8192)) return PLF_E_BADARG;
            if (!strcmp(name, "spec_spins") && v < 64) return PLF_E_BADARG;
            if (!strcmp(name, "wpg") && v != PLF_TUNE_AUTO && (v < 1 || v > 16)) return PLF_E_BADARG;
            if (!strcmp(name, "spec_bands") && v != PLF_TUNE_AUTO && (v < 0 || v > 64)) return PLF_E_BADARG;   // (0 / 1: speculation off; the call clamps to what the frame allows)
            if ((!strcmp(name, "nfa_small") && (v < 0 || v > 2)) || ((!strcmp(name, "nfa_table") || !strcmp(name, "nfa_two_pass") || !strcmp(name, "balance")) && (v < 0 || v > 1)))
                return PLF_E_BADARG;
            *e.p = v;
            return PLF_OK;
        }
    if (!strcmp(name, "spec_fill_tol")) { t.spec_fill_tol = (float)value; return PLF_OK; }
    if (!strcmp(name, "spec_stagger")) { t.spec_stagger = (float)value; return PLF_OK; }
    if (!strcmp(name, "slow_factor")) { if (value < 0) return PLF_E_BADARG; t.slow_factor = (float)value; return PLF_OK; }
    if (!strcmp(name, "slow_floor_ms")) { if (value < 0) return PLF_E_BADARG; t.slow_floor_ms = (float)value; return PLF_OK; }
    return PLF_E_BADARG;
}

extern "C" void plf_line_destroy(plf_line *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    line_free(h);
    free(h);
}

// Frames (= waves) per workgroup of k_lsd_regions2.  With 8 per workgroup a batch of 512 frames is 64 workgroups: a quarter of the CUs run eight chains each while
// the rest idle.  Round 6 (tools/midrange_sweep.sh, the step of bench.py --batch N): 512 frames in flight 11.0 k -> 12.9 k frames/s with 2 per workgroup, 1024:
// 21.6 k -> 22.4 k with 4; from 2048 on the launch fills the chip either way and 8 keeps the LDS of a CU for the co-running tiles.
static int line_wpg(const LineTune &T, int B) { return T.wpg != PLF_TUNE_AUTO ? T.wpg : (B <= 768 ? 2 : B <= 1536 ? 4 : 8); }

static int line_enqueue(plf_line *h, const uint8_t *d_gray, int B, ptrdiff_t pitch, ptrdiff_t fstride, plf_keyline *d_lines, uint8_t *d_ldesc,
                        double *d_eq, int *d_nout, int capacity, hipStream_t s)
{
    const LsdGeom &g = h->g;
    const size_t MB = (size_t)h->prm.max_batch;
    int *nrect = h->d_counters, *nseg = h->d_counters + MB, *status = h->d_counters + 3 * MB, *nfail_unused = h->d_counters + 3 * MB + 16;
    (void)nfail_unused;
    PLF_HIP_TRY(hipMemsetAsync(status, 0, (16 + MB) * sizeof(int), s));
    // large batches (the one-wave-per-frame kernel k_lsd_regions2): the frames are dealt to its waves by chain length = defined pixels, counted by k_lsd_pre
    const bool balance = h->tune.balance && B > h->tune.spec_max && B > h->tune.lat_max && B <= 65536 && h->prm.seed_order == 0;
    int *d_cost = balance ? h->d_balance : nullptr, *d_perm = balance ? h->d_balance + h->prm.max_batch + 16 : nullptr;
    if (balance) PLF_HIP_TRY(hipMemsetAsync(d_cost, 0, (size_t)B * sizeof(int), s));
    // (bitmap of the static singles: k_lsd_pre writes whole words when the scaled rows are multiples of 32 pixels, and ORs into a cleared map otherwise)
    if (g.sw & 31) PLF_HIP_TRY(hipMemsetAsync(h->d_sgl, 0, (size_t)B * (g.s_stride / 8), s));
    hipLaunchKernelGGL(k_lsd_pre, dim3((g.sw + PRE_TW - 1) / PRE_TW, (g.sh + PRE_TH - 1) / PRE_TH, B), dim3(PRE_NT), 0, s, d_gray, pitch, fstride, h->d_ang, h->d_modgrad,
                       h->d_cs, h->d_cs0, g, h->taps, h->d_xofs, h->d_xa, h->d_yofs, h->d_yb, d_cost);
    if (balance) {
        // (here, in front of the Sobel kernel, not in front of k_lsd_regions2: the region kernel must follow its predecessor back to back -- in the 30 us of a small
        // kernel the ORB tiles of the step, released by ev_front, took the CUs first and the region stage went from 68 to 110 ms inside bench.py)
        const int wpg = line_wpg(h->tune, B);
        PLF_HIP_TRY(hipMemsetAsync(d_perm, 0xFF, (size_t)((B + wpg - 1) / wpg) * wpg * sizeof(int), s));   // (-1: wave slots past the batch)
        hipLaunchKernelGGL(k_lsd_balance, dim3((B + 255) / 256), dim3(256), 0, s, d_cost, d_perm, B, wpg);
    }
    if (h->prm.lbd_sobel_input == PLF_LBD_RAW)
        hipLaunchKernelGGL(k_sobel3, dim3((((g.w + 3) / 4) * g.h + 255) / 256, 1, B), dim3(256), 0, s, d_gray, pitch, fstride, h->d_grad, g);
    else
        hipLaunchKernelGGL(k_blur5_sobel3, dim3((g.w + BS_TW - 1) / BS_TW, (g.h + BS_TH - 1) / BS_TH, B), dim3(256), 0, s, d_gray, pitch, fstride, h->d_grad, g, h->blur5);
    if (!h->ev_front) PLF_HIP_TRY(hipEventCreateWithFlags(&h->ev_front, hipEventDisableTiming));
    PLF_HIP_TRY(hipEventRecord(h->ev_front, s));
    h->ev_front_set = true;
    const bool prof = h->prof_on && h->prof_n < 512;
    if (prof) {
        if (!h->prof_ev[2 * h->prof_n]) { (void)hipEventCreate(&h->prof_ev[2 * h->prof_n]); (void)hipEventCreate(&h->prof_ev[2 * h->prof_n + 1]); }
        (void)hipEventRecord(h->prof_ev[2 * h->prof_n], s);
    }
    const bool budget = g.budget_ticks != 0u;
    const uint32_t *seeds = nullptr;
    if (h->prm.seed_order == 1) {   // published LSD order: bins descending, raster inside a bin
        std::vector<int> off(2 * (size_t)B);
        for (int f = 0; f < B; f++) { off[f] = (int)((size_t)f * g.s_stride); off[B + f] = off[f] + g.sw * g.sh; }
        PLF_HIP_TRY(hipStreamSynchronize(s));   // a previous sort may still read the offsets
        PLF_HIP_TRY(hipMemcpyAsync(h->d_seg_off, off.data(), sizeof(int) * 2 * B, hipMemcpyHostToDevice, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
        hipLaunchKernelGGL(k_lsd_maxgrad, dim3(B), dim3(256), 0, s, h->d_ang, h->d_modgrad, h->d_maxgrad, g);
        hipLaunchKernelGGL(k_lsd_seedkeys, dim3((g.sw * g.sh + 255) / 256, B), dim3(256), 0, s, h->d_ang, h->d_modgrad, h->d_maxgrad, h->d_keys[0], g);
        size_t tmp = h->sort_tmp_bytes;
        PLF_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(h->d_sort_tmp, tmp, h->d_keys[0], h->d_keys[1], (int)((size_t)h->prm.max_batch * g.s_stride), B,
                                                               h->d_seg_off, h->d_seg_off + B, 0, 30, s));
        seeds = h->d_keys[1];
    }
    // up to one frame per XCD: latency mode (the maps of one frame fit the XCD's 4 MB L2); otherwise the batch hides the latency
    const LineTune &T = h->tune;
    const int lat_max = T.lat_max;
    // measured on one MI355X (VGA, frames/s with 16 / 8 / 4 / 2 bands): 8 frames 448 / 378 / 291 / 197; 32: 1277 / 1252 / 1079 / 679; 128: 3392 / 4321 /
    // 4007 / 1511; 256: 3866 / 5713 / 6800 / 5381; 512: - / 7237 / 8352 / 8856 (serial kernel: 7220); at 1024 the batch itself hides the latency
    // of the one-wave-per-frame kernel (9.8k with 2 bands vs 11.9k)
    // Round 6 (tools/midrange_sweep2.sh): with the validation rounds up to 256 frames in flight (spec_z) the best band count keeps frames x bands around 768-2048 --
    // a little above the 512 band workgroups the chip holds at the validation's LDS size: 32 frames 3.2 k -> 4.0 k frames/s (24 bands), 64: 5.2 k -> 6.1 k (12), 128: 8.5 k
    // -> 9.4 k (8), 256: 9.4 k -> 10.1 k (8); above 256 the one-wave-per-frame kernel with 2 frames per workgroup wins (line_wpg).
    // One frame in flight: 64 bands (tools/experiments/r06/bands_one_frame.py: LSD+LBD call, mean over 24 polygon scenes / 16 natural-image-like frames / 14 windows of real
    // photographs, 48 -> 64 bands: 3.94 -> 3.89, 7.47 -> 7.20, 10.98 -> 10.35 ms; from two frames on the two are equal within the noise).
    const int spec_bands_req = T.spec_bands != PLF_TUNE_AUTO ? T.spec_bands : (B <= 1 ? 64 : B <= 8 ? 48 : B <= 16 ? 32 : B <= 32 ? 24 : B <= 64 ? 12 : B <= 256 ? 8 : B <= 384 ? 4 : 2);
    const int spec_max = T.spec_max;
    const int bm_words = (g.sw * g.sh + 31) / 32, list_words = (g.rcap + 1 + 15) & ~15;
    const int coarse_words = (((((g.sw + 7) >> 3) + 31) & ~31) >> 5) * ((g.sh + 7) >> 3);   // tile rows padded to whole words (spec_commit_body)
    // commit wave: T and S in LDS when they fit, otherwise S in global memory
    const bool s_global = (size_t)(list_words + 2 * bm_words + coarse_words + 5 * 512 + 512 / 32 + 1024 + 33) * 4 + 64 > 150 * 1024;
    const int commit_extra = 5 * 512 + 512 / 32 + 1024 + 1 + 16 + 16;   // SPEC_COMMIT_EXTRA_WORDS of lsd_kernels.hip (+ the 16-word alignment of the tile map): record headers, SUSPECT mask, "defined, no record" bits
    const size_t lds_grow = (size_t)(list_words + bm_words) * 4, lds_commit = (size_t)(list_words + (s_global ? 1 : 2) * bm_words + coarse_words + commit_extra) * 4 + 64;
    int spec_bands = spec_bands_req;
    if (T.spec_bands == PLF_TUNE_AUTO && B <= 16) {
        // every band workgroup of the batch should be resident at once: a band wave and its validation hold the frame's flag bitmap(s) in LDS -- 29 / 68 KB at
        // VGA, 103 / 119 KB at 1280x960, i.e. one workgroup per CU there: 8 such frames x 48 bands ran as two rounds of workgroups (277 frames/s; 372 with 32 bands)
        const int per_cu = (int)std::max<size_t>(1, (size_t)(160 * 1024) / std::max(lds_grow, lds_commit));
        if (h->n_cus <= 0) { hipDeviceProp_t prop; h->n_cus = hipGetDeviceProperties(&prop, h->device) == hipSuccess ? prop.multiProcessorCount : 256; (void)hipGetLastError(); }
        const int fit = h->n_cus * per_cu / B;
        if (fit < spec_bands) spec_bands = std::max(8, fit & ~7);
    }
    bool spec = !seeds && g.sh <= 8192 && (g.sh - 1) / 4 >= spec_bands && spec_bands >= 2 && spec_bands <= 64 && B <= spec_max && lds_commit <= 150 * 1024 && h->spec_frames >= 0;
    // validation rounds instead of the serial commit wave (k_lsd_spec_validate): up to PLF_LSD_SPEC_Z frames in flight (16), never with a time budget
    const int zmax = T.spec_z;
    bool zmode = spec && !budget && B <= zmax;
    if (spec) {   // scratch of the speculation: ~23 MB per VGA frame, ~65 MB per 1280x960 frame (x 2.5 with the buffers of the validation rounds); keep it below 32 GiB
        size_t Fr = 8;
        while (Fr < (size_t)B) Fr <<= 1;
        const size_t per_frame = ((size_t)2 * spec_bands + 3) * g.s_stride * sizeof(uint32_t) + (size_t)spec_bands * 8192 * sizeof(SpecRec);
        const size_t per_frame_z = (size_t)spec_bands * (3 * g.s_stride * sizeof(uint32_t) + 8192 * sizeof(SpecRec) + 3 * (size_t)bm_words * sizeof(uint32_t));
        if (Fr * per_frame > ((size_t)8 << 30)) spec = false;
        if (zmode && Fr * (per_frame + per_frame_z) > ((size_t)32 << 30)) zmode = false;
        if (!spec) zmode = false;
    }
    // (the band count depends on the batch size -- 48 / 32 / 16 / ... -- and every change used to free and re-allocate hundreds of MB behind a stream
    // synchronisation, e.g. in a loop that mixes 1-frame and 12-frame calls.  The buffers are indexed by (frame * nbands + band) with the CURRENT band count as the
    // stride, so an allocation made for Fr frames x K bands serves every call with B <= Fr and B * nbands <= Fr * K.)
    if (spec && (h->spec_frames < B || (size_t)B * spec_bands > h->spec_slots || h->spec.bm_words != bm_words || (size_t)h->spec.tcap != g.s_stride || (zmode && !h->spec.out) || h->spec.rcap_rec != T.spec_reccap)) {
        // (re)allocate for lat_max frames of the current geometry
        void *old[] = {h->spec.rxy, h->spec.tl, h->spec.recs, h->spec.cnt, h->spec.seedmap, h->spec.defmap, h->spec.tl2, h->spec.band_y, h->spec.done, h->spec.sglob, h->spec.halo, h->d_spec_stats,
                       h->spec.out, h->spec.pre, h->spec.tl_alt, h->spec.recs_alt, h->spec.cnt_alt, h->spec.side, h->spec.nrects, h->spec.reach, h->spec.round_state, h->spec.tl2b, h->spec.band_ticks, h->spec.round_log, h->d_spec_rowcnt};
        PLF_HIP_TRY(hipStreamSynchronize(s));
        for (void *q : old) if (q) (void)hipFree(q);
        memset(&h->spec, 0, sizeof(h->spec)); h->d_spec_stats = nullptr; h->d_spec_rowcnt = nullptr;
        size_t Fr = 8;   // frames the buffers are sized for: the batch rounded up to a power of two (~23 MB per VGA frame)
        while (Fr < (size_t)B) Fr <<= 1;
        const size_t K = (size_t)spec_bands;
        h->spec.nbands = spec_bands; h->spec.bm_words = bm_words; h->spec.tcap = (int)g.s_stride; h->spec.frames_cap = (int)Fr;
        h->spec.rcap_rec = T.spec_reccap;   // (8192; test hook: a small value forces the overflow fallback -- a change re-allocates, see the condition above)
        bool ok = hipMalloc((void **)&h->spec.rxy, Fr * K * g.s_stride * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.tl, Fr * K * (size_t)h->spec.tcap * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.recs, Fr * K * (size_t)h->spec.rcap_rec * sizeof(SpecRec)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.cnt, Fr * K * 4 * sizeof(int)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.seedmap, Fr * (size_t)bm_words * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.defmap, Fr * (size_t)bm_words * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.tl2, Fr * 2 * g.s_stride * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.band_y, Fr * (K + 1) * sizeof(int)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.done, Fr * K * sizeof(int)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.sglob, Fr * (zmode ? K : 1) * (size_t)bm_words * sizeof(uint32_t)) == hipSuccess &&   // (one per band for the validation rounds)
                  hipMalloc((void **)&h->spec.side, Fr * K * sizeof(int)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.band_ticks, Fr * K * 2 * sizeof(int)) == hipSuccess &&
                  hipMalloc((void **)&h->spec.halo, Fr * K * (size_t)bm_words * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&h->d_spec_stats, (Fr * 8 + 200) * sizeof(int)) == hipSuccess &&
                  hipMalloc((void **)&h->d_spec_rowcnt, Fr * (1024 + 64 * 65) * sizeof(int)) == hipSuccess;
        if (ok && zmode) {
            ok = hipMalloc((void **)&h->spec.out, Fr * K * (size_t)bm_words * sizeof(uint32_t)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.pre, Fr * K * (size_t)bm_words * sizeof(uint32_t)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.tl_alt, Fr * K * (size_t)h->spec.tcap * sizeof(uint32_t)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.recs_alt, Fr * K * (size_t)h->spec.rcap_rec * sizeof(SpecRec)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.cnt_alt, Fr * K * 4 * sizeof(int)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.nrects, Fr * K * sizeof(int)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.reach, Fr * K * 2 * sizeof(int)) == hipSuccess &&   // (the rows a band's records reach, spec_reach: indexed by slot, whatever the call's band count)
                 hipMalloc((void **)&h->spec.round_state, Fr * 5 * sizeof(int)) == hipSuccess &&
                 hipMalloc((void **)&h->spec.tl2b, Fr * K * 2 * g.s_stride * sizeof(uint32_t)) == hipSuccess;
        }
#ifdef PLF_ROUND_LOG
        if (ok && zmode && getenv("PLF_LSD_ROUND_LOG")) ok = hipMalloc((void **)&h->spec.round_log, Fr * K * 64 * sizeof(int)) == hipSuccess;
#endif
        if (!ok) { (void)hipGetLastError(); h->spec_frames = -1; spec = false; zmode = false; }
        else { h->spec_frames = (int)Fr; h->spec_slots = Fr * K; h->spec_sglob_per_band = zmode; }
    }
    if (spec) {
        h->spec.nbands = spec_bands;   // (the stride of this call; the allocation may hold more)
        h->spec.s_global = s_global ? 1 : 0;
        h->spec.spin_bound = T.spec_spins;   // (test hook: a short bound must still let slow band waves finish -- heartbeat)
        // warm-up rows above a band: 16 when a serial commit wave follows (every region it has to regrow is serial time), 4 with the validation rounds
        // (the bands redo their conflicts in parallel; the warm-up rows are band-wave time): single VGA frame 7.2 / 7.2 / 6.8 ms with 12 / 8 / 4 rows
        h->spec.halo_rows = T.spec_halo != PLF_TUNE_AUTO ? T.spec_halo : (zmode ? 4 : 16);
        // validation rounds: the band's guess of what the earlier bands take from it can be made WITHOUT growing anything (k spec_grow_body: fill): rows and tolerance
        h->spec.fill_rows = zmode ? T.spec_fill : 0;
        h->spec.fill_tol_deg = T.spec_fill_tol;
        if (h->spec.fill_rows > 0) h->spec.halo_rows = 0;
        // rows below the band the warm-up regions may reach (< 0: unbounded).  Clipping shortens the band waves (2.79 -> 2.51 ms, one VGA frame) and lengthens the
        // validation rounds (2.08 -> 2.77 ms): it loses for one frame (6.0 vs 6.5 ms) and wins once a round lasts as long as the slowest band of several frames
        // anyway (8 frames in flight: 977 -> 1014 frames/s).  Round 6, with 64 bands for one frame and the refined validity rule: 16 rows for one frame too (polygons
        // 3.89 -> 3.87 ms, natural-image-like 7.21 -> 7.03, real photographs 10.3 -> 9.9: tools/experiments/r06/halo_rows.py)
        h->spec.halo_clip = T.spec_clip != PLF_TUNE_AUTO ? T.spec_clip : 16;
        // one launch while all its workgroups fit the chip at the commit wave's LDS size (2 per CU): the commit wave follows the bands as they
        // finish, and the bands are staggered for that; otherwise two launches with equal bands
        // ... and only while EVERY workgroup of the launch can be resident at the same time: the commit workgroups wait for band workgroups of the
        // same launch, and HIP does not promise a dispatch order (occupancy query per launch configuration, cached)
        if (h->fused_lds != lds_commit) {
            int per_cu = 0, cus = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_lsd_spec_fused, 256, lds_commit) != hipSuccess ||
                hipGetDeviceProperties(&prop, h->device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; cus = 0; }
            else cus = prop.multiProcessorCount;
            h->fused_lds = lds_commit; h->fused_capacity = (size_t)per_cu * (size_t)cus;
        }
        zmode = zmode && h->spec.out != nullptr;
        const bool fused = !zmode && (size_t)B * (spec_bands + 1) <= 448 && (size_t)B * (spec_bands + 1) <= h->fused_capacity && !T.spec_nofuse;
        h->spec.stagger = fused ? T.spec_stagger : 0.f;
        // seed map, band flags, row counts, round state: one launch (k_lsd_spec_clear)
        hipLaunchKernelGGL(k_lsd_spec_clear, dim3(B), dim3(256), 0, s, h->spec, h->d_spec_rowcnt, zmode ? 1 : 0);
        // band boundaries: row counts + defined-pixel bitmap over the whole frame in parallel, then one wave per frame balances the bands
        hipLaunchKernelGGL(k_lsd_spec_rows, dim3((bm_words * 32 + 255) / 256, B), dim3(256), 0, s, h->d_ang, g, h->spec, h->d_spec_rowcnt);
        hipLaunchKernelGGL(k_lsd_spec_bands, dim3(B), dim3(64), 0, s, g, h->spec, h->d_spec_rowcnt, h->d_spec_rowcnt + (size_t)h->spec_frames * 1024);
        SpecBufs SBn = h->spec; SBn.out = nullptr; SBn.round_state = nullptr;   // (schedules without validation rounds: the band waves skip their part of them)
        if (zmode) {
            // band waves (equal shares), then rounds in which every band validates itself against what the bands before it mark, all bands at once;
            // rectangles assembled from the bands' logs; frames that did not reach the fixpoint within the rounds are committed serially (exact either way)
            const int rounds = T.spec_rounds;   // (3-10 rounds reach the fixpoint on the synthetic frames, up to 14 on real photographs; a launch of a frame that has converged returns at once: ~4 us per idle round)
#ifdef PLF_ROUND_LOG
            if (h->spec.round_log) PLF_HIP_TRY(hipMemsetAsync(h->spec.round_log, 0, (size_t)h->spec_slots * 64 * sizeof(int), s));
#endif
            const SpecBufs SBz = h->spec;
            hipLaunchKernelGGL(k_lsd_spec_grow, dim3(spec_bands, B), dim3(256), lds_grow, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, g, SBz);   // (waves 1-3 warm the L2)
            for (int r = 1; r <= rounds; r++) {
                hipLaunchKernelGGL(k_lsd_spec_prefix, dim3((bm_words + 255) / 256, B), dim3(256), 0, s, SBz, r);
                hipLaunchKernelGGL(k_lsd_spec_validate, dim3(spec_bands, B), dim3(256), lds_commit, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, g, SBz, r);
            }
            hipLaunchKernelGGL(k_lsd_spec_assemble, dim3(spec_bands, B), dim3(64), 0, s, h->d_rects, nrect, status, g, SBz, rounds);
            hipLaunchKernelGGL(k_lsd_spec_commit_rest, dim3(B), dim3(256), lds_commit, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, h->d_rxy, h->d_rects, nrect, status, g,
                               SBz, h->d_spec_stats, rounds);
        } else if (fused) {
            hipLaunchKernelGGL(budget ? k_lsd_spec_fused_budget : k_lsd_spec_fused, dim3(B * (spec_bands + 1)), dim3(256), lds_commit, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, h->d_rxy, h->d_rects, nrect,
                               status, g, SBn, h->d_spec_stats, B);
        } else {
        hipLaunchKernelGGL(budget ? k_lsd_spec_grow_budget : k_lsd_spec_grow, dim3(spec_bands, B), dim3(256), lds_grow, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, g, SBn);
        hipLaunchKernelGGL(budget ? k_lsd_spec_commit_budget : k_lsd_spec_commit, dim3(B), dim3(256), lds_commit, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, h->d_rxy, h->d_rects, nrect, status, g,
                           SBn, h->d_spec_stats);
        }
    } else if (B <= lat_max)
        hipLaunchKernelGGL(budget ? k_lsd_regions_lat_budget : k_lsd_regions_lat, dim3(B), dim3(256), h->regions_lds, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, h->d_rxy,
                           h->d_rects, nrect, status, g, seeds, status);
    else {   // 8 frames per workgroup (one wave each), 5120 + 1024 bytes of LDS per wave: see k_lsd_regions2 / regions_body
        const int wpg = line_wpg(T, B);
        const size_t wave_lds = PLF_LSD_WAVE_LDS;
        hipLaunchKernelGGL(budget ? k_lsd_regions2_budget : k_lsd_regions2, dim3((B + wpg - 1) / wpg), dim3(64 * wpg), wpg * wave_lds, s, h->d_ang, h->d_modgrad, h->d_cs, h->d_cs0, h->d_rxy,
                           h->d_rects, nrect, status, g, seeds, B, balance ? d_perm : nullptr);
    }
    if (prof) { (void)hipEventRecord(h->prof_ev[2 * h->prof_n + 1], s); h->prof_n++; }
    // rect_improve.  Few frames in flight: one wave per rectangle runs all five stages (k_nfa_fused: a rectangle only waits for itself); otherwise the staged
    // kernels: first evaluation + 5 search stages, each = (wave-parallel pixel count, lane-parallel NFA math) over work lists compacted over the batch
    const int nfa_fused_max = T.nfa_fused;
    if (T.nfa_table && h->nfa_tab_log_nt != g.log_nt) {   // (first batch of this image size)
        hipLaunchKernelGGL(k_nfa_table, dim3(NFA_TAB_N, NFA_TAB_P), dim3(256), 0, s, h->d_nfa_tab, h->d_lgam, g.log_nt);
        h->nfa_tab_log_nt = g.log_nt;
    }
    const bool small_first = T.nfa_table && (T.nfa_small >= 2 || (T.nfa_small == 1 && B > nfa_fused_max));
    if (!small_first && B <= nfa_fused_max) {
        hipLaunchKernelGGL(k_nfa_fused, dim3(1024, B), dim3(64), 0, s, h->d_ang, h->d_lgam, T.nfa_table ? h->d_nfa_tab : nullptr, h->d_rects, nrect, h->d_keep, h->d_seg, g);
    } else {
        PLF_HIP_TRY(hipMemsetAsync(h->d_nfa_counters, 0, 16 * sizeof(int), s));
        if (small_first)   // rectangles of fewer than 512 pixels: all five stages by 16 lanes, values from the table; the others are queued in the stage-0 work list
        {
            // two passes for large batches: stage 0 of every rectangle, then stages 1-4 of the two thirds it does not decide (queued in the second state buffer, which the
            // staged kernels only write later in the stream)
            NfaState *surv = (T.nfa_two_pass && B > nfa_fused_max) ? h->d_st[1] : nullptr;
            const int scap = (int)std::min<size_t>((size_t)g.nfa_pool / (size_t)B, (size_t)g.rect_cap);   // queue entries per frame
            if (surv) PLF_HIP_TRY(hipMemsetAsync(h->d_nfa_fcnt, 0, (size_t)B * sizeof(int), s));
            hipLaunchKernelGGL(k_nfa_small, dim3(8 * (g.rect_cap < 640 ? (g.rect_cap + 3) / 4 : 160), (B + 7) / 8), dim3(64), 0, s, h->d_ang, h->d_nfa_tab, h->d_rects, nrect,
                               h->d_keep, h->d_seg, h->d_ent[0], h->d_st[0], h->d_nfa_counters, status, g, B, surv, h->d_nfa_fcnt, scap);
            if (surv)
                hipLaunchKernelGGL(k_nfa_small2, dim3(8 * (g.rect_cap < 512 ? (g.rect_cap + 3) / 4 : 128), (B + 7) / 8), dim3(64), 0, s, h->d_ang, h->d_nfa_tab, h->d_rects, h->d_keep,
                                   h->d_seg, h->d_ent[0], h->d_st[0], h->d_nfa_counters, status, g, B, surv, h->d_nfa_fcnt, scap);
        }
        else
            hipLaunchKernelGGL(k_nfa_init, dim3(g.rect_cap < 4096 ? (g.rect_cap + 255) / 256 : 16, B), dim3(256), 0, s, h->d_rects, nrect, h->d_keep, h->d_ent[0], h->d_st[0],
                               h->d_nfa_counters, status, g);
        {
            hipLaunchKernelGGL(k_nfa_clamp, dim3(1), dim3(1), 0, s, h->d_nfa_counters, status, g);
#ifndef PLF_NFA_COUNT_WAVES
#define PLF_NFA_COUNT_WAVES (256 * 64)   // persistent waves of k_nfa_count: 16 per SIMD offered (occupancy by VGPRs: 8); the kernel is HBM-latency bound
#endif
            const int count_waves = PLF_NFA_COUNT_WAVES, math_blocks = 1024;
            for (int stage = 0; stage <= 4; stage++) {
                const int in = stage & 1, out = in ^ 1;
                if (stage >= 1 && stage <= 3)
                    hipLaunchKernelGGL(small_first ? k_nfa_count1_w : k_nfa_count1, dim3(count_waves), dim3(64), 0, s, h->d_ang, h->d_ent[in], h->d_nfa_counters, stage, 5,
                                       h->d_cnt, g);
                else
                    hipLaunchKernelGGL(small_first ? k_nfa_count_w : k_nfa_count, dim3(count_waves), dim3(64), 0, s, h->d_ang, h->d_ent[in], h->d_nfa_counters, stage, 1,
                                       h->d_cnt, g);
                hipLaunchKernelGGL(k_nfa_eval, dim3(2 * math_blocks), dim3(256), 0, s, stage, h->d_lgam, T.nfa_table ? h->d_nfa_tab : nullptr, h->d_cnt, h->d_ent[in],
                                   h->d_nfa_counters, h->d_vals, g);
                hipLaunchKernelGGL(k_nfa_math, dim3(math_blocks), dim3(64), 0, s, stage, h->d_vals, h->d_ent[in], h->d_st[in], h->d_st[out],
                                   h->d_ent[out], h->d_nfa_counters, h->d_seg, h->d_keep, g);
            }
        }
    }
    hipLaunchKernelGGL(k_lsd_finalize, dim3(B), dim3(256), h->finalize_lds, s, h->d_seg, h->d_keep, nrect, h->d_segs_out, nseg, h->d_kl_tmp,
                       d_lines, d_eq, d_nout, capacity, status, h->d_sort_scratch, g);
    hipLaunchKernelGGL(k_lbd, dim3(capacity < g.nkeep ? capacity : g.nkeep, B), dim3(64), 0, s, h->d_grad, d_lines, d_nout, d_ldesc, capacity, g,
                       h->d_lbd);
    PLF_HIP_TRY(hipGetLastError());
    h->last_frames = B;
    return PLF_OK;
}

extern "C" int plf_line_extract_batch(plf_line *h, const uint8_t *gray, int32_t in_mem, int32_t n_frames, int32_t width, int32_t height,
                                      ptrdiff_t pitch, ptrdiff_t frame_stride, plf_keyline *lines, uint8_t *ldesc, double *line_eq,
                                      int32_t *n_out, int32_t out_mem, int32_t capacity, void *stream)
{
    if (!h) return PLF_E_BADARG;
    if (!gray || width <= 0 || height <= 0 || n_frames <= 0) return PLF_E_EMPTY;
    if (n_frames > h->prm.max_batch || width > h->prm.max_width || height > h->prm.max_height || pitch < width || !lines || !ldesc ||
        !line_eq || !n_out || capacity < 1)
        return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    int rc = line_configure(h, width, height);
    if (rc != PLF_OK) return rc;
    if (h->retry_depth == 0) { h->retry_valid = false; h->retry_status = 0; h->slow_last = false; }
    const auto t_call0 = std::chrono::steady_clock::now();
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    // handle-owned scratch is ordered by the stream of the previous call: a call on another stream waits for it first (include/plf.h, "Streams")
    plf_order_begin(h->order, s);
    PlfOrderGuard order_guard_{h->order, s};
    h->last_stream = s; h->last_stream_set = true;
    const uint8_t *d_gray = gray;
    ptrdiff_t dpitch = pitch, dfstride = frame_stride;
    if (in_mem == PLF_MEM_HOST) {
        dpitch = width; dfstride = (ptrdiff_t)width * height;
        for (int f = 0; f < n_frames; f++)
            PLF_HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)f * dfstride, dpitch, gray + (size_t)f * frame_stride, pitch, width, height,
                                         hipMemcpyHostToDevice, s));
        d_gray = h->d_in;
    }
    const bool host_out = out_mem == PLF_MEM_HOST;
    const int cap_dev = host_out ? h->prm.nlines : capacity;
    plf_keyline *d_lines = host_out ? h->d_lines : lines;
    uint8_t *d_ldesc = host_out ? h->d_ldesc : ldesc;
    double *d_eq = host_out ? h->d_lineeq : line_eq;
    int *d_nout = host_out ? h->d_counters + 2 * (size_t)h->prm.max_batch : n_out;
    rc = line_enqueue(h, d_gray, n_frames, dpitch, dfstride, d_lines, d_ldesc, d_eq, d_nout, cap_dev, s);
    if (rc != PLF_OK) return rc;
    if (!host_out && in_mem == PLF_MEM_DEVICE) return PLF_OK;
    int status = 0;
    int ret = PLF_OK;
    // A few frames in flight (the drop-in loop): status, counts and the three result arrays come back through pinned memory with ONE synchronisation -- five
    // copies into pageable memory, the first two followed by a wait for the counts, cost 150 us of a 6 ms call.
    const int PIN_FRAMES = 8;
    const size_t per_frame = (size_t)cap_dev * (sizeof(plf_keyline) + 32 + 3 * sizeof(double));
    if (host_out && n_frames <= PIN_FRAMES) {
        const size_t need = 64 + sizeof(int) * PIN_FRAMES + (size_t)PIN_FRAMES * per_frame + 16;
        if (!h->h_pin || h->h_pin_bytes < need) {
            if (h->h_pin) { (void)hipHostFree(h->h_pin); h->h_pin = nullptr; }
            PLF_HIP_TRY(hipHostMalloc((void **)&h->h_pin, need, hipHostMallocDefault));
            h->h_pin_bytes = need;
        }
        int *p_status = (int *)h->h_pin, *p_cnt = (int *)(h->h_pin + 64);
        uint8_t *p_lines = h->h_pin + 64 + sizeof(int) * PIN_FRAMES;
        uint8_t *p_eq = p_lines + (size_t)n_frames * cap_dev * sizeof(plf_keyline);      // (doubles: 8-byte aligned, plf_keyline is 68 bytes -- cap_dev * n * 68 is a multiple of 4 only)
        p_eq += (8 - ((uintptr_t)p_eq & 7)) & 7;
        uint8_t *p_desc = p_eq + (size_t)n_frames * cap_dev * 3 * sizeof(double);
        PLF_HIP_TRY(hipMemcpyAsync(p_status, h->d_counters + 3 * (size_t)h->prm.max_batch, sizeof(int), hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipMemcpyAsync(p_cnt, d_nout, sizeof(int) * n_frames, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipMemcpyAsync(p_lines, d_lines, (size_t)n_frames * cap_dev * sizeof(plf_keyline), hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipMemcpyAsync(p_eq, d_eq, (size_t)n_frames * cap_dev * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipMemcpyAsync(p_desc, d_ldesc, (size_t)n_frames * cap_dev * 32, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
        status = *p_status;
        if (!(status & (4 | 1))) {
            for (int f = 0; f < n_frames; f++) {
                int n = p_cnt[f];
                if (n > capacity) { n = capacity; ret = PLF_E_CAPACITY; }
                n_out[f] = n;
                if (n > 0) {
                    memcpy(lines + (size_t)f * capacity, p_lines + (size_t)f * cap_dev * sizeof(plf_keyline), sizeof(plf_keyline) * n);
                    memcpy(ldesc + (size_t)f * capacity * 32, p_desc + (size_t)f * cap_dev * 32, (size_t)32 * n);
                    memcpy(line_eq + (size_t)f * capacity * 3, p_eq + (size_t)f * cap_dev * 3 * sizeof(double), sizeof(double) * 3 * n);
                }
            }
        }
    } else {
    PLF_HIP_TRY(hipMemcpyAsync(&status, h->d_counters + 3 * (size_t)h->prm.max_batch, sizeof(int), hipMemcpyDeviceToHost, s));
    if (host_out) {
        std::vector<int> cnt(n_frames);
        PLF_HIP_TRY(hipMemcpyAsync(cnt.data(), d_nout, sizeof(int) * n_frames, hipMemcpyDeviceToHost, s));
        PLF_HIP_TRY(hipStreamSynchronize(s));
        for (int f = 0; f < n_frames; f++) {
            int n = cnt[f];
            if (n > capacity) { n = capacity; ret = PLF_E_CAPACITY; }
            n_out[f] = n;
            if (n > 0) {
                PLF_HIP_TRY(hipMemcpyAsync(lines + (size_t)f * capacity, d_lines + (size_t)f * cap_dev, sizeof(plf_keyline) * n, hipMemcpyDeviceToHost, s));
                PLF_HIP_TRY(hipMemcpyAsync(ldesc + (size_t)f * capacity * 32, d_ldesc + (size_t)f * cap_dev * 32, (size_t)32 * n, hipMemcpyDeviceToHost, s));
                PLF_HIP_TRY(hipMemcpyAsync(line_eq + (size_t)f * capacity * 3, d_eq + (size_t)f * cap_dev * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
            }
        }
    }
    PLF_HIP_TRY(hipStreamSynchronize(s));
    }
    if (status & 4) return PLF_E_HIP;   // the fused speculative launch gave up waiting for a band wave (never seen; see spec_wait_band)
    if (status & 1) {
        // The batch as a whole produced more rectangles than the pooled NFA buffers hold (thousands per frame on average: synthetic textures).
        // One frame always fits, so the batch is redone in halves.
        if (n_frames == 1) return PLF_E_RECTS;
        const int32_t h1 = n_frames / 2;
        int32_t *const flags0 = h->retry_flags;        // the pieces file their flags at their own frame offsets
        h->retry_depth++;
        const int ra = plf_line_extract_batch(h, gray, in_mem, h1, width, height, pitch, frame_stride, lines, ldesc, line_eq, n_out, out_mem, capacity,
                                              stream);
        h->retry_flags = flags0 + h1;
        const int rb = plf_line_extract_batch(h, gray + (size_t)h1 * frame_stride, in_mem, n_frames - h1, width, height, pitch, frame_stride,
                                              lines + (size_t)h1 * capacity, ldesc + (size_t)h1 * capacity * 32, line_eq + (size_t)h1 * capacity * 3,
                                              n_out + h1, out_mem, capacity, stream);
        h->retry_flags = flags0;
        h->retry_depth--;
        h->last_frames = n_frames;
        if (h->retry_depth == 0) h->retry_valid = true;
        if (ra != PLF_OK && ra != PLF_E_CAPACITY) return ra;
        if (rb != PLF_OK && rb != PLF_E_CAPACITY) return rb;
        return (ra == PLF_E_CAPACITY || rb == PLF_E_CAPACITY) ? PLF_E_CAPACITY : PLF_OK;
    }
    if (h->retry_depth > 0) {   // a piece of a batch that is being redone in halves: its flags and status bits go to the host copy
        h->retry_status |= status & (2 | 8);
        if (h->g.budget_ticks) {
            PLF_HIP_TRY(hipMemcpyAsync(h->retry_flags, h->d_counters + 3 * (size_t)h->prm.max_batch + 16, sizeof(int32_t) * n_frames, hipMemcpyDeviceToHost, s));
            PLF_HIP_TRY(hipStreamSynchronize(s));
        } else memset(h->retry_flags, 0, sizeof(int32_t) * n_frames);
    }
    if (status & 2) ret = PLF_E_CAPACITY;
    // PLF_W_SLOW (include/plf.h): this call against the handle's own history at this image size.  Host-output calls only (they end with a synchronisation, so the
    // wall time is the work's); pieces of a batch redone in halves and calls that ran out of max_ms do not enter the history.
    if (host_out && h->retry_depth == 0 && !(status & 8)) {
        const float ms = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call0).count() / n_frames);
        if (h->slow_w != width || h->slow_h != height) { h->slow_n = 0; h->slow_w = width; h->slow_h = height; }
        const int have = std::min(h->slow_n, 32);
        if (have >= 8 && h->tune.slow_factor > 0.f) {
            float tmp[32];
            memcpy(tmp, h->slow_hist, sizeof(float) * have);
            std::nth_element(tmp, tmp + have / 2, tmp + have);
            if (ms > h->tune.slow_factor * tmp[have / 2] && ms > h->tune.slow_floor_ms) { h->slow_last = true; if (ret == PLF_OK) ret = PLF_W_SLOW; }
        }
        h->slow_hist[h->slow_n % 32] = ms;
        h->slow_n = h->slow_n < (1 << 30) ? h->slow_n + 1 : 32;   // (calls so far at this size: the ring's write index, and min(n, 32) its fill)
    }
    return ret;
}

extern "C" int plf_line_last_status(plf_line *h, void *stream)
{
    if (!h) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    // handle-owned scratch is ordered by the stream of the previous call: a call on another stream waits for it first (include/plf.h, "Streams")
    plf_order_begin(h->order, s);
    PlfOrderGuard order_guard_{h->order, s};
    h->last_stream = s; h->last_stream_set = true;
    int status = 0;
    PLF_HIP_TRY(hipMemcpyAsync(&status, h->d_counters + 3 * (size_t)h->prm.max_batch, sizeof(int), hipMemcpyDeviceToHost, s));
    PLF_HIP_TRY(hipStreamSynchronize(s));
    if (h->retry_valid) status = (status & ~1) | h->retry_status;   // a batch redone in halves: the bits of all its pieces
    return (status & 4) ? PLF_E_HIP : (status & 1) ? PLF_E_RECTS : (status & 2) ? PLF_E_CAPACITY : (status & 8) ? PLF_W_TRUNCATED : h->slow_last ? PLF_W_SLOW : PLF_OK;
}

// which frames of the last batch ran out of their time budget (plf_line_params.max_ms): flags[f] = 1 / 0
extern "C" int plf_line_truncated(plf_line *h, int32_t *flags, int32_t n)
{
    if (!h || !flags || n < 1 || n > h->last_frames) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    if (h->retry_valid) { memcpy(flags, h->retry_flags, sizeof(int32_t) * n); return PLF_OK; }   // the batch was redone in halves: flags collected piece by piece
    // on the handle's own stream, behind the event of the last call (never on the caller's stream of that call, which may be gone by now: ADVICE r03)
    hipStream_t s = h->stream;
    plf_order_begin(h->order, s);
    PlfOrderGuard order_guard_{h->order, s};
    PLF_HIP_TRY(hipMemcpyAsync(flags, h->d_counters + 3 * (size_t)h->prm.max_batch + 16, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    PLF_HIP_TRY(hipStreamSynchronize(s));
    return PLF_OK;
}

// batch driver (batch_host.hip): the status word of the batch just enqueued on `s`, copied to pinned host memory in stream order
// (host_flags: per-frame "ran out of max_ms" flags of the same batch, n of them; only fetched for handles with a time budget)
int plf_line_status_async(plf_line *h, int32_t *host_dst, int32_t *host_flags, int n, hipStream_t s)
{
    PLF_HIP_TRY(hipMemcpyAsync(host_dst, h->d_counters + 3 * (size_t)h->prm.max_batch, sizeof(int), hipMemcpyDeviceToHost, s));
    if (host_flags && n > 0 && h->g.budget_ticks)
        PLF_HIP_TRY(hipMemcpyAsync(host_flags, h->d_counters + 3 * (size_t)h->prm.max_batch + 16, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    return PLF_OK;
}

extern "C" int plf_line_extract(plf_line *h, const uint8_t *gray, int32_t width, int32_t height, ptrdiff_t pitch, plf_keyline *lines,
                                uint8_t *ldesc, double *line_eq, int32_t capacity, int32_t *n_out)
{
    return plf_line_extract_batch(h, gray, PLF_MEM_HOST, 1, width, height, pitch, (ptrdiff_t)pitch * height, lines, ldesc, line_eq, n_out,
                                  PLF_MEM_HOST, capacity, nullptr);
}

static void line_prof_collect(plf_line *h)
{
    for (int i = 0; i < h->prof_n; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(h->prof_ev[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]) == hipSuccess) {
            h->prof_ms += ms;
            h->prof_launches++;
        }
    }
    h->prof_n = 0;
}

extern "C" int plf_line_debug_spec_stats(plf_line *h, int32_t *out8)
{
    if (!h || !out8 || !h->d_spec_stats) return PLF_E_BADARG;
    PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy(out8, h->d_spec_stats, 8 * sizeof(int), hipMemcpyDeviceToHost));
    if (getenv("PLF_LSD_SPEC_TIMELINE") && h->spec.band_ticks) {   // band waves of frame 0: rows, run time, accepted pixels logged
        std::vector<int> bt(2 * (size_t)h->spec.nbands), by((size_t)h->spec.nbands + 1);
        PLF_HIP_TRY(hipMemcpy(bt.data(), h->spec.band_ticks, bt.size() * sizeof(int), hipMemcpyDeviceToHost));
        PLF_HIP_TRY(hipMemcpy(by.data(), h->spec.band_y, by.size() * sizeof(int), hipMemcpyDeviceToHost));
        if (h->spec.round_state) {
            int rs[4];
            PLF_HIP_TRY(hipMemcpy(rs, h->spec.round_state, sizeof(rs), hipMemcpyDeviceToHost));
            fprintf(stderr, "[plf] validation rounds of frame 0: bands changed in the last even / odd round %d / %d, converged early %d, fell back to the serial commit %d\n", rs[0], rs[1], rs[2], rs[3]);
        }
        fprintf(stderr, "[plf] band waves of frame 0 (rows: us, logged pixels):");
        for (int b = 0; b < h->spec.nbands; b++) fprintf(stderr, " %d-%d: %d, %d |", by[b], by[b + 1], bt[2 * b] / 100, bt[2 * b + 1]);
        fprintf(stderr, "\n");
    }
    if (getenv("PLF_LSD_SPEC_TIMELINE")) {   // per band: grow end, commit start (100 MHz ticks); last: commit end
        int tl[200];
        PLF_HIP_TRY(hipMemcpy(tl, h->d_spec_stats + 8, 200 * sizeof(int), hipMemcpyDeviceToHost));
        fprintf(stderr, "[plf] commit wave of frame 0, kilo-cycles: bulk commits %d, event scans %d, re-classification %d, segment set-up %d\n", tl[193], tl[194], tl[195], tl[196]);
        const int t0 = tl[1];
        fprintf(stderr, "[plf] speculation timeline (us after the commit of band 0 started):");
        for (int b = 0; b < h->spec.nbands && b < 63; b++) fprintf(stderr, " b%d grow_end %d commit_start %d |", b, (tl[3 * b] - t0) / 100, (tl[3 * b + 1] - t0) / 100);
        fprintf(stderr, " commit_end %d\n", (tl[3 * 63 + 2] - t0) / 100);
    }
    return PLF_OK;
}

// diagnostics (tools/spec_redo.py): per frame of the last batch that took the validation-round schedule, out[4 f ..] = {bands changed in the last even round, in the
// last odd round, round that found nothing left to change (0: none within the enqueued rounds), 1 if the frame was finished by the serial commit wave instead}
extern "C" int plf_line_debug_spec_rounds(plf_line *h, int32_t *out, int32_t n_frames)
{
    // (round_state holds frames_cap frames -- the last speculative allocation, not max_batch -- and is only meaningful for the frames of the last call: ADVICE r04)
    if (!h || !out || n_frames < 1 || n_frames > h->last_frames || n_frames > h->spec.frames_cap || !h->spec.round_state) return PLF_E_BADARG;
    PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy(out, h->spec.round_state, (size_t)n_frames * 4 * sizeof(int), hipMemcpyDeviceToHost));
    return PLF_OK;
}

#ifdef PLF_ROUND_LOG
// diagnostics (tools/round_log.py, -DPLF_ROUND_LOG builds): out[((f * nbands + band) * 16 + round - 1) * 4 ..] = {100 MHz ticks, seeds regrown, pixels regrown, records that stood};
// returns the band count of the last call
extern "C" int plf_line_debug_round_log(plf_line *h, int32_t *out, int32_t n_frames, int32_t *band_ticks_out)
{
    if (!h || !out || !h->spec.round_log || n_frames < 1 || (size_t)n_frames * h->spec.nbands > h->spec_slots) return PLF_E_BADARG;
    PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy(out, h->spec.round_log, (size_t)n_frames * h->spec.nbands * 64 * sizeof(int), hipMemcpyDeviceToHost));
    if (band_ticks_out) PLF_HIP_TRY(hipMemcpy(band_ticks_out, h->spec.band_ticks, (size_t)n_frames * h->spec.nbands * 2 * sizeof(int), hipMemcpyDeviceToHost));
    return h->spec.nbands;
}
#endif

// diagnostics (tools/nfa_stats.py): rectangles that entered each rect_improve stage of the last staged batch (out16[0..5]; [5] = not meaningful after the last stage)
extern "C" int plf_line_debug_nfa_counters(plf_line *h, int32_t *out16)
{
    if (!h || !out16 || !h->d_nfa_counters) return PLF_E_BADARG;
    PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy(out16, h->d_nfa_counters, 16 * sizeof(int), hipMemcpyDeviceToHost));
    return PLF_OK;
}

// diagnostics (bench.py): length of each frame's region-growing chain in the last batch = pixels left marked USED (k_lsd_count_used; serial-chain
// schedules only: the speculative schedule keeps its flags in LDS and reports 0)
extern "C" int plf_line_chain_lengths(plf_line *h, int32_t *out, int32_t n)
{
    if (!h || !out || n < 1 || n > h->last_frames) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    PLF_HIP_TRY(hipDeviceSynchronize());
    int *d_out = h->d_counters + 4 * (size_t)h->prm.max_batch + 16;
    hipLaunchKernelGGL(k_lsd_count_used, dim3(n), dim3(256), 0, h->stream, h->d_ang, d_out, h->g);
    PLF_HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    PLF_HIP_TRY(hipStreamSynchronize(h->stream));
    return PLF_OK;
}

// diagnostics (bench.py): rectangles per frame of the last batch that went into the NFA validation
extern "C" int plf_line_rect_counts(plf_line *h, int32_t *out, int32_t n)
{
    if (!h || !out || n < 1 || n > h->last_frames) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    PLF_HIP_TRY(hipDeviceSynchronize());
    PLF_HIP_TRY(hipMemcpy(out, h->d_counters, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return PLF_OK;
}

extern "C" int plf_line_wait_front(plf_line *h, void *stream)
{
    if (!h) return PLF_E_BADARG;
    if (!h->ev_front_set) return PLF_OK;
    PLF_HIP_TRY(hipSetDevice(h->device));
    PLF_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->ev_front, 0));
    return PLF_OK;
}

extern "C" int plf_line_profile(plf_line *h, int32_t enable, int32_t reset, double *ms_total, int32_t *launches)
{
    if (!h) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    PLF_HIP_TRY(hipDeviceSynchronize());
    line_prof_collect(h);
    if (ms_total) *ms_total = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    if (reset) { h->prof_ms = 0; h->prof_launches = 0; }
    h->prof_on = enable ? 1 : 0;
    return PLF_OK;
}

extern "C" int plf_line_get_segments(plf_line *h, int32_t frame, float *segs, int32_t capacity, int32_t *n_out)
{
    if (!h || !n_out || frame < 0 || frame >= h->last_frames) return PLF_E_BADARG;
    PLF_HIP_TRY(hipSetDevice(h->device));
    PLF_HIP_TRY(hipDeviceSynchronize());
    int n = 0;
    PLF_HIP_TRY(hipMemcpy(&n, h->d_counters + (size_t)h->prm.max_batch + frame, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (segs && n > 0) {
        const int m = n < capacity ? n : capacity;
        PLF_HIP_TRY(hipMemcpy(segs, h->d_segs_out + (size_t)frame * h->g.rect_cap, sizeof(float4) * m, hipMemcpyDeviceToHost));
    }
    return PLF_OK;
}
