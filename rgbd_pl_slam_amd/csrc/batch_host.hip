// batch_host.hip -- multi-GPU batch driver: the caller loop of the reference (Examples/RGB-D/rgbd_tum.cc:84-128:
// imread -> SLAM.TrackRGBD -> Tracking::GrabImageRGBD (include/Tracking.h:69) -> Frame ctor -> ExtractORB / ExtractLSD
// -> SearchByProjection) for a batch of INDEPENDENT host frames (BASELINE configs 3-4).
//
// One worker thread per GPU.  A worker owns an ORB handle, a line handle, two matcher handles, four HIP streams (copy-in, ORB,
// lines, copy-out / match) and two pipeline slots of pinned staging + device I/O buffers.  Its block of frames
// (plf_batch_shard: contiguous blocks) is cut into chunks of `frames_in_flight` frames; chunk k is staged + uploaded while the
// kernels of chunk k-1 run and the outputs of chunk k-2 are unpacked.  No collective and no peer access: frames are
// independent, the local map is replicated.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>
#include <pthread.h>
#include "plf_common.h"
#include "orb_geom.h"

// status words of the extractor handles, copied asynchronously on `s` (orb_host.hip / line_host.hip)
int plf_orb_status_async(plf_orb *h, int32_t *host_dst, hipStream_t s);
int plf_line_status_async(plf_line *h, int32_t *host_dst, int32_t *host_flags, int n, hipStream_t s);

namespace {

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Slot {
    uint8_t *h_in = nullptr, *d_in = nullptr, *d_gray = nullptr;
    plf_keypoint *d_kps = nullptr, *h_kps = nullptr;
    uint8_t *d_desc = nullptr, *h_desc = nullptr;
    int32_t *d_nk = nullptr, *h_nk = nullptr;
    plf_keyline *d_lines = nullptr, *h_lines = nullptr;
    uint8_t *d_ldesc = nullptr, *h_ldesc = nullptr;
    double *d_eq = nullptr, *h_eq = nullptr;
    int32_t *d_nl = nullptr, *h_nl = nullptr;
    int32_t *d_mkp = nullptr, *h_mkp = nullptr, *d_nmkp = nullptr, *h_nmkp = nullptr;
    int32_t *d_mln = nullptr, *h_mln = nullptr, *d_nmln = nullptr, *h_nmln = nullptr;
    // RGB-D Frame tail (plf_batch_params.rgbd): depth staging / device images, mvKeysUn, mvuRight, mvDepth and the line-side members
    uint16_t *h_dep = nullptr, *d_dep16 = nullptr;
    float *d_depf = nullptr;
    plf_keypoint *d_kun = nullptr, *h_kun = nullptr;
    float *d_ur = nullptr, *h_ur = nullptr, *d_kd = nullptr, *h_kd = nullptr;
    plf_keyline *d_lun = nullptr, *h_lun = nullptr;
    float *d_le[4] = {nullptr, nullptr, nullptr, nullptr}, *h_le[4] = {nullptr, nullptr, nullptr, nullptr};   // uright start / end, depth start / end
    int32_t *h_status = nullptr;   // [0] ORB status word, [1] line status word
    int32_t *h_trunc = nullptr;    // per frame of the chunk: the line extractor ran out of plf_line_params.max_ms
    hipEvent_t ev_in = nullptr, ev_orb = nullptr, ev_line = nullptr, ev_out = nullptr;
    plf_matcher *mat = nullptr;
    std::vector<plf_frame_view> fviews;
    std::vector<plf_lineframe_view> lviews;
    int64_t first = 0;   // first frame (call-relative) of the chunk in the slot
    int n = 0;           // frames of that chunk; 0 = slot free
    const uint8_t *src = nullptr;
    bool has_depth = false;   // the chunk in the slot came with depth images (mvuRight is valid)
};

struct LocalMap {   // device replica
    plf_mappoint_view pts; plf_mapline_view lns;
    std::vector<void *> owned;
    bool has_pts = false, has_lns = false;
};

struct Job {
    const uint8_t *images; int64_t first, count; int w, h; ptrdiff_t pitch, fstride;
    plf_batch_outputs out;
    bool rgbd = false;
    plf_batch_rgbd R;
};

struct Worker {
    plf_batch *owner = nullptr;
    int device = 0, index = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int cmd = 0;            // 0 idle, 1 extract, 2 upload local map, 3 quit
    bool done = true;
    int rc = PLF_OK;
    Job job;
    // device-side state (touched by the worker thread only)
    plf_orb *orb = nullptr;
    plf_line *line = nullptr;
    hipStream_t s_in = nullptr, s_orb = nullptr, s_line = nullptr, s_out = nullptr;
    Slot slot[2];
    LocalMap map;
    float *d_scale = nullptr;
    int orb_cap = 0, line_cap = 0, nlevels = 0;
    size_t in_bytes_per_frame = 0;
    double t_total = 0, t_stage = 0, t_wait = 0, t_unpack = 0;
    int64_t n_trunc = 0;   // frames of the last call whose line extraction ran out of its time budget
    int numa_node = -1, numa_cpus = 0;   // NUMA node of the GPU (sysfs; -1 unknown) and the CPUs the worker thread was bound to (0: not bound)
};

}  // namespace

struct plf_batch {
    plf_batch_params prm;
    std::vector<int> devices;
    std::vector<Worker *> workers;
    // local-map staging (host pointers of the caller, valid during plf_batch_set_local_map only)
    const plf_mappoint_view *lm_pts = nullptr; const plf_mapline_view *lm_lns = nullptr;
    float th = 3.f, nnratio = 0.8f, bounds[4] = {0, 0, 0, 0};
    bool map_set = false;
};

extern "C" int plf_batch_shard(int64_t n_frames, int32_t parts, int32_t part, int64_t *first, int64_t *count)
{
    if (n_frames < 0 || parts < 1 || part < 0 || part >= parts || !first || !count) return PLF_E_BADARG;
    const int64_t lo = (int64_t)part * n_frames / parts, hi = ((int64_t)part + 1) * n_frames / parts;
    *first = lo; *count = hi - lo;
    return PLF_OK;
}

extern "C" int plf_host_alloc(size_t bytes, void **out)
{
    if (!out) return PLF_E_BADARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return PLF_E_NOMEM; }
    return PLF_OK;
}

extern "C" void plf_host_free(void *p) { if (p) (void)hipHostFree(p); }

namespace {

#define W_TRY(expr)                                                                                        \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            fprintf(stderr, "[plf] batch worker (device %d): %s at %s:%d: %s\n", w->device, hipGetErrorName(_e), \
                    __FILE__, __LINE__, hipGetErrorString(_e));                                            \
            return PLF_E_HIP;                                                                              \
        }                                                                                                  \
    } while (0)
#define W_RC(expr)                      \
    do {                                \
        int _r = (expr);                \
        if (_r != PLF_OK) return _r;    \
    } while (0)

template <class T> static int dev_alloc(Worker *w, T **p, size_t n)
{
    W_TRY(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    return PLF_OK;
}
template <class T> static int pin_alloc(Worker *w, T **p, size_t n)
{
    W_TRY(hipHostMalloc((void **)p, (n ? n : 1) * sizeof(T), hipHostMallocPortable));
    return PLF_OK;
}

// Bind the calling worker thread to the CPUs of its GPU's NUMA node (sysfs: /sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist).
// Called BEFORE the worker allocates its pinned staging slots, so that first touch places them on that node as well: on a two-socket node with eight GPUs the
// PCIe-inclusive rate (~9.4 GB/s of host reads per GPU) would otherwise cross the socket interconnect for half of the workers.  Best effort: any failure leaves
// the thread unbound (numa_cpus = 0).  PLF_BATCH_NO_AFFINITY=1 switches it off.
static void worker_bind_numa(Worker *w)
{
    w->numa_node = -1; w->numa_cpus = 0;
    if (const char *e = getenv("PLF_BATCH_NO_AFFINITY")) { if (atoi(e) != 0) return; }
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, w->device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *fh = fopen(path, "r");
    int node = -1;
    if (fh) { if (fscanf(fh, "%d", &node) != 1) node = -1; fclose(fh); }
    if (node < 0) return;   // (single-socket hosts and containers without the sysfs view report -1)
    w->numa_node = node;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    fh = fopen(path, "r");
    if (!fh) return;
    char list[4096] = {0};
    const size_t got = fread(list, 1, sizeof(list) - 1, fh);
    fclose(fh);
    if (!got) return;
    cpu_set_t want, cur;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return;
    int n = 0;
    char *save = nullptr;   // (strtok_r: every worker thread parses its own list at the same time -- strtok keeps its position in process-global state: ADVICE r04)
    for (char *tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {   // "0-63,128-191"
        int a = -1, b = -1;
        if (sscanf(tok, "%d-%d", &a, &b) == 2) {} else if (sscanf(tok, "%d", &a) == 1) b = a; else continue;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET(c, &cur)) { CPU_SET(c, &want); n++; }   // never beyond what the process may use
    }
    if (n > 0 && pthread_setaffinity_np(pthread_self(), sizeof(want), &want) == 0) w->numa_cpus = n;
}

static int worker_init(Worker *w)
{
    const plf_batch_params &P = w->owner->prm;
    const size_t C = (size_t)P.frames_in_flight;
    W_TRY(hipSetDevice(w->device));
    worker_bind_numa(w);
    int lo = 0, hi = 0;
    W_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    W_TRY(hipStreamCreateWithFlags(&w->s_in, hipStreamNonBlocking));
    W_TRY(hipStreamCreateWithFlags(&w->s_orb, hipStreamNonBlocking));
    // the line extractor holds the latency-bound stage (LSD region growing): highest priority, as in bench.py
    W_TRY(hipStreamCreateWithPriority(&w->s_line, hipStreamNonBlocking, hi));
    W_TRY(hipStreamCreateWithFlags(&w->s_out, hipStreamNonBlocking));
    const int bpp = P.input_format == PLF_FMT_GRAY8 ? 1 : 3;
    const int mw = P.orb.nfeatures > 0 ? P.orb.max_width : P.line.max_width, mh = P.orb.nfeatures > 0 ? P.orb.max_height : P.line.max_height;
    w->in_bytes_per_frame = (size_t)mw * mh * bpp;
    if (P.orb.nfeatures > 0) {
        plf_orb_params op = P.orb;
        op.device = w->device; op.max_batch = (int32_t)C;
        W_RC(plf_orb_create(&op, &w->orb));
        w->orb_cap = plf_orb_capacity(w->orb);
    }
    {   // mvScaleFactors (ORBextractor ctor, include/ORBextractor.h:51-52: float(double(previous) * scaleFactor)): the line matcher reads them too, so the
        // table exists whether or not this worker extracts ORB features (a lines-only batch with map lines silently matched nothing)
        float sc[PLF_MAX_LEVELS];
        if (w->orb) W_RC(plf_orb_get_tables(w->orb, &w->nlevels, sc, nullptr, nullptr, nullptr, nullptr));
        else if (P.max_maplines > 0) {
            const int nl = P.orb.nlevels;
            if (nl < 1 || nl > PLF_MAX_LEVELS || !(P.orb.scale_factor > 1.0f)) {
                fprintf(stderr, "[plf] plf_batch_create: max_maplines > 0 needs orb.scale_factor / orb.nlevels (mvScaleFactors) even with orb.nfeatures <= 0\n");
                return PLF_E_BADARG;
            }
            w->nlevels = nl;
            sc[0] = 1.0f;
            for (int i = 1; i < nl; i++) sc[i] = (float)((double)sc[i - 1] * (double)P.orb.scale_factor);
        }
        if (w->nlevels > 0) {
            W_RC(dev_alloc(w, &w->d_scale, PLF_MAX_LEVELS));
            W_TRY(hipMemcpy(w->d_scale, sc, sizeof(float) * w->nlevels, hipMemcpyHostToDevice));
        }
    }
    if (P.line.nlines > 0) {
        plf_line_params lp = P.line;
        lp.device = w->device; lp.max_batch = (int32_t)C;
        W_RC(plf_line_create(&lp, &w->line));
        w->line_cap = P.line.nlines;
    }
    for (Slot &s : w->slot) {
        W_RC(pin_alloc(w, &s.h_in, C * w->in_bytes_per_frame));
        W_RC(dev_alloc(w, &s.d_in, C * w->in_bytes_per_frame));
        if (bpp == 3) W_RC(dev_alloc(w, &s.d_gray, C * (size_t)mw * mh));
        W_RC(pin_alloc(w, &s.h_status, 4));
        W_RC(pin_alloc(w, &s.h_trunc, C));
        memset(s.h_trunc, 0, C * sizeof(int32_t));
        if (P.rgbd) {
            const size_t px = C * (size_t)mw * mh;
            W_RC(pin_alloc(w, &s.h_dep, px)); W_RC(dev_alloc(w, &s.d_dep16, px)); W_RC(dev_alloc(w, &s.d_depf, px));
            if (w->orb) {
                const size_t K = C * (size_t)w->orb_cap;
                W_RC(dev_alloc(w, &s.d_kun, K)); W_RC(pin_alloc(w, &s.h_kun, K));
                W_RC(dev_alloc(w, &s.d_ur, K)); W_RC(pin_alloc(w, &s.h_ur, K));
                W_RC(dev_alloc(w, &s.d_kd, K)); W_RC(pin_alloc(w, &s.h_kd, K));
            }
            if (w->line) {
                const size_t K = C * (size_t)w->line_cap;
                W_RC(dev_alloc(w, &s.d_lun, K)); W_RC(pin_alloc(w, &s.h_lun, K));
                for (int q = 0; q < 4; q++) { W_RC(dev_alloc(w, &s.d_le[q], K)); W_RC(pin_alloc(w, &s.h_le[q], K)); }
            }
        }
        if (w->orb) {
            const size_t K = C * (size_t)w->orb_cap;
            W_RC(dev_alloc(w, &s.d_kps, K)); W_RC(pin_alloc(w, &s.h_kps, K));
            W_RC(dev_alloc(w, &s.d_desc, K * 32)); W_RC(pin_alloc(w, &s.h_desc, K * 32));
            W_RC(dev_alloc(w, &s.d_nk, C)); W_RC(pin_alloc(w, &s.h_nk, C));
        }
        if (w->line) {
            const size_t K = C * (size_t)w->line_cap;
            W_RC(dev_alloc(w, &s.d_lines, K)); W_RC(pin_alloc(w, &s.h_lines, K));
            W_RC(dev_alloc(w, &s.d_ldesc, K * 32)); W_RC(pin_alloc(w, &s.h_ldesc, K * 32));
            W_RC(dev_alloc(w, &s.d_eq, K * 3)); W_RC(pin_alloc(w, &s.h_eq, K * 3));
            W_RC(dev_alloc(w, &s.d_nl, C)); W_RC(pin_alloc(w, &s.h_nl, C));
        }
        if (P.max_mappoints > 0 || P.max_maplines > 0) {
            W_RC(plf_matcher_create(w->device, w->orb ? w->orb_cap : 1, P.max_mappoints > P.max_maplines ? P.max_mappoints : P.max_maplines,
                                    w->line ? w->line_cap : 1, (int32_t)C, &s.mat));
            if (w->orb) {
                W_RC(dev_alloc(w, &s.d_mkp, C * (size_t)w->orb_cap)); W_RC(pin_alloc(w, &s.h_mkp, C * (size_t)w->orb_cap));
                W_RC(dev_alloc(w, &s.d_nmkp, C)); W_RC(pin_alloc(w, &s.h_nmkp, C));
            }
            if (w->line) {
                W_RC(dev_alloc(w, &s.d_mln, C * (size_t)w->line_cap)); W_RC(pin_alloc(w, &s.h_mln, C * (size_t)w->line_cap));
                W_RC(dev_alloc(w, &s.d_nmln, C)); W_RC(pin_alloc(w, &s.h_nmln, C));
            }
        }
        W_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        W_TRY(hipEventCreateWithFlags(&s.ev_orb, hipEventDisableTiming));
        W_TRY(hipEventCreateWithFlags(&s.ev_line, hipEventDisableTiming));
        W_TRY(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
        s.fviews.resize(C); s.lviews.resize(C);
    }
    W_TRY(hipDeviceSynchronize());
    return PLF_OK;
}

static void worker_free_map(Worker *w)
{
    for (void *p : w->map.owned) (void)hipFree(p);
    w->map = LocalMap();
}

static void worker_shutdown(Worker *w)
{
    (void)hipSetDevice(w->device);
    (void)hipDeviceSynchronize();
    worker_free_map(w);
    for (Slot &s : w->slot) {
        void *dev[] = {s.d_in, s.d_gray, s.d_kps, s.d_desc, s.d_nk, s.d_lines, s.d_ldesc, s.d_eq, s.d_nl, s.d_mkp, s.d_nmkp, s.d_mln, s.d_nmln,
                       s.d_dep16, s.d_depf, s.d_kun, s.d_ur, s.d_kd, s.d_lun, s.d_le[0], s.d_le[1], s.d_le[2], s.d_le[3]};
        for (void *p : dev) if (p) (void)hipFree(p);
        void *pin[] = {s.h_in, s.h_kps, s.h_desc, s.h_nk, s.h_lines, s.h_ldesc, s.h_eq, s.h_nl, s.h_mkp, s.h_nmkp, s.h_mln, s.h_nmln, s.h_status, s.h_trunc,
                       s.h_dep, s.h_kun, s.h_ur, s.h_kd, s.h_lun, s.h_le[0], s.h_le[1], s.h_le[2], s.h_le[3]};
        for (void *p : pin) if (p) (void)hipHostFree(p);
        hipEvent_t ev[] = {s.ev_in, s.ev_orb, s.ev_line, s.ev_out};
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (s.mat) plf_matcher_destroy(s.mat);
    }
    if (w->orb) plf_orb_destroy(w->orb);
    if (w->line) plf_line_destroy(w->line);
    if (w->d_scale) (void)hipFree(w->d_scale);
    hipStream_t st[] = {w->s_in, w->s_orb, w->s_line, w->s_out};
    for (hipStream_t s : st) if (s) (void)hipStreamDestroy(s);
}

template <class T> static int up(Worker *w, const T *host, size_t n, const T **dst)
{
    T *d = nullptr;
    W_TRY(hipMalloc((void **)&d, (n ? n : 1) * sizeof(T)));
    w->map.owned.push_back(d);
    if (n) W_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = d;
    return PLF_OK;
}

static int worker_upload_map(Worker *w)
{
    plf_batch *b = w->owner;
    W_TRY(hipSetDevice(w->device));
    W_TRY(hipDeviceSynchronize());
    worker_free_map(w);
    if (b->lm_pts && b->lm_pts->m > 0 && w->orb && w->slot[0].mat) {
        const plf_mappoint_view &p = *b->lm_pts;
        const size_t m = (size_t)p.m;
        plf_mappoint_view &d = w->map.pts;
        memset(&d, 0, sizeof(d));
        d.m = p.m;
        W_RC(up(w, p.proj_x, m, &d.proj_x)); W_RC(up(w, p.proj_y, m, &d.proj_y)); W_RC(up(w, p.proj_xr, m, &d.proj_xr));
        W_RC(up(w, p.level, m, &d.level)); W_RC(up(w, p.view_cos, m, &d.view_cos)); W_RC(up(w, p.in_view, m, &d.in_view));
        W_RC(up(w, p.desc, m * 32, &d.desc));
        if (p.obs_positive) W_RC(up(w, p.obs_positive, m, &d.obs_positive));
        w->map.has_pts = true;
    }
    if (b->lm_lns && b->lm_lns->m > 0 && w->line && w->slot[0].mat && w->d_scale) {   // (mvScaleFactors come from the ORB extractor)
        const plf_mapline_view &p = *b->lm_lns;
        const size_t m = (size_t)p.m;
        plf_mapline_view &d = w->map.lns;
        memset(&d, 0, sizeof(d));
        d.m = p.m;
        W_RC(up(w, p.x1, m, &d.x1)); W_RC(up(w, p.y1, m, &d.y1)); W_RC(up(w, p.x2, m, &d.x2)); W_RC(up(w, p.y2, m, &d.y2));
        W_RC(up(w, p.level, m, &d.level)); W_RC(up(w, p.view_cos, m, &d.view_cos)); W_RC(up(w, p.in_view, m, &d.in_view));
        W_RC(up(w, p.desc, m * 32, &d.desc));
        w->map.has_lns = true;
    }
    return PLF_OK;
}

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// stage + upload + enqueue the kernels and the output copies of one chunk
// the line half of the Frame tail for the chunk in slot s (lines in s.d_lines / s.d_nl), enqueued on `st`
static int line_tail_submit(Worker *w, Slot &s, const Job &J, int n, hipStream_t st)
{
    return plf_frame_line_tail(s.d_lines, s.d_nl, 0, n, w->line_cap, J.R.depth ? s.d_depf : nullptr, J.w, J.h, &J.R.cam, s.d_lun, s.d_le[0], s.d_le[1],
                               s.d_le[2], s.d_le[3], w->device, st);
}

static int line_tail_download(Worker *w, Slot &s, int n, hipStream_t st)
{
    const size_t K = (size_t)n * w->line_cap;
    W_TRY(hipMemcpyAsync(s.h_lun, s.d_lun, K * sizeof(plf_keyline), hipMemcpyDeviceToHost, st));
    for (int q = 0; q < 4; q++) W_TRY(hipMemcpyAsync(s.h_le[q], s.d_le[q], K * sizeof(float), hipMemcpyDeviceToHost, st));
    return PLF_OK;
}

static int chunk_submit(Worker *w, Slot &s, const Job &J, int64_t first, int n, bool src_pinned, bool depth_pinned)
{
    const plf_batch_params &P = w->owner->prm;
    const int bpp = P.input_format == PLF_FMT_GRAY8 ? 1 : 3;
    const size_t row = (size_t)J.w * bpp, fbytes = row * J.h;
    const uint8_t *src = J.images + (size_t)first * J.fstride;
    s.first = first; s.n = n;
    // slot reuse: its previous upload must have left h_in, its previous kernels must have read d_in, its outputs must have been downloaded
    const bool tight = (size_t)J.pitch == row && (size_t)J.fstride == fbytes;
    const auto t0 = clk::now();
    const uint8_t *up_src = src;
    if (!(src_pinned && tight)) {
        W_TRY(hipEventSynchronize(s.ev_in));
        for (int f = 0; f < n; f++) {
            const uint8_t *fs = src + (size_t)f * J.fstride;
            uint8_t *fd = s.h_in + (size_t)f * fbytes;
            if ((size_t)J.pitch == row) memcpy(fd, fs, fbytes);
            else for (int y = 0; y < J.h; y++) memcpy(fd + (size_t)y * row, fs + (size_t)y * J.pitch, row);
        }
        up_src = s.h_in;
    }
    const uint16_t *up_dep = nullptr;
    const size_t dpx = (size_t)J.w * J.h;
    if (J.rgbd && J.R.depth) {
        const uint16_t *dsrc = J.R.depth + (size_t)first * J.R.depth_frame_stride_elems;
        const bool dtight = J.R.depth_pitch_elems == (ptrdiff_t)J.w && J.R.depth_frame_stride_elems == (ptrdiff_t)dpx;
        up_dep = dsrc;
        if (!(depth_pinned && dtight)) {
            W_TRY(hipEventSynchronize(s.ev_in));
            for (int f = 0; f < n; f++) {
                const uint16_t *fs = dsrc + (size_t)f * J.R.depth_frame_stride_elems;
                uint16_t *fd = s.h_dep + (size_t)f * dpx;
                if (J.R.depth_pitch_elems == (ptrdiff_t)J.w) memcpy(fd, fs, dpx * 2);
                else for (int y = 0; y < J.h; y++) memcpy(fd + (size_t)y * J.w, fs + (size_t)y * J.R.depth_pitch_elems, (size_t)J.w * 2);
            }
            up_dep = s.h_dep;
        }
    }
    w->t_stage += secs(t0, clk::now());
    W_TRY(hipStreamWaitEvent(w->s_in, s.ev_orb, 0));
    W_TRY(hipStreamWaitEvent(w->s_in, s.ev_line, 0));
    W_TRY(hipMemcpyAsync(s.d_in, up_src, (size_t)n * fbytes, hipMemcpyHostToDevice, w->s_in));
    const uint8_t *d_gray = s.d_in;
    if (bpp == 3) {
        W_RC(plf_rgb_to_gray(s.d_in, n, J.w, J.h, (ptrdiff_t)row, (ptrdiff_t)fbytes, P.input_format == PLF_FMT_BGR8, s.d_gray, J.w, (ptrdiff_t)J.w * J.h,
                             w->device, w->s_in));
        d_gray = s.d_gray;
    }
    if (up_dep) {   // imDepth.convertTo(CV_32F, mDepthMapFactor)
        W_TRY(hipMemcpyAsync(s.d_dep16, up_dep, (size_t)n * dpx * 2, hipMemcpyHostToDevice, w->s_in));
        W_RC(plf_depth_to_float(s.d_dep16, n, J.w, J.h, J.w, (ptrdiff_t)dpx, J.R.depth_factor, s.d_depf, w->device, w->s_in));
    }
    W_TRY(hipEventRecord(s.ev_in, w->s_in));
    if (w->line) {
        W_TRY(hipStreamWaitEvent(w->s_line, s.ev_in, 0));
        W_TRY(hipStreamWaitEvent(w->s_line, s.ev_out, 0));
        W_RC(plf_line_extract_batch(w->line, d_gray, PLF_MEM_DEVICE, n, J.w, J.h, J.w, (ptrdiff_t)J.w * J.h, s.d_lines, s.d_ldesc, s.d_eq, s.d_nl,
                                    PLF_MEM_DEVICE, w->line_cap, w->s_line));
        W_RC(plf_line_status_async(w->line, &s.h_status[1], s.h_trunc, n, w->s_line));
        if (J.rgbd) W_RC(line_tail_submit(w, s, J, n, w->s_line));
        W_TRY(hipEventRecord(s.ev_line, w->s_line));
    }
    if (w->orb) {
        W_TRY(hipStreamWaitEvent(w->s_orb, s.ev_in, 0));
        W_TRY(hipStreamWaitEvent(w->s_orb, s.ev_out, 0));
        // ORB starts when the line extractor reaches region growing (a latency-bound chain that leaves room on every CU) instead of competing with its
        // throughput-bound front stages -- the schedule bench.py measures as the better one
        if (w->line) W_RC(plf_line_wait_front(w->line, w->s_orb));
        W_RC(plf_orb_extract_batch(w->orb, d_gray, PLF_MEM_DEVICE, n, J.w, J.h, J.w, (ptrdiff_t)J.w * J.h, s.d_kps, s.d_desc, s.d_nk, PLF_MEM_DEVICE,
                                   w->orb_cap, w->s_orb));
        W_RC(plf_orb_status_async(w->orb, &s.h_status[0], w->s_orb));
        if (J.rgbd)   // Frame::UndistortKeyPoints + Frame::ComputeStereoFromRGBD
            W_RC(plf_frame_tail(s.d_kps, s.d_nk, 0, n, w->orb_cap, up_dep ? s.d_depf : nullptr, J.w, J.h, &J.R.cam, s.d_kun, s.d_ur, s.d_kd, w->device, w->s_orb));
        W_TRY(hipEventRecord(s.ev_orb, w->s_orb));
    }
    s.has_depth = up_dep != nullptr;
    s.src = src;
    return PLF_OK;
}

// matchers + downloads of the chunk in slot s on the output stream.  behind_front: the NEXT chunk's extraction has been enqueued; these light kernels are
// released when its line extractor reaches region growing, i.e. they run in that kernel's shadow instead of colliding with the next front stages
static int chunk_outputs(Worker *w, Slot &s, const Job &J, bool behind_front)
{
    const int n = s.n;
    if (n == 0) return PLF_OK;
    const bool match_pts = w->map.has_pts && s.mat, match_lns = w->map.has_lns && s.mat;
    if (behind_front && w->line) W_RC(plf_line_wait_front(w->line, w->s_out));
    if (w->orb) {
        W_TRY(hipStreamWaitEvent(w->s_out, s.ev_orb, 0));
        const size_t K = (size_t)n * w->orb_cap;
        if (match_pts) {
            W_TRY(hipMemsetAsync(s.d_mkp, 0xFF, K * sizeof(int32_t), w->s_out));
            for (int f = 0; f < n; f++) {
                plf_frame_view &v = s.fviews[f];
                v.n = w->orb_cap; v.n_device = s.d_nk + f; v.keys_un = (J.rgbd ? s.d_kun : s.d_kps) + (size_t)f * w->orb_cap;
                v.uright = (J.rgbd && s.has_depth) ? s.d_ur + (size_t)f * w->orb_cap : nullptr;
                v.desc = s.d_desc + (size_t)f * w->orb_cap * 32;
                v.min_x = w->owner->bounds[0]; v.min_y = w->owner->bounds[1]; v.max_x = w->owner->bounds[2]; v.max_y = w->owner->bounds[3];
                v.scale_factors = w->d_scale; v.nlevels = w->nlevels;
            }
            W_RC(plf_match_project_points(s.mat, s.fviews.data(), n, &w->map.pts, w->owner->th, w->owner->nnratio, s.d_mkp, w->orb_cap, s.d_nmkp, w->s_out));
            W_TRY(hipMemcpyAsync(s.h_mkp, s.d_mkp, K * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
            W_TRY(hipMemcpyAsync(s.h_nmkp, s.d_nmkp, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
        }
        W_TRY(hipMemcpyAsync(s.h_nk, s.d_nk, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
        W_TRY(hipMemcpyAsync(s.h_kps, s.d_kps, K * sizeof(plf_keypoint), hipMemcpyDeviceToHost, w->s_out));
        W_TRY(hipMemcpyAsync(s.h_desc, s.d_desc, K * 32, hipMemcpyDeviceToHost, w->s_out));
        if (J.rgbd) {
            W_TRY(hipMemcpyAsync(s.h_kun, s.d_kun, K * sizeof(plf_keypoint), hipMemcpyDeviceToHost, w->s_out));
            W_TRY(hipMemcpyAsync(s.h_ur, s.d_ur, K * sizeof(float), hipMemcpyDeviceToHost, w->s_out));
            W_TRY(hipMemcpyAsync(s.h_kd, s.d_kd, K * sizeof(float), hipMemcpyDeviceToHost, w->s_out));
        }
    }
    if (w->line) {
        W_TRY(hipStreamWaitEvent(w->s_out, s.ev_line, 0));
        const size_t K = (size_t)n * w->line_cap;
        if (match_lns) {
            W_TRY(hipMemsetAsync(s.d_mln, 0xFF, K * sizeof(int32_t), w->s_out));
            for (int f = 0; f < n; f++) {
                plf_lineframe_view &v = s.lviews[f];
                v.n = w->line_cap; v.n_device = s.d_nl + f; v.lines_un = (J.rgbd ? s.d_lun : s.d_lines) + (size_t)f * w->line_cap;
                v.desc = s.d_ldesc + (size_t)f * w->line_cap * 32;
                v.scale_factors = w->d_scale;
            }
            W_RC(plf_match_project_lines(s.mat, s.lviews.data(), n, &w->map.lns, w->owner->th, w->owner->nnratio, s.d_mln, w->line_cap, s.d_nmln, w->s_out));
            W_TRY(hipMemcpyAsync(s.h_mln, s.d_mln, K * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
            W_TRY(hipMemcpyAsync(s.h_nmln, s.d_nmln, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
        }
        W_TRY(hipMemcpyAsync(s.h_nl, s.d_nl, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
        W_TRY(hipMemcpyAsync(s.h_lines, s.d_lines, K * sizeof(plf_keyline), hipMemcpyDeviceToHost, w->s_out));
        W_TRY(hipMemcpyAsync(s.h_ldesc, s.d_ldesc, K * 32, hipMemcpyDeviceToHost, w->s_out));
        W_TRY(hipMemcpyAsync(s.h_eq, s.d_eq, K * 3 * sizeof(double), hipMemcpyDeviceToHost, w->s_out));
        if (J.rgbd) W_RC(line_tail_download(w, s, n, w->s_out));
    }
    W_TRY(hipEventRecord(s.ev_out, w->s_out));
    return PLF_OK;
}

// wait for a chunk and scatter its outputs into the caller's arrays; *soft collects PLF_E_CAPACITY
static int chunk_retire(Worker *w, Slot &s, const Job &J, int *soft)
{
    if (s.n == 0) return PLF_OK;
    const plf_batch_outputs &O = J.out;
    auto t0 = clk::now();
    W_TRY(hipEventSynchronize(s.ev_out));
    auto t1 = clk::now();
    w->t_wait += secs(t0, t1);
    const bool match_pts = w->map.has_pts && s.mat, match_lns = w->map.has_lns && s.mat;
    if (w->orb) {
        if (s.h_status[0] & 5) return PLF_E_HIP;   // internal pool / selection overflow: excluded by the sizing rules
        for (int f = 0; f < s.n; f++) {
            const size_t g = (size_t)(s.first + f);
            int n = s.h_nk[f];
            if (n > w->orb_cap) n = w->orb_cap;
            if (n > O.kp_capacity) { n = O.kp_capacity; *soft = PLF_E_CAPACITY; }
            O.n_kps[g] = n;
            memcpy(O.kps + g * O.kp_capacity, s.h_kps + (size_t)f * w->orb_cap, sizeof(plf_keypoint) * n);
            memcpy(O.desc + g * O.kp_capacity * 32, s.h_desc + (size_t)f * w->orb_cap * 32, (size_t)32 * n);
            if (O.match_of_kp) {
                if (match_pts) memcpy(O.match_of_kp + g * O.kp_capacity, s.h_mkp + (size_t)f * w->orb_cap, sizeof(int32_t) * n);
                else for (int i = 0; i < n; i++) O.match_of_kp[g * O.kp_capacity + i] = -1;
            }
            if (O.n_kp_matches) {
                int nm = match_pts ? s.h_nmkp[f] : 0;
                if (match_pts && n < s.h_nk[f]) {     // the caller's capacity cut key points: count what match_of_kp shows
                    nm = 0;
                    for (int i = 0; i < n; i++) nm += s.h_mkp[(size_t)f * w->orb_cap + i] >= 0;
                }
                O.n_kp_matches[g] = nm;
            }
            if (J.rgbd) {
                const plf_batch_rgbd &R = J.R;
                if (R.kps_un) memcpy(R.kps_un + g * O.kp_capacity, s.h_kun + (size_t)f * w->orb_cap, sizeof(plf_keypoint) * n);
                if (R.uright) memcpy(R.uright + g * O.kp_capacity, s.h_ur + (size_t)f * w->orb_cap, sizeof(float) * n);
                if (R.kp_depth) memcpy(R.kp_depth + g * O.kp_capacity, s.h_kd + (size_t)f * w->orb_cap, sizeof(float) * n);
            }
        }
    }
    if (w->line) {
        if (s.h_status[1] & 4) return PLF_E_HIP;
        if (s.h_status[1] & 1) {
            // pooled NFA buffers exceeded by this chunk (pathological textures): redo it through the host-memory entry point, which
            // halves the batch until it fits (line_host.hip).  Stream-ordered after everything queued on the line stream.
            const int bpp = w->owner->prm.input_format == PLF_FMT_GRAY8 ? 1 : 3;
            W_TRY(hipStreamSynchronize(w->s_line));
            const uint8_t *d_gray = bpp == 3 ? s.d_gray : s.d_in;   // still intact: the slot is not reused before it is retired
            std::vector<int32_t> nl(s.n);
            int rc = plf_line_extract_batch(w->line, d_gray, PLF_MEM_DEVICE, s.n, J.w, J.h, J.w, (ptrdiff_t)J.w * J.h, s.h_lines, s.h_ldesc, s.h_eq, nl.data(),
                                            PLF_MEM_HOST, w->line_cap, w->s_line);
            if (rc < 0 && rc != PLF_E_CAPACITY) return rc;   // (> 0: warnings -- PLF_W_SLOW: the redo of a pathological chunk IS slow; outputs complete)
            if (rc == PLF_E_CAPACITY) *soft = PLF_E_CAPACITY;
            for (int f = 0; f < s.n; f++) s.h_nl[f] = nl[f];
            // the time-budget flags of the failed pass are meaningless as well: those of the redo (collected over its pieces by the line extractor)
            s.h_status[1] &= ~8;
            // (the flags are read whatever plf_line_last_status reports: it returns PLF_E_CAPACITY before it looks at the truncation bit, so a redo that clipped lines to
            // capacity AND ran out of max_ms would otherwise be counted as complete: ADVICE r04)
            if (w->owner->prm.line.max_ms > 0.f && plf_line_truncated(w->line, s.h_trunc, s.n) == PLF_OK) {
                for (int f = 0; f < s.n; f++) if (s.h_trunc[f]) { s.h_status[1] |= 8; break; }
            }
            if (match_lns || J.rgbd) {   // the Frame tail / matches of the failed pass are meaningless: redo them on the fresh lines
                const size_t K = (size_t)s.n * w->line_cap;
                W_TRY(hipMemcpyAsync(s.d_lines, s.h_lines, K * sizeof(plf_keyline), hipMemcpyHostToDevice, w->s_out));
                W_TRY(hipMemcpyAsync(s.d_ldesc, s.h_ldesc, K * 32, hipMemcpyHostToDevice, w->s_out));
                W_TRY(hipMemcpyAsync(s.d_nl, s.h_nl, (size_t)s.n * sizeof(int32_t), hipMemcpyHostToDevice, w->s_out));
                if (J.rgbd) { W_RC(line_tail_submit(w, s, J, s.n, w->s_out)); W_RC(line_tail_download(w, s, s.n, w->s_out)); }
            }
            if (match_lns) {
                const size_t K = (size_t)s.n * w->line_cap;
                W_TRY(hipMemsetAsync(s.d_mln, 0xFF, K * sizeof(int32_t), w->s_out));
                W_RC(plf_match_project_lines(s.mat, s.lviews.data(), s.n, &w->map.lns, w->owner->th, w->owner->nnratio, s.d_mln, w->line_cap, s.d_nmln, w->s_out));
                W_TRY(hipMemcpyAsync(s.h_mln, s.d_mln, K * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
                W_TRY(hipMemcpyAsync(s.h_nmln, s.d_nmln, (size_t)s.n * sizeof(int32_t), hipMemcpyDeviceToHost, w->s_out));
            }
            W_TRY(hipStreamSynchronize(w->s_out));
        }
        for (int f = 0; f < s.n; f++) {
            const size_t g = (size_t)(s.first + f);
            int n = s.h_nl[f];
            if (n > w->line_cap) n = w->line_cap;
            if (n > O.line_capacity) { n = O.line_capacity; *soft = PLF_E_CAPACITY; }
            O.n_lines[g] = n;
            memcpy(O.lines + g * O.line_capacity, s.h_lines + (size_t)f * w->line_cap, sizeof(plf_keyline) * n);
            memcpy(O.ldesc + g * O.line_capacity * 32, s.h_ldesc + (size_t)f * w->line_cap * 32, (size_t)32 * n);
            if (O.line_eq) memcpy(O.line_eq + g * O.line_capacity * 3, s.h_eq + (size_t)f * w->line_cap * 3, sizeof(double) * 3 * n);
            if (O.match_of_line) {
                if (match_lns) memcpy(O.match_of_line + g * O.line_capacity, s.h_mln + (size_t)f * w->line_cap, sizeof(int32_t) * n);
                else for (int i = 0; i < n; i++) O.match_of_line[g * O.line_capacity + i] = -1;
            }
            if (O.n_line_matches) {
                int nm = match_lns ? s.h_nmln[f] : 0;
                if (match_lns && n < s.h_nl[f]) {
                    nm = 0;
                    for (int i = 0; i < n; i++) nm += s.h_mln[(size_t)f * w->line_cap + i] >= 0;
                }
                O.n_line_matches[g] = nm;
            }
            if ((s.h_status[1] & 8) && s.h_trunc[f]) w->n_trunc++;
            if (J.rgbd) {
                const plf_batch_rgbd &R = J.R;
                float *dst[4] = {R.uright_start, R.uright_end, R.depth_start, R.depth_end};
                if (R.lines_un) memcpy(R.lines_un + g * O.line_capacity, s.h_lun + (size_t)f * w->line_cap, sizeof(plf_keyline) * n);
                for (int q = 0; q < 4; q++)
                    if (dst[q]) memcpy(dst[q] + g * O.line_capacity, s.h_le[q] + (size_t)f * w->line_cap, sizeof(float) * n);
            }
        }
    }
    w->t_unpack += secs(t1, clk::now());
    s.n = 0;
    return PLF_OK;
}

static int worker_extract(Worker *w)
{
    const Job &J = w->job;
    w->t_total = w->t_stage = w->t_wait = w->t_unpack = 0;
    w->n_trunc = 0;
    if (J.count <= 0) return PLF_OK;
    const auto t0 = clk::now();
    W_TRY(hipSetDevice(w->device));
    const int C = w->owner->prm.frames_in_flight;
    const bool pinned = is_pinned(J.images + (size_t)J.first * J.fstride);
    const bool dpinned = J.rgbd && J.R.depth && is_pinned(J.R.depth + (size_t)J.first * J.R.depth_frame_stride_elems);
    int soft = PLF_OK, k = 0;
    Slot *pending = nullptr;   // extraction enqueued, matchers / downloads not yet
    for (int64_t done = 0; done < J.count; done += C, k++) {
        Slot &s = w->slot[k & 1];
        int rc = chunk_retire(w, s, J, &soft);   // chunk k-2 (normally retired already)
        if (rc != PLF_OK) return rc;
        const int n = (int)(J.count - done < C ? J.count - done : C);
        rc = chunk_submit(w, s, J, J.first + done, n, pinned, dpinned);
        if (rc != PLF_OK) return rc;
        if (pending) {
            rc = chunk_outputs(w, *pending, J, true);   // chunk k-1, behind the front stages of chunk k
            if (rc != PLF_OK) return rc;
        }
        pending = &s;
        rc = chunk_retire(w, w->slot[(k & 1) ^ 1], J, &soft);   // chunk k-1 while chunk k runs
        if (rc != PLF_OK) return rc;
    }
    if (pending) {
        const int rc = chunk_outputs(w, *pending, J, false);
        if (rc != PLF_OK) return rc;
    }
    for (int i = 0; i < 2; i++) {
        int rc = chunk_retire(w, w->slot[(k + i) & 1], J, &soft);
        if (rc != PLF_OK) return rc;
    }
    w->t_total = secs(t0, clk::now());
    return soft;
}

static void worker_main(Worker *w)
{
    int rc = worker_init(w);
    {
        std::lock_guard<std::mutex> g(w->mu);
        w->rc = rc; w->done = true;
    }
    w->cv.notify_all();
    for (;;) {
        int cmd;
        {
            std::unique_lock<std::mutex> g(w->mu);
            w->cv.wait(g, [w] { return w->cmd != 0; });
            cmd = w->cmd;
        }
        int r = PLF_OK;
        if (cmd == 1) r = worker_extract(w);
        else if (cmd == 2) r = worker_upload_map(w);
        if (cmd == 1 && r != PLF_OK && r != PLF_E_CAPACITY) {   // leave the device quiescent and the slots free after a failure
            (void)hipDeviceSynchronize();
            for (Slot &s : w->slot) s.n = 0;
        }
        if (cmd == 3) worker_shutdown(w);
        {
            std::lock_guard<std::mutex> g(w->mu);
            w->rc = r; w->cmd = 0; w->done = true;
        }
        w->cv.notify_all();
        if (cmd == 3) return;
    }
}

static void post(Worker *w, int cmd)
{
    {
        std::lock_guard<std::mutex> g(w->mu);
        w->cmd = cmd; w->done = false;
    }
    w->cv.notify_all();
}

static int wait_done(Worker *w)
{
    std::unique_lock<std::mutex> g(w->mu);
    w->cv.wait(g, [w] { return w->done; });
    return w->rc;
}

static int run_all(plf_batch *b, int cmd)
{
    for (Worker *w : b->workers) post(w, cmd);
    int hard = PLF_OK, soft = PLF_OK;
    for (Worker *w : b->workers) {
        const int r = wait_done(w);
        if (r == PLF_E_CAPACITY) soft = r;
        else if (r != PLF_OK && hard == PLF_OK) hard = r;
    }
    return hard != PLF_OK ? hard : soft;
}

}  // namespace

extern "C" int plf_batch_create(const plf_batch_params *p, plf_batch **out)
{
    if (!p || !out) return PLF_E_BADARG;
    *out = nullptr;
    if (p->frames_in_flight < 1 || p->n_devices < 0 || (p->orb.nfeatures <= 0 && p->line.nlines <= 0) || p->input_format < PLF_FMT_GRAY8 ||
        p->input_format > PLF_FMT_BGR8 || p->max_mappoints < 0 || p->max_maplines < 0)
        return PLF_E_BADARG;
    if (p->orb.nfeatures > 0 && p->line.nlines > 0 && (p->orb.max_width != p->line.max_width || p->orb.max_height != p->line.max_height)) return PLF_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        fprintf(stderr, "[plf] no HIP device available: the batch driver has no CPU path\n");
        return PLF_E_HIP;
    }
    plf_batch *b = new (std::nothrow) plf_batch();
    if (!b) return PLF_E_NOMEM;
    b->prm = *p;
    const int n = p->n_devices > 0 ? p->n_devices : ndev;
    for (int i = 0; i < n; i++) {
        const int d = (p->n_devices > 0 && p->devices) ? p->devices[i] : i;
        if (d < 0 || d >= ndev) { delete b; return PLF_E_BADARG; }
        b->devices.push_back(d);
    }
    b->prm.devices = nullptr;
    for (int i = 0; i < n; i++) {
        Worker *w = new (std::nothrow) Worker();
        if (!w) { plf_batch_destroy(b); return PLF_E_NOMEM; }
        w->owner = b; w->device = b->devices[i]; w->index = i; w->done = false;
        b->workers.push_back(w);
        w->th = std::thread(worker_main, w);
    }
    int rc = PLF_OK;
    for (Worker *w : b->workers) {
        const int r = wait_done(w);
        if (r != PLF_OK && rc == PLF_OK) rc = r;
    }
    if (rc != PLF_OK) { plf_batch_destroy(b); return rc; }
    *out = b;
    return PLF_OK;
}

extern "C" void plf_batch_destroy(plf_batch *b)
{
    if (!b) return;
    for (Worker *w : b->workers) {
        post(w, 3);
        (void)wait_done(w);
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    delete b;
}

extern "C" int plf_batch_device_count(const plf_batch *b) { return b ? (int)b->workers.size() : PLF_E_BADARG; }
extern "C" int plf_batch_device(const plf_batch *b, int32_t i) { return (b && i >= 0 && i < (int)b->workers.size()) ? b->devices[i] : PLF_E_BADARG; }

extern "C" int plf_batch_set_local_map(plf_batch *b, const plf_mappoint_view *points, const plf_mapline_view *lines, float th, float nnratio,
                                       float min_x, float min_y, float max_x, float max_y)
{
    if (!b) return PLF_E_BADARG;
    if (points && (points->m < 0 || points->m > b->prm.max_mappoints)) return PLF_E_BADARG;
    if (lines && (lines->m < 0 || lines->m > b->prm.max_maplines)) return PLF_E_BADARG;
    if (points && points->m > 0 && (!points->proj_x || !points->proj_y || !points->proj_xr || !points->level || !points->view_cos || !points->in_view || !points->desc))
        return PLF_E_BADARG;
    if (lines && lines->m > 0 && (!lines->x1 || !lines->y1 || !lines->x2 || !lines->y2 || !lines->level || !lines->view_cos || !lines->in_view || !lines->desc))
        return PLF_E_BADARG;
    if (points && points->m > 0 && (!(max_x > min_x) || !(max_y > min_y))) return PLF_E_BADARG;
    b->lm_pts = points; b->lm_lns = lines; b->th = th; b->nnratio = nnratio;
    b->bounds[0] = min_x; b->bounds[1] = min_y; b->bounds[2] = max_x; b->bounds[3] = max_y;
    const int rc = run_all(b, 2);
    b->lm_pts = nullptr; b->lm_lns = nullptr;
    return rc;
}

extern "C" int plf_batch_extract(plf_batch *b, const uint8_t *images, int64_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch,
                                 ptrdiff_t frame_stride, const plf_batch_outputs *out)
{
    return plf_batch_extract_rgbd(b, images, n_frames, width, height, pitch, frame_stride, out, nullptr);
}

extern "C" int plf_batch_extract_rgbd(plf_batch *b, const uint8_t *images, int64_t n_frames, int32_t width, int32_t height, ptrdiff_t pitch,
                                      ptrdiff_t frame_stride, const plf_batch_outputs *out, const plf_batch_rgbd *rgbd)
{
    if (!b || !out) return PLF_E_BADARG;
    if (rgbd) {
        if (!b->prm.rgbd) return PLF_E_BADARG;
        if (rgbd->depth && (rgbd->depth_pitch_elems < (ptrdiff_t)width ||
                            rgbd->depth_frame_stride_elems < rgbd->depth_pitch_elems * (ptrdiff_t)(height - 1) + (ptrdiff_t)width))
            return PLF_E_BADARG;
    }
    if (!images || n_frames <= 0 || width <= 0 || height <= 0) return PLF_E_EMPTY;   // reference: silent return on an empty image, so@0x76dda
    const plf_batch_params &P = b->prm;
    const int bpp = P.input_format == PLF_FMT_GRAY8 ? 1 : 3;
    const bool orb = P.orb.nfeatures > 0, line = P.line.nlines > 0;
    const int mw = orb ? P.orb.max_width : P.line.max_width, mh = orb ? P.orb.max_height : P.line.max_height;
    if (width > mw || height > mh || pitch < (ptrdiff_t)width * bpp || frame_stride < pitch * (ptrdiff_t)(height - 1) + (ptrdiff_t)width * bpp) return PLF_E_BADARG;
    if (orb && (!out->kps || !out->desc || !out->n_kps || out->kp_capacity < 1)) return PLF_E_BADARG;
    if (line && (!out->lines || !out->ldesc || !out->n_lines || out->line_capacity < 1)) return PLF_E_BADARG;
    const int nw = (int)b->workers.size();
    for (int i = 0; i < nw; i++) {
        Job &J = b->workers[i]->job;
        J.images = images; J.w = width; J.h = height; J.pitch = pitch; J.fstride = frame_stride; J.out = *out;
        J.rgbd = rgbd != nullptr;
        if (rgbd) J.R = *rgbd; else memset(&J.R, 0, sizeof(J.R));
        (void)plf_batch_shard(n_frames, nw, i, &J.first, &J.count);
    }
    return run_all(b, 1);
}

// frames of the last plf_batch_extract* call whose LSD region growing ran out of plf_line_params.max_ms (0 without a budget)
extern "C" int64_t plf_batch_truncated_frames(const plf_batch *b)
{
    if (!b) return 0;   // (a count, not a status: no handle, no frames)
    int64_t n = 0;
    for (const Worker *w : b->workers) n += w->n_trunc;
    return n;
}

extern "C" int plf_batch_worker_affinity(const plf_batch *b, int32_t worker, int32_t *numa_node, int32_t *n_cpus)
{
    if (!b || worker < 0 || worker >= (int32_t)b->workers.size()) return PLF_E_BADARG;
    if (numa_node) *numa_node = b->workers[worker]->numa_node;
    if (n_cpus) *n_cpus = b->workers[worker]->numa_cpus;
    return PLF_OK;
}

extern "C" int plf_batch_worker_timing(const plf_batch *b, int32_t worker, double *out4)
{
    if (!b || !out4 || worker < 0 || worker >= (int32_t)b->workers.size()) return PLF_E_BADARG;
    const Worker *w = b->workers[worker];
    out4[0] = w->t_total; out4[1] = w->t_stage; out4[2] = w->t_wait; out4[3] = w->t_unpack;
    return PLF_OK;
}

extern "C" int plf_batch_last_timing(const plf_batch *b, double *out4)
{
    if (!b || !out4) return PLF_E_BADARG;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (const Worker *w : b->workers) {
        if (w->t_total > out4[0]) out4[0] = w->t_total;
        if (w->t_stage > out4[1]) out4[1] = w->t_stage;
        if (w->t_wait > out4[2]) out4[2] = w->t_wait;
        if (w->t_unpack > out4[3]) out4[3] = w->t_unpack;
    }
    return PLF_OK;
}
