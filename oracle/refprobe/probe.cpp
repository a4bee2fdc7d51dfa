// probe.cpp -- executes pure functions of the PREBUILT reference binary
// (/root/reference/lib/libORB_SLAM2.so) to produce golden fixtures for tests/golden/.
//
// TEST TOOLING.  Runs only in the build container; nothing here ships or is imported by the
// product.  No reference source is compiled (none exists for this path); the reference's own
// machine code is called through look-alike declarations:
//   tier A  (no OpenCV code involved): ORBextractor ctor tables, DistributeOctTree/DivideNode,
//           computeOrientation integer moments (cv::fastAtan2 replaced by a recorder),
//           ORBmatcher::DescriptorDistance, RadiusByViewingCos, ComputeThreeMaxima, constants.
//   tier B  ("glue"): ORBextractor::ComputeKeyPointsOctTree driven on a hand-laid pyramid with
//           cv::FAST / cv::fastAtan2 / cv::Mat(ROI) supplied by THIS repo's restated primitives
//           (oracle/orb_oracle.c).  Pins cell tiling, retry rule, octree, border offsets, size
//           truncation, orientation and output order against the reference's machine code; it does
//           NOT pin the OpenCV primitives themselves (they are ours on both sides).
//
// A bump `operator new` makes node addresses monotonic so that DistributeOctTree's
// pair<int,ExtractorNode*> tie-break equals creation order (Appendix B determinism rule).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <climits>
#include <new>
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>
#include <dlfcn.h>
#include <execinfo.h>
#include <link.h>

#include "../oracle.h"

// ---------------------------------------------------------------- bump allocator (interposes the .so's operator new)
static char *g_arena = nullptr;
static size_t g_arena_off = 0, g_arena_cap = 0;
static void *bump(size_t n)
{
    if (!g_arena) { g_arena_cap = (size_t)3 << 30; g_arena = (char *)malloc(g_arena_cap); }
    n = (n + 15) & ~(size_t)15;
    if (g_arena_off + n > g_arena_cap) { fprintf(stderr, "arena exhausted: request %zu at offset %zu\n", n, g_arena_off); void *bt[16]; int k = backtrace(bt, 16); backtrace_symbols_fd(bt, k, 2); abort(); }
    void *p = g_arena + g_arena_off;
    g_arena_off += n;
    return p;
}
void *operator new(size_t n) { return bump(n); }
void *operator new[](size_t n) { return bump(n); }
void operator delete(void *) noexcept {}
void operator delete[](void *) noexcept {}
void operator delete(void *, size_t) noexcept {}
void operator delete[](void *, size_t) noexcept {}
static void arena_reset() {}   // (no reuse: long-lived std::vectors of this file also live in the arena)

// ---------------------------------------------------------------- look-alike OpenCV 3.3 PODs
namespace cv {
struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct Range { int start, end; };
// cv::Size_ of 3.3 has a user-provided copy constructor: by-value arguments travel by reference (Itanium ABI)
template <class T> struct Size_ { T width, height; Size_() : width(0), height(0) {} Size_(const Size_ &s) : width(s.width), height(s.height) {} };
template <class T> struct Rect_ { T x, y, width, height; };
template <class T> struct Scalar_ { T v[4]; };
struct _InputArray;
struct _OutputArray;
struct MatExpr;
struct Mat {  // OpenCV 3.3 layout, 96 bytes
    int flags, dims, rows, cols;
    unsigned char *data;
    const unsigned char *datastart, *dataend, *datalimit;
    void *allocator, *u;
    int *size_p;
    size_t *step_p;
    size_t step_buf[2];
    Mat() {}
    Mat(const Mat &m) { memcpy((void *)this, &m, sizeof(Mat)); size_p = &rows; step_p = step_buf; u = nullptr; }
    Mat &operator=(const Mat &m) { memcpy((void *)this, &m, sizeof(Mat)); size_p = &rows; step_p = step_buf; u = nullptr; return *this; }
    ~Mat();
    Mat(const Mat &m, const Range &rowRange, const Range &colRange);
    Mat(const Mat &m, const Rect_<int> &roi);
    void deallocate();
    void create(int ndims, const int *sizes, int type);
    void copyTo(const _OutputArray &dst) const;
    void copySize(const Mat &m);
    static MatExpr zeros(int rows, int cols, int type);
    MatExpr t() const;
    double dot(const _InputArray &m) const;
    Mat reshape(int cn, int rows = 0) const;
};
struct _InputArray {
    int flags; void *obj; int sz_w, sz_h;
    int kind() const;
    bool empty() const;
    Mat getMat_(int idx) const;
};
struct _OutputArray : _InputArray {
    void create(int rows, int cols, int type, int i, bool allowTransposed, int fixedDepthMask) const;
    void release() const;
};
struct MatOp {  // vtable: [0],[1] destructors, [2] elementWise, [3] assign  (the binary calls slot 3 after Mat::zeros)
    virtual ~MatOp() {}
    virtual bool elementWise(const MatExpr &) const { return false; }
    virtual void assign(const MatExpr &, Mat &m, int type) const;
};
struct MatExpr { const MatOp *op; int flags; Mat a, b, c; double alpha, beta; Scalar_<double> s; MatExpr() {} ~MatExpr(); };
MatExpr operator*(const Mat &a, const Mat &b);
MatExpr operator*(const MatExpr &e, const Mat &b);
MatExpr operator+(const MatExpr &e, const Mat &b);
MatExpr operator-(const MatExpr &e);
MatExpr operator-(const Mat &a, const Mat &b);
MatExpr operator/(const Mat &a, double s);
MatExpr operator*(double s, const Mat &a);
MatExpr operator*(double s, const MatExpr &e);
MatExpr operator-(const Mat &a);
double norm(const _InputArray &src, int normType, const _InputArray &mask);
const _InputArray &noArray();
void resize(const _InputArray &, const _OutputArray &, Size_<int>, double, double, int);
void copyMakeBorder(const _InputArray &, const _OutputArray &, int, int, int, int, int, const Scalar_<double> &);
void GaussianBlur(const _InputArray &, const _OutputArray &, Size_<int>, double, double, int);
void undistortPoints(const _InputArray &src, const _OutputArray &dst, const _InputArray &cameraMatrix, const _InputArray &distCoeffs, const _InputArray &R, const _InputArray &P);
float fastAtan2(float y, float x);
void fastFree(void *);
void FAST(const _InputArray &, std::vector<KeyPoint> &, int, bool);
}  // namespace cv
static_assert(sizeof(cv::Mat) == 96, "cv::Mat 3.3 layout");
static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
static_assert(sizeof(cv::MatExpr) == 352, "cv::MatExpr 3.3 layout");

static void mat_init(cv::Mat *m, unsigned char *data, int rows, int cols, size_t pitch)
{
    memset(m, 0, sizeof(*m));
    m->flags = 0x42FF0000;  // MAGIC_VAL | CV_8UC1 (continuity flag not set)
    m->dims = 2; m->rows = rows; m->cols = cols; m->data = data;
    m->datastart = data; m->dataend = m->datalimit = data + pitch * rows;
    m->size_p = &m->rows; m->step_p = m->step_buf;
    m->step_buf[0] = pitch; m->step_buf[1] = 1;
}

// ---- substitutes for the few OpenCV entry points the executed reference code reaches
static int g_mode = 0;  // 0: record fastAtan2 args; 1: real polynomial
static std::vector<float> g_atan_rec;
struct FastCall { const unsigned char *p; int w, h, th, nonmax; };
static std::vector<FastCall> g_fast_calls;
static int g_fast_mode = 0;  // 0: record + synthetic (force empty at ini threshold optional), 1: real
static int g_force_retry = 0;

float cv::fastAtan2(float y, float x)
{
    if (g_mode == 0) { g_atan_rec.push_back(y); g_atan_rec.push_back(x); return 0.f; }
    return orc_fast_atan2(y, x);
}
void cv::fastFree(void *) {}
void cv::Mat::deallocate() {}
cv::Mat::Mat(const Mat &m, const Range &rr, const Range &cr)
{
    memcpy((void *)this, &m, sizeof(Mat));
    size_p = &rows; step_p = step_buf; u = nullptr;
    if (!(rr.start == INT_MIN && rr.end == INT_MAX)) { data += step_buf[0] * rr.start; rows = rr.end - rr.start; }
    if (!(cr.start == INT_MIN && cr.end == INT_MAX)) { data += (size_t)cr.start * ((flags & 0xFFF) == 5 ? 4 : 1); cols = cr.end - cr.start; }
}
void cv::FAST(const _InputArray &arr, std::vector<KeyPoint> &kps, int th, bool nonmax)
{
    const Mat *m = (const Mat *)arr.obj;
    FastCall c;
    c.p = m->data;  // decoded to (level, x, y) by the caller, which knows the planes
    c.w = m->cols; c.h = m->rows; c.th = th; c.nonmax = nonmax;
    g_fast_calls.push_back(c);
    kps.clear();
    if (g_fast_mode == 0) {
        if (g_force_retry && th > 10) return;  // pretend the ini-threshold call found nothing
        KeyPoint k = {3.f, 3.f, 7.f, -1.f, 30.f, 0, -1};
        kps.push_back(k);
        return;
    }
    std::vector<orc_keypoint> tmp(4096);
    int n = orc_fast9_16(m->data, m->cols, m->rows, (ptrdiff_t)m->step_buf[0], th, tmp.data(), 4096);
    for (int i = 0; i < n && i < 4096; i++) {
        KeyPoint k = {tmp[i].x, tmp[i].y, tmp[i].size, tmp[i].angle, tmp[i].response, tmp[i].octave, tmp[i].class_id};
        kps.push_back(k);
    }
}

// ---- tier C: the OpenCV 3.3 entry points ORBextractor::operator() / ComputePyramid reach (so@0x76da0, so@0x70430).
// Headers and ownership only (u = NULL: nothing is reference counted, memory comes from the bump arena); the pixel
// work is done by oracle/orb_oracle.c.
static void mat_empty(cv::Mat *m)
{
    memset((void *)m, 0, sizeof(*m));
    m->flags = 0x42FF0000; m->size_p = &m->rows; m->step_p = m->step_buf;
}
static void mat_alloc(cv::Mat *m, int rows, int cols, int type)
{
    if (type != 0 && type != 5) { fprintf(stderr, "refprobe: unexpected Mat type %d\n", type); abort(); }
    const size_t esz = type == 5 ? 4 : 1;   // CV_8UC1 / CV_32FC1
    if (m->data && m->dims == 2 && m->rows == rows && m->cols == cols && (m->flags & 0xFFF) == type) return;   // Mat::create keeps a matching buffer
    unsigned char *d = (unsigned char *)bump((size_t)rows * cols * esz + 64);
    mat_init(m, d, rows, cols, (size_t)cols * esz);
    m->step_buf[1] = esz;
    m->flags = 0x42FF0000 | 0x4000 | type;  // MAGIC | CONTINUOUS | type
}
cv::Mat::~Mat() {}
cv::Mat::Mat(const Mat &m, const Rect_<int> &roi)
{
    memcpy((void *)this, &m, sizeof(Mat));
    size_p = &rows; step_p = step_buf; u = nullptr;
    data += step_buf[0] * roi.y + roi.x; rows = roi.height; cols = roi.width;
    if (roi.width < m.cols) flags &= ~0x4000;
}
void cv::Mat::create(int ndims, const int *sizes, int type)
{
    if (ndims != 2) { fprintf(stderr, "refprobe: Mat::create ndims %d\n", ndims); abort(); }
    mat_alloc(this, sizes[0], sizes[1], type & 0xFFF);
}
void cv::Mat::copySize(const Mat &m) { dims = m.dims; rows = m.rows; cols = m.cols; step_buf[0] = m.step_buf[0]; step_buf[1] = m.step_buf[1]; }
void cv::Mat::copyTo(const _OutputArray &dst) const
{
    Mat *d = (Mat *)dst.obj;
    mat_alloc(d, rows, cols, flags & 0xFFF);
    for (int y = 0; y < rows; y++) memcpy(d->data + d->step_buf[0] * y, data + step_buf[0] * y, (size_t)cols * ((flags & 0xFFF) == 5 ? 4 : 1));
}
void cv::MatOp::assign(const MatExpr &, Mat &m, int) const   // the only expression the path builds: Mat::zeros
{
    for (int y = 0; y < m.rows; y++) memset(m.data + m.step_buf[0] * y, 0, (size_t)m.cols);
}
static cv::MatOp g_zero_op;
static int g_zeros_rows = 0;
cv::MatExpr cv::Mat::zeros(int rows, int cols, int type)
{
    MatExpr e;
    e.op = &g_zero_op; e.flags = 0; e.alpha = 0; e.beta = 0; memset(&e.s, 0, sizeof(e.s));
    mat_empty(&e.a); mat_empty(&e.b); mat_empty(&e.c);
    g_zeros_rows = rows; (void)cols; (void)type;
    return e;
}
// ---- the little matrix algebra of SearchByProjection(CurrentFrame, LastFrame): -Rcw.t()*tcw, Rlw*twc+tlw, Rcw*x3Dw+tcw
// (3x3 and 3x1 CV_32F).  Arithmetic as cv::gemm of OpenCV 3.3 does it (matmul.cpp): A*B(+C) with flags == 0 takes the
// small-matrix float path d = float(double(t)*alpha + beta*double(c)), t = a0*b0 + a1*b1 + a2*b2 in float;
// the transposed product goes through GEMMSingleMul<float,double>: d = float(double-sum * alpha).
struct GemmOp : cv::MatOp { void assign(const cv::MatExpr &e, cv::Mat &m, int) const override; };
// a.t() * alpha assigned on its own (sR21 = (1.0 / s12) * R12.t()): cv::transpose, then Mat::convertTo with the scale -- cvtScale32f of
// OpenCV 3.3 works in float: d = s * float(alpha).  a * alpha / a / s (cv::operator*(double, Mat), operator/(Mat, double), -Mat) is the same conversion.
struct TransOp : cv::MatOp { void assign(const cv::MatExpr &e, cv::Mat &m, int) const override; };
struct ScaleOp : cv::MatOp { void assign(const cv::MatExpr &e, cv::Mat &m, int) const override; };
static GemmOp g_gemm_op;
static TransOp g_trans_op;
static ScaleOp g_scale_op;
static float matf(const cv::Mat &m, int r, int c) { return *(const float *)(m.data + m.step_buf[0] * r + 4 * (size_t)c); }
static void expr_init(cv::MatExpr *e, const cv::MatOp *op)
{
    e->op = op; e->flags = 0; e->alpha = 1; e->beta = 0; memset(&e->s, 0, sizeof(e->s));
    mat_empty(&e->a); mat_empty(&e->b); mat_empty(&e->c);
}
void GemmOp::assign(const cv::MatExpr &e, cv::Mat &m, int) const
{
    const cv::Mat &A = e.a, &B = e.b, &Cm = e.c;
    const bool tr = (e.flags & 1) != 0;
    if ((A.flags & 0xFFF) != 5 || A.rows != 3 || A.cols != 3 || B.rows != 3 || B.cols != 1) { fprintf(stderr, "refprobe: unexpected gemm shape\n"); abort(); }
    float d[3];
    for (int i = 0; i < 3; i++) {
        if (tr) {
            double sum = 0;
            for (int k = 0; k < 3; k++) sum += (double)matf(A, k, i) * (double)matf(B, k, 0);
            d[i] = (float)(sum * e.alpha);
            if (Cm.data) { fprintf(stderr, "refprobe: unexpected gemm form\n"); abort(); }
        } else {
            const float t = matf(A, i, 0) * matf(B, 0, 0) + matf(A, i, 1) * matf(B, 1, 0) + matf(A, i, 2) * matf(B, 2, 0);
            d[i] = (float)((double)t * e.alpha + (Cm.data ? e.beta * (double)matf(Cm, i, 0) : 0.0));
        }
    }
    mat_alloc(&m, 3, 1, 5);
    for (int i = 0; i < 3; i++) *(float *)(m.data + m.step_buf[0] * i) = d[i];
}
static void mat_alloc(cv::Mat *m, int rows, int cols, int type);
void TransOp::assign(const cv::MatExpr &e, cv::Mat &m, int) const
{
    const cv::Mat &A = e.a;
    if ((A.flags & 0xFFF) != 5 || A.rows > 4 || A.cols > 4) { fprintf(stderr, "refprobe: unexpected transpose shape\n"); abort(); }
    float d[16];
    const float al = (float)e.alpha;
    for (int r = 0; r < A.cols; r++) for (int c = 0; c < A.rows; c++) d[r * A.rows + c] = e.alpha == 1.0 ? matf(A, c, r) : matf(A, c, r) * al;
    const int R = A.cols, Cn = A.rows;
    mat_alloc(&m, R, Cn, 5);
    for (int r = 0; r < R; r++) for (int c = 0; c < Cn; c++) *(float *)(m.data + m.step_buf[0] * r + 4 * (size_t)c) = d[r * Cn + c];
}
void ScaleOp::assign(const cv::MatExpr &e, cv::Mat &m, int) const
{
    const cv::Mat &A = e.a;
    if ((A.flags & 0xFFF) != 5 || A.rows > 4 || A.cols > 4) { fprintf(stderr, "refprobe: unexpected scale shape\n"); abort(); }
    float d[16];
    const float al = (float)e.alpha;
    for (int r = 0; r < A.rows; r++) for (int c = 0; c < A.cols; c++) d[r * A.cols + c] = matf(A, r, c) * al;
    const int R = A.rows, Cn = A.cols;
    mat_alloc(&m, R, Cn, 5);
    for (int r = 0; r < R; r++) for (int c = 0; c < Cn; c++) *(float *)(m.data + m.step_buf[0] * r + 4 * (size_t)c) = d[r * Cn + c];
}
cv::MatExpr::~MatExpr() {}
cv::MatExpr cv::Mat::t() const { MatExpr e; expr_init(&e, &g_trans_op); e.a = *this; return e; }
cv::MatExpr cv::operator-(const MatExpr &x) { MatExpr e; expr_init(&e, x.op); e.flags = x.flags; e.a = x.a; e.b = x.b; e.c = x.c; e.alpha = -x.alpha; e.beta = -x.beta; return e; }
cv::MatExpr cv::operator*(const Mat &a, const Mat &b) { MatExpr e; expr_init(&e, &g_gemm_op); e.a = a; e.b = b; return e; }
cv::MatExpr cv::operator*(const MatExpr &x, const Mat &b)
{
    if (x.op == &g_scale_op) { MatExpr e; expr_init(&e, &g_gemm_op); e.a = x.a; e.b = b; e.alpha = x.alpha; return e; }   // (-A) * b = gemm(A, b, alpha = -1)
    if (x.op != &g_trans_op) { fprintf(stderr, "refprobe: unexpected expr * Mat\n"); abort(); }
    MatExpr e; expr_init(&e, &g_gemm_op); e.flags = 1; e.a = x.a; e.b = b; e.alpha = x.alpha; return e;
}
cv::MatExpr cv::operator+(const MatExpr &x, const Mat &c)
{
    if (x.op != &g_gemm_op || x.c.data) { fprintf(stderr, "refprobe: unexpected expr + Mat\n"); abort(); }
    MatExpr e; expr_init(&e, &g_gemm_op); e.flags = x.flags; e.a = x.a; e.b = x.b; e.alpha = x.alpha; e.c = c; e.beta = 1; return e;
}
struct SubOp : cv::MatOp { void assign(const cv::MatExpr &e, cv::Mat &m, int) const override; };
static SubOp g_sub_op;
void SubOp::assign(const cv::MatExpr &e, cv::Mat &m, int) const   // a - b, 3x1 CV_32F
{
    if ((e.a.flags & 0xFFF) != 5 || e.a.rows != 3 || e.a.cols != 1 || e.b.rows != 3 || e.b.cols != 1) { fprintf(stderr, "refprobe: unexpected a - b shape\n"); abort(); }
    float d[3];
    for (int i = 0; i < 3; i++) d[i] = matf(e.a, i, 0) - matf(e.b, i, 0);
    mat_alloc(&m, 3, 1, 5);
    for (int i = 0; i < 3; i++) *(float *)(m.data + m.step_buf[0] * i) = d[i];
}
cv::MatExpr cv::operator/(const Mat &a, double sc) { MatExpr e; expr_init(&e, &g_scale_op); e.a = a; e.alpha = 1.0 / sc; return e; }
cv::MatExpr cv::operator*(double sc, const Mat &a) { MatExpr e; expr_init(&e, &g_scale_op); e.a = a; e.alpha = sc; return e; }
cv::MatExpr cv::operator-(const Mat &a) { MatExpr e; expr_init(&e, &g_scale_op); e.a = a; e.alpha = -1.0; return e; }
cv::MatExpr cv::operator*(double sc, const MatExpr &x)
{
    if (x.op != &g_trans_op) { fprintf(stderr, "refprobe: unexpected double * expr\n"); abort(); }
    MatExpr e; expr_init(&e, &g_trans_op); e.a = x.a; e.alpha = x.alpha * sc; return e;
}
cv::MatExpr cv::operator-(const Mat &a, const Mat &b) { MatExpr e; expr_init(&e, &g_sub_op); e.a = a; e.b = b; return e; }
double cv::Mat::dot(const _InputArray &o) const   // CV_32F vectors: dotProd_<float> accumulates in double, element by element
{
    const Mat *b = (const Mat *)o.obj;
    if ((flags & 0xFFF) != 5 || (cols != 1 && rows != 1) || b->cols != cols || b->rows != rows) { fprintf(stderr, "refprobe: unexpected dot call\n"); abort(); }
    double r = 0;
    if (cols == 1) for (int i = 0; i < rows; i++) r += (double)matf(*this, i, 0) * (double)matf(*b, i, 0);
    else for (int i = 0; i < cols; i++) r += (double)matf(*this, 0, i) * (double)matf(*b, 0, i);
    return r;
}
static cv::_InputArray g_no_array = {0, nullptr, 0, 0};
const cv::_InputArray &cv::noArray() { return g_no_array; }
double cv::norm(const _InputArray &src, int normType, const _InputArray &)   // NORM_L2 of a CV_32F vector: double accumulation (normL2Sqr_<float, double>)
{
    const Mat *m = (const Mat *)src.obj;
    if (normType != 4 || (m->flags & 0xFFF) != 5 || m->cols != 1) { fprintf(stderr, "refprobe: unexpected norm call\n"); abort(); }
    double s2 = 0;
    for (int i = 0; i < m->rows; i++) { const double v = (double)matf(*m, i, 0); s2 += v * v; }
    return sqrt(s2);
}
int cv::_InputArray::kind() const { return flags & (31 << 16); }
bool cv::_InputArray::empty() const { const Mat *m = (const Mat *)obj; return m->data == nullptr || m->rows * m->cols == 0; }
cv::Mat cv::_InputArray::getMat_(int) const { return *(const Mat *)obj; }
void cv::_OutputArray::create(int rows, int cols, int type, int, bool, int) const { mat_alloc((Mat *)obj, rows, cols, type & 0xFFF); }
void cv::_OutputArray::release() const { mat_empty((Mat *)obj); }
void cv::resize(const _InputArray &src, const _OutputArray &dst, Size_<int> sz, double, double, int interp)
{
    const Mat *s = (const Mat *)src.obj; Mat *d = (Mat *)dst.obj;
    if (interp != 1 || d->cols != sz.width || d->rows != sz.height) { fprintf(stderr, "refprobe: unexpected resize call\n"); abort(); }
    orc_resize_linear_8u(s->data, s->cols, s->rows, (ptrdiff_t)s->step_buf[0], d->data, d->cols, d->rows, (ptrdiff_t)d->step_buf[0]);
}
void cv::copyMakeBorder(const _InputArray &src, const _OutputArray &dst, int t, int b, int l, int r, int type, const Scalar_<double> &)
{
    const Mat *s = (const Mat *)src.obj; Mat *d = (Mat *)dst.obj;
    if (t != 19 || b != 19 || l != 19 || r != 19 || (type & 15) != 4) { fprintf(stderr, "refprobe: unexpected copyMakeBorder call\n"); abort(); }
    mat_alloc(d, s->rows + 38, s->cols + 38, 0);
    unsigned char *inner = d->data + 19 * d->step_buf[0] + 19;
    if (inner != s->data)
        for (int y = 0; y < s->rows; y++) memmove(inner + d->step_buf[0] * y, s->data + s->step_buf[0] * y, (size_t)s->cols);
    orc_border_reflect101(d->data, s->cols, s->rows, (ptrdiff_t)d->step_buf[0], 19);
}
void cv::GaussianBlur(const _InputArray &src, const _OutputArray &dst, Size_<int> k, double sx, double sy, int border)
{
    const Mat *s = (const Mat *)src.obj; Mat *d = (Mat *)dst.obj;
    if (k.width != 7 || k.height != 7 || sx != 2.0 || sy != 2.0 || border != 4) { fprintf(stderr, "refprobe: unexpected GaussianBlur call\n"); abort(); }
    mat_alloc(d, s->rows, s->cols, 0);
    orc_gaussian_blur7_8u(s->data, (ptrdiff_t)s->step_buf[0], d->data, (ptrdiff_t)d->step_buf[0], s->cols, s->rows);
}

// ---- tier Q: Frame::UndistortKeyPoints (so@0xf8630) reaches Mat::reshape and cv::undistortPoints; the point arithmetic is oracle/frame_oracle.c
cv::Mat cv::Mat::reshape(int cn, int) const
{
    Mat m(*this);
    const int total_cols = cols * (((flags >> 3) & 511) + 1);
    if ((flags & 7) != 5 || total_cols % cn) { fprintf(stderr, "refprobe: unexpected reshape\n"); abort(); }
    m.cols = total_cols / cn;
    m.flags = (flags & ~0xFFF) | 5 | ((cn - 1) << 3);
    m.step_buf[1] = (size_t)4 * cn;
    return m;
}
void cv::undistortPoints(const _InputArray &src, const _OutputArray &dst, const _InputArray &K, const _InputArray &D, const _InputArray &R, const _InputArray &P)
{
    const Mat *s = (const Mat *)src.obj, *k = (const Mat *)K.obj, *d = (const Mat *)D.obj, *r = (const Mat *)R.obj, *pm = (const Mat *)P.obj;
    if (dst.obj != src.obj || (s->flags & 0xFFF) != 13 || s->cols != 1 || (r && r->data) || !pm || pm->data != k->data) { fprintf(stderr, "refprobe: unexpected undistortPoints call\n"); abort(); }
    const int n = s->rows, nd = d->rows * d->cols;
    float cam[9] = {matf(*k, 0, 0), matf(*k, 1, 1), matf(*k, 0, 2), matf(*k, 1, 2), 0, 0, 0, 0, 0};
    for (int i = 0; i < nd && i < 5; i++) cam[4 + i] = d->cols == 1 ? matf(*d, i, 0) : matf(*d, 0, i);
    std::vector<orc_keypoint> a(n), b(n);
    for (int i = 0; i < n; i++) { memset(&a[i], 0, sizeof(a[i])); a[i].x = *(float *)(s->data + s->step_buf[0] * i); a[i].y = *(float *)(s->data + s->step_buf[0] * i + 4); }
    orc_undistort_keypoints(a.data(), n, cam, b.data());
    for (int i = 0; i < n; i++) { *(float *)(s->data + s->step_buf[0] * i) = b[i].x; *(float *)(s->data + s->step_buf[0] * i + 4) = b[i].y; }
}

// ---------------------------------------------------------------- look-alike reference classes
namespace ORB_SLAM2 {
class ORBextractor {
public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    std::vector<cv::KeyPoint> DistributeOctTree(const std::vector<cv::KeyPoint> &, const int &, const int &,
                                                const int &, const int &, const int &, const int &);
    void ComputeKeyPointsOctTree(std::vector<std::vector<cv::KeyPoint>> &);
    void operator()(const cv::_InputArray &image, const cv::_InputArray &mask, std::vector<cv::KeyPoint> &keypoints,
                    const cv::_OutputArray &descriptors);
    char storage[1024];
};
class KeyFrame;
class MapPoint {   // hand-laid raw memory; the two map-mutating members Fuse calls are defined HERE (recorders, tier K): the binary reaches
public:            // them through its PLT, so the executable's definitions are the ones that run
    void AddObservation(KeyFrame *pKF, size_t idx);
    void Replace(MapPoint *pMP);
};
class Frame {   // only the exported statics are named; the object itself is hand-laid raw memory (tier D)
public:
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv, fx, fy, cx, cy;
    bool isInFrustum(MapPoint *pMP, float viewingCosLimit);
    void ComputeStereoFromRGBD(const cv::Mat &imDepth);
    void AssignFeaturesToGrid();
    void UndistortKeyPoints();
};
class ORBmatcher {
public:
    ORBmatcher(float nnratio, bool checkOri);
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th);
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo);
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th);
    static int DescriptorDistance(const cv::Mat &, const cv::Mat &);
    float RadiusByViewingCos(const float &);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &, int &, int &);
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
    char storage[64];
};
}  // namespace ORB_SLAM2

// recorders behind MapPoint::AddObservation / MapPoint::Replace (tier K)
struct FuseEvent { int kind; void *a, *b; size_t idx; };   // kind 0: a->AddObservation(kf, idx); 1: a->Replace(b)
static std::vector<FuseEvent> g_fuse_events;
void ORB_SLAM2::MapPoint::AddObservation(KeyFrame *, size_t idx) { FuseEvent e = {0, this, nullptr, idx}; g_fuse_events.push_back(e); }
void ORB_SLAM2::MapPoint::Replace(MapPoint *pMP) { FuseEvent e = {1, this, pMP, 0}; g_fuse_events.push_back(e); }

// ---------------------------------------------------------------- shared PRNG (same LCG in tests/refgen.py)
static uint64_t g_rng = 1;
static void rng_seed(uint64_t s) { g_rng = s; }
static uint32_t rng_u32() { g_rng = g_rng * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(g_rng >> 32); }
static uint32_t rng_below(uint32_t n) { return (uint32_t)(((uint64_t)rng_u32() * n) >> 32); }

template <class T> static std::vector<T> &vec_at(void *obj, size_t off) { return *(std::vector<T> *)((char *)obj + off); }

static FILE *J;
static void jarr_f(const char *name, const std::vector<float> &v, bool last = false)
{
    fprintf(J, "\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) { uint32_t b; memcpy(&b, &v[i], 4); fprintf(J, "%s%u", i ? "," : "", b); }
    fprintf(J, "]%s", last ? "" : ", ");
}
static void jarr_i(const char *name, const std::vector<int> &v, bool last = false)
{
    fprintf(J, "\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) fprintf(J, "%s%d", i ? "," : "", v[i]);
    fprintf(J, "]%s", last ? "" : ", ");
}

// synthetic image shared with tests (LCG blocks + noise) -- generator mirrored in tests/refgen.py
static void synth_image(uint64_t seed, int w, int h, std::vector<unsigned char> &img)
{
    img.assign((size_t)w * h, 0);
    rng_seed(seed);
    unsigned base = 96 + rng_below(64);
    for (size_t i = 0; i < img.size(); i++) img[i] = (unsigned char)base;
    int nrect = 40 + (int)rng_below(40);
    for (int r = 0; r < nrect; r++) {
        int x0 = (int)rng_below(w), y0 = (int)rng_below(h);
        int rw = 8 + (int)rng_below(w / 4), rh = 8 + (int)rng_below(h / 4);
        unsigned char v = (unsigned char)rng_below(256);
        for (int y = y0; y < y0 + rh && y < h; y++)
            for (int x = x0; x < x0 + rw && x < w; x++) img[(size_t)y * w + x] = v;
    }
    for (size_t i = 0; i < img.size(); i++) {
        int v = img[i] + (int)rng_below(9) - 4;
        img[i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

int main(int argc, char **argv)
{
    const char *outdir = argc > 1 ? argv[1] : ".";
    using namespace ORB_SLAM2;
    std::string path;

    // ------------------------------------------------------------ A1: ctor tables
    path = std::string(outdir) + "/ref_orb_tables.json";
    J = fopen(path.c_str(), "w");
    fprintf(J, "{\"_doc\": \"ORBextractor ctor executed from the reference binary; floats as uint32 bit patterns\", \"cases\": [\n");
    struct { int nf; float sf; int nl; } cfgs[] = {{1000, 1.2f, 8}, {2000, 1.2f, 8}, {4000, 1.2f, 8}, {500, 1.2f, 8},
                                                    {1500, 1.5f, 5}, {1200, 1.1f, 12}, {300, 2.0f, 4}, {1000, 1.2f, 1}};
    for (size_t c = 0; c < sizeof(cfgs) / sizeof(cfgs[0]); c++) {
        ORBextractor *e = new ORBextractor(cfgs[c].nf, cfgs[c].sf, cfgs[c].nl, 20, 7);
        fprintf(J, "{\"nfeatures\": %d, \"scaleFactor_bits\": %u, \"nlevels\": %d, ", cfgs[c].nf,
                *(uint32_t *)&cfgs[c].sf, cfgs[c].nl);
        jarr_i("perLevel", vec_at<int>(e, 0x50));
        jarr_i("umax", vec_at<int>(e, 0x68));
        jarr_f("scale", vec_at<float>(e, 0x80));
        jarr_f("inv", vec_at<float>(e, 0x98));
        jarr_f("sigma2", vec_at<float>(e, 0xb0));
        jarr_f("invsigma2", vec_at<float>(e, 0xc8), true);
        fprintf(J, "}%s\n", c + 1 < sizeof(cfgs) / sizeof(cfgs[0]) ? "," : "");
        if (c == 0) {  // pattern copy inside the object must equal the .data table
            std::vector<int> &pat = vec_at<int>(e, 0x18);  // vector<cv::Point> viewed as ints
            FILE *pf = fopen((std::string(outdir) + "/ref_pattern_from_ctor.bin").c_str(), "wb");
            fwrite(pat.data(), 4, 1024, pf);
            fclose(pf);
        }
    }
    fprintf(J, "]}\n");
    fclose(J);

    // ------------------------------------------------------------ A2: DistributeOctTree
    path = std::string(outdir) + "/ref_octree_cases.json";
    J = fopen(path.c_str(), "w");
    fprintf(J, "{\"_doc\": \"DistributeOctTree executed from the reference binary under a monotonic allocator. "
               "Inputs are regenerated from the LCG (tests/refgen.py: octree_case); out = class_id (input index) of each returned keypoint, in order\", \"cases\": [\n");
    ORBextractor *ex = new ORBextractor(1000, 1.2f, 8, 20, 7);
    const int NOCT = 48;
    for (int c = 0; c < NOCT; c++) {
        arena_reset();
        ex = new ORBextractor(1000, 1.2f, 8, 20, 7);
        rng_seed(1000 + c);
        int W = 64 + (int)rng_below(1200), H = 48 + (int)rng_below(900);
        if (W < H) { int t = W; W = H; H = t; }          // landscape (portrait > 2:1 divides by zero in the reference)
        int nk = 1 + (int)rng_below(c % 4 == 0 ? 60 : 3500);
        int N = 1 + (int)rng_below(450);
        int resp_levels = (c % 3 == 0) ? 4 : 200;          // tie-heavy vs tie-light responses
        int cluster = (c % 5 == 1);
        std::vector<cv::KeyPoint> in(nk);
        for (int i = 0; i < nk; i++) {
            int x, y;
            if (cluster) { x = (int)rng_below(W / 4 + 1) + (int)rng_below(2) * (W / 2); y = (int)rng_below(H / 3 + 1); }
            else { x = (int)rng_below(W); y = (int)rng_below(H); }
            if (x >= W) x = W - 1;
            cv::KeyPoint k = {(float)x, (float)y, 7.f, -1.f, (float)(7 + rng_below(resp_levels)), 0, i};
            in[i] = k;
        }
        int minX = 16, maxX = 16 + W, minY = 16, maxY = 16 + H, lvl = 0;
        std::vector<cv::KeyPoint> out = ex->DistributeOctTree(in, minX, maxX, minY, maxY, N, lvl);
        std::vector<int> ids(out.size());
        for (size_t i = 0; i < out.size(); i++) ids[i] = out[i].class_id;
        fprintf(J, "{\"seed\": %d, \"W\": %d, \"H\": %d, \"nk\": %d, \"N\": %d, \"resp_levels\": %d, \"cluster\": %d, ", 1000 + c, W, H, nk,
                N, resp_levels, cluster);
        jarr_i("out", ids, true);
        fprintf(J, "}%s\n", c + 1 < NOCT ? "," : "");
    }
    fprintf(J, "]}\n");
    fclose(J);

    // ------------------------------------------------------------ A3: computeOrientation integer moments
    {
        struct link_map *lm = nullptr;
        void *h = dlopen("libORB_SLAM2.so", RTLD_LAZY | RTLD_NOLOAD);
        if (!h) h = dlopen("/root/reference/lib/libORB_SLAM2.so", RTLD_LAZY);
        dlinfo(h, RTLD_DI_LINKMAP, &lm);
        typedef void (*orient_fn)(const cv::Mat &, std::vector<cv::KeyPoint> &, const std::vector<int> &);
        orient_fn computeOrientation = (orient_fn)((char *)lm->l_addr + 0x6fb10);
        path = std::string(outdir) + "/ref_ic_moments.json";
        J = fopen(path.c_str(), "w");
        fprintf(J, "{\"_doc\": \"computeOrientation (so@0x6fb10) executed with a recording cv::fastAtan2: (m01,m10) per keypoint; images from tests/refgen.py: synth_image\", \"cases\": [\n");
        std::vector<int> &umax = vec_at<int>(ex, 0x68);
        for (int c = 0; c < 6; c++) {
            int w = 96 + 16 * c, hh = 80 + 8 * c;
            std::vector<unsigned char> img;
            synth_image(7000 + c, w, hh, img);
            cv::Mat m; mat_init(&m, img.data(), hh, w, w);
            rng_seed(7100 + c);
            std::vector<cv::KeyPoint> kps;
            std::vector<int> xs, ys;
            for (int i = 0; i < 40; i++) {
                int x = 16 + (int)rng_below(w - 32), y = 16 + (int)rng_below(hh - 32);
                cv::KeyPoint k = {(float)x, (float)y, 31.f, -1.f, 20.f, 0, i};
                kps.push_back(k); xs.push_back(x); ys.push_back(y);
            }
            g_mode = 0; g_atan_rec.clear();
            computeOrientation(m, kps, umax);
            std::vector<int> m01, m10;
            for (size_t i = 0; i + 1 < g_atan_rec.size(); i += 2) { m01.push_back((int)g_atan_rec[i]); m10.push_back((int)g_atan_rec[i + 1]); }
            fprintf(J, "{\"seed\": %d, \"w\": %d, \"h\": %d, \"pt_seed\": %d, ", 7000 + c, w, hh, 7100 + c);
            jarr_i("x", xs); jarr_i("y", ys); jarr_i("m01", m01); jarr_i("m10", m10, true);
            fprintf(J, "}%s\n", c + 1 < 6 ? "," : "");
        }
        fprintf(J, "]}\n");
        fclose(J);
    }

    // ------------------------------------------------------------ A4: matcher scalars
    {
        path = std::string(outdir) + "/ref_matcher.json";
        J = fopen(path.c_str(), "w");
        ORBmatcher *mt = new ORBmatcher(0.8f, true);
        fprintf(J, "{\"_doc\": \"ORBmatcher constants and pure functions executed from the reference binary\",\n");
        fprintf(J, "\"TH_LOW\": %d, \"TH_HIGH\": %d, \"HISTO_LENGTH\": %d,\n", ORBmatcher::TH_LOW, ORBmatcher::TH_HIGH, ORBmatcher::HISTO_LENGTH);
        std::vector<float> cs = {1.0f, 0.9999f, 0.9981f, 0.998f, 0.99799f, 0.99f, 0.5f, 0.0f, -1.0f}, rs;
        for (float c : cs) rs.push_back(mt->RadiusByViewingCos(c));
        jarr_f("radius_cos", cs); jarr_f("radius", rs);
        // Hamming
        fprintf(J, "\n\"hamming\": [");
        rng_seed(4242);
        for (int c = 0; c < 64; c++) {
            unsigned char a[32], b[32];
            for (int i = 0; i < 32; i++) { a[i] = (unsigned char)rng_below(256); b[i] = (c % 4 == 0) ? (unsigned char)(a[i] ^ (1u << rng_below(8))) : (unsigned char)rng_below(256); }
            if (c == 0) memcpy(b, a, 32);
            if (c == 1) for (int i = 0; i < 32; i++) b[i] = (unsigned char)~a[i];
            cv::Mat ma, mb; mat_init(&ma, a, 1, 32, 32); mat_init(&mb, b, 1, 32, 32);
            int d = ORBmatcher::DescriptorDistance(ma, mb);
            fprintf(J, "%s{\"a\": \"", c ? "," : "");
            for (int i = 0; i < 32; i++) fprintf(J, "%02x", a[i]);
            fprintf(J, "\", \"b\": \"");
            for (int i = 0; i < 32; i++) fprintf(J, "%02x", b[i]);
            fprintf(J, "\", \"d\": %d}", d);
        }
        fprintf(J, "],\n\"three_maxima\": [");
        rng_seed(999);
        for (int c = 0; c < 200; c++) {
            std::vector<int> histo[30];
            std::vector<int> sizes;
            int mode = c % 4;
            for (int b = 0; b < 30; b++) {
                int n = mode == 0 ? (int)rng_below(50) : mode == 1 ? (int)rng_below(4) : mode == 2 ? ((int)rng_below(10) == 0 ? 100 + (int)rng_below(3) : (int)rng_below(8)) : (int)rng_below(200);
                histo[b].assign(n, 0); sizes.push_back(n);
            }
            int i1 = -1, i2 = -1, i3 = -1;
            mt->ComputeThreeMaxima(histo, 30, i1, i2, i3);
            fprintf(J, "%s{", c ? "," : "");
            jarr_i("sizes", sizes);
            fprintf(J, "\"ind\": [%d,%d,%d]}", i1, i2, i3);
        }
        fprintf(J, "]}\n");
        fclose(J);
    }

    // ------------------------------------------------------------ B: ComputeKeyPointsOctTree (glue)
    {
        path = std::string(outdir) + "/ref_cells.json";
        FILE *JC = fopen(path.c_str(), "w");
        fprintf(JC, "{\"_doc\": \"cv::FAST call rectangles issued by the reference ComputeKeyPointsOctTree (so@0x75fa0), interior coords, per level\", \"cases\": [\n");
        path = std::string(outdir) + "/ref_glue_keypoints.json";
        FILE *JG = fopen(path.c_str(), "w");
        fprintf(JG, "{\"_doc\": \"ComputeKeyPointsOctTree from the reference binary on a pyramid built by oracle/orb_oracle.c from tests/refgen.py:synth_image, cv::FAST/fastAtan2 = this repo's restatement. per level: x,y,size,angle,response bit patterns\", \"cases\": [\n");
        struct { int w, h, nf; uint64_t seed; } gc[] = {{640, 480, 1000, 9001}, {1280, 960, 4000, 9002}, {320, 240, 500, 9003}, {752, 480, 1200, 9004}};
        const int NG = 4;
        for (int c = 0; c < NG; c++) {
            arena_reset();
            ORBextractor *e = new ORBextractor(gc[c].nf, 1.2f, 8, 20, 7);
            std::vector<unsigned char> img;
            synth_image(gc[c].seed, gc[c].w, gc[c].h, img);
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(gc[c].nf, 1.2f, 8, scale, inv, s2, is2, per, um);
            uint8_t *planes[16]; int lw[16], lh[16];
            orc_compute_pyramid(img.data(), gc[c].w, gc[c].h, gc[c].w, 8, inv, planes, lw, lh);
            cv::Mat *mats = (cv::Mat *)bump(sizeof(cv::Mat) * 8);
            for (int l = 0; l < 8; l++) {
                size_t pp = lw[l] + 38;
                mat_init(&mats[l], planes[l] + 19 * pp + 19, lh[l], lw[l], pp);
            }
            void **vec = (void **)e;  // mvImagePyramid is the first member (offset 0 of the object)
            vec[0] = mats; vec[1] = mats + 8; vec[2] = mats + 8;
            for (int pass = 0; pass < 3; pass++) {
                // pass 0: recorder (every call returns 1 kp), pass 1: recorder with forced retry, pass 2: real FAST
                // The per-level base pointer is needed to decode the rectangles, so hook through a wrapper:
                g_fast_calls.clear();
                g_fast_mode = pass == 2 ? 1 : 0; g_force_retry = pass == 1; g_mode = 1;
                std::vector<std::vector<cv::KeyPoint>> all;
                e->ComputeKeyPointsOctTree(all);
                if (pass < 2) {
                    fprintf(JC, "{\"w\": %d, \"h\": %d, \"force_retry\": %d, \"levels\": [", gc[c].w, gc[c].h, pass);
                    // re-decode rectangles per level
                    std::vector<std::vector<int>> rects(8);
                    for (const FastCall &fc : g_fast_calls) {
                        const unsigned char *p = fc.p;
                        for (int l = 0; l < 8; l++) {
                            size_t pp = lw[l] + 38;
                            const unsigned char *lo = planes[l], *hi = planes[l] + pp * (lh[l] + 38);
                            if (p >= lo && p < hi) {
                                ptrdiff_t off = p - (planes[l] + 19 * pp + 19);
                                int yy = (int)floor((double)off / (double)pp); int xx = (int)(off - (ptrdiff_t)yy * (ptrdiff_t)pp);
                                rects[l].push_back(xx); rects[l].push_back(yy); rects[l].push_back(fc.w); rects[l].push_back(fc.h);
                                rects[l].push_back(fc.th); rects[l].push_back(fc.nonmax);
                                break;
                            }
                        }
                    }
                    for (int l = 0; l < 8; l++) {
                        fprintf(JC, "%s[", l ? "," : "");
                        for (size_t i = 0; i < rects[l].size(); i++) fprintf(JC, "%s%d", i ? "," : "", rects[l][i]);
                        fprintf(JC, "]");
                    }
                    fprintf(JC, "]}%s\n", (c + 1 < NG || pass < 1) ? "," : "");
                } else {
                    fprintf(JG, "{\"seed\": %llu, \"w\": %d, \"h\": %d, \"nfeatures\": %d, \"levels\": [", (unsigned long long)gc[c].seed, gc[c].w, gc[c].h, gc[c].nf);
                    for (int l = 0; l < 8; l++) {
                        std::vector<float> flat;
                        for (const cv::KeyPoint &k : all[l]) { flat.push_back(k.x); flat.push_back(k.y); flat.push_back(k.size); flat.push_back(k.angle); flat.push_back(k.response); }
                        fprintf(JG, "%s{", l ? "," : "");
                        J = JG;
                        jarr_f("kp", flat, true);
                        fprintf(JG, "}");
                    }
                    fprintf(JG, "]}%s\n", c + 1 < NG ? "," : "");
                }
            }
            for (int l = 0; l < 8; l++) free(planes[l]);
        }
        fprintf(JC, "]}\n"); fclose(JC);
        fprintf(JG, "]}\n"); fclose(JG);
    }
    // ------------------------------------------------------------ C: the whole ORBextractor::operator() (glue)
    {
        path = std::string(outdir) + "/ref_glue_operator.json";
        FILE *JO = fopen(path.c_str(), "w");
        fprintf(JO, "{\"_doc\": \"ORBextractor::operator() (so@0x76da0) executed from the reference binary on tests/refgen.py:synth_image; cv::resize / copyMakeBorder / "
                    "GaussianBlur / FAST / fastAtan2 are this repo's restatements (oracle/orb_oracle.c), everything else -- pyramid sizes, cell loop, octree, "
                    "orientation, blur-then-BRIEF with sincosf and FMA, scaling, output order -- is the reference's machine code. kp: x,y,size,angle,response "
                    "bit patterns, octave; desc: hex\", \"cases\": [\n");
        struct { int w, h, nf; uint64_t seed; } oc[] = {{640, 480, 1000, 9101}, {320, 240, 500, 9102}, {752, 480, 1200, 9103}};
        const int NO = 3;
        g_fast_mode = 1; g_force_retry = 0; g_mode = 1;
        for (int c = 0; c < NO; c++) {
            arena_reset();
            ORBextractor *e = new ORBextractor(oc[c].nf, 1.2f, 8, 20, 7);
            std::vector<unsigned char> img;
            synth_image(oc[c].seed, oc[c].w, oc[c].h, img);
            cv::Mat im, dm, mm;
            mat_init(&im, img.data(), oc[c].h, oc[c].w, (size_t)oc[c].w);
            im.flags |= 0x4000;
            mat_empty(&dm); mat_empty(&mm);
            cv::_InputArray ia; ia.flags = 0x01010000; ia.obj = &im; ia.sz_w = ia.sz_h = 0;
            cv::_InputArray ma; ma.flags = 0x01010000; ma.obj = &mm; ma.sz_w = ma.sz_h = 0;
            cv::_OutputArray oa; oa.flags = 0x02010000; oa.obj = &dm; oa.sz_w = oa.sz_h = 0;
            std::vector<cv::KeyPoint> kps;
            (*e)(ia, ma, kps, oa);
            std::vector<float> flat; std::vector<int> oct;
            for (const cv::KeyPoint &k : kps) { flat.push_back(k.x); flat.push_back(k.y); flat.push_back(k.size); flat.push_back(k.angle); flat.push_back(k.response); oct.push_back(k.octave); }
            fprintf(JO, "{\"seed\": %llu, \"w\": %d, \"h\": %d, \"nfeatures\": %d, \"n\": %d, \"desc_rows\": %d, ", (unsigned long long)oc[c].seed, oc[c].w, oc[c].h,
                    oc[c].nf, (int)kps.size(), dm.rows);
            J = JO;
            jarr_f("kp", flat);
            jarr_i("octave", oct);
            fprintf(JO, "\"desc\": \"");
            for (int r = 0; r < dm.rows; r++)
                for (int b = 0; b < 32; b++) fprintf(JO, "%02x", dm.data[dm.step_buf[0] * r + b]);
            fprintf(JO, "\"}%s\n", c + 1 < NO ? "," : "");
        }
        fprintf(JO, "]}\n"); fclose(JO);
    }
    // ------------------------------------------------------------ D: ORBmatcher::SearchByProjection(Frame&, map points, th) (glue)
    // Frame / MapPoint objects are hand-laid at the offsets the binary's code uses (read from the disassembly of
    // so@0x79f10, Frame::GetFeaturesInArea so@0xfbc60, MapPoint::isBad / Observations / GetDescriptor):
    //   Frame: N @0xec, mvKeysUn @0x120, mvuRight @0x138, mDescriptors @0x1c8, mvpMapPoints @0x288, mGrid[64][48] @0x2c8,
    //          mvScaleFactors @0x12348; statics mnMinX/mnMinY/mfGridElement{Width,Height}Inv are exported data symbols
    //   MapPoint: nObs @0x18, mTrackProjX/Y/XR @0x1c/0x20/0x24, mnTrackScaleLevel @0x28, mTrackViewCos @0x2c,
    //          mbTrackInView @0x30, mDescriptor @0x1c8, mbBad @0x238, mutexes zero-initialised
    {
        path = std::string(outdir) + "/ref_glue_search_map.json";
        FILE *JS = fopen(path.c_str(), "w");
        fprintf(JS, "{\"_doc\": \"ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (so@0x79f10) with Frame::GetFeaturesInArea, MapPoint::isBad/"
                    "Observations/GetDescriptor, DescriptorDistance, RadiusByViewingCos all executed from the reference binary on hand-laid objects. floats as "
                    "uint32 bit patterns; match[k] = index of the map point assigned to key point k, -1 none, -2 occupied before the call by a point with "
                    "observations\", \"cases\": [\n");
        struct { int n, m; float th; int with_ur; uint64_t seed; } sc[] = {{400, 1500, 3.0f, 0, 9201}, {300, 900, 1.0f, 1, 9202}, {500, 2500, 5.0f, 1, 9203}};
        const int NS = 3;
        for (int c = 0; c < NS; c++) {
            arena_reset();
            rng_seed(sc[c].seed);
            const int N = sc[c].n, M = sc[c].m;
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            std::vector<cv::KeyPoint> keys(N);
            std::vector<float> uright(N);
            std::vector<uint8_t> desc((size_t)N * 32);
            for (int i = 0; i < N; i++) {
                keys[i].x = 20.f + uf() * 600.f; keys[i].y = 20.f + uf() * 440.f; keys[i].size = 31.f; keys[i].angle = uf() * 360.f;
                keys[i].response = 30.f; keys[i].octave = (int)rng_below(8); keys[i].class_id = -1;
                uright[i] = (sc[c].with_ur && uf() < 0.7f) ? keys[i].x - (2.f + uf() * 28.f) : -1.f;
                for (int b = 0; b < 32; b++) desc[(size_t)i * 32 + b] = (uint8_t)rng_below(256);
            }
            // map points
            std::vector<float> px(M), py(M), pxr(M), vc(M);
            std::vector<int> lvl(M), nobs(M), inview(M), bad(M);
            std::vector<uint8_t> mdesc((size_t)M * 32);
            for (int m = 0; m < M; m++) {
                const int src = (int)rng_below(N);
                const bool real = uf() < 0.6f;
                px[m] = real ? keys[src].x + (uf() - 0.5f) * 6.f : -40.f + uf() * 720.f;
                py[m] = real ? keys[src].y + (uf() - 0.5f) * 6.f : -40.f + uf() * 560.f;
                lvl[m] = real ? std::min(7, keys[src].octave + (int)rng_below(2)) : (int)rng_below(8);
                for (int b = 0; b < 32; b++) mdesc[(size_t)m * 32 + b] = real ? desc[(size_t)src * 32 + b] : (uint8_t)rng_below(256);
                if (real) { const int flips = (int)rng_below(41); for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); mdesc[(size_t)m * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); } }
                vc[m] = 0.99f + uf() * 0.01f;
                inview[m] = uf() < 0.9f; bad[m] = uf() < 0.05f;
                pxr[m] = px[m] - (2.f + uf() * 28.f);
                nobs[m] = uf() < 0.1f ? 0 : 1 + (int)rng_below(4);
            }
            // key points already holding a map point before the call: with observations (blocks) or without (does not)
            std::vector<int> init(N, -1);
            char *frame = (char *)bump(0x12800); memset(frame, 0, 0x12800);
            char *mps = (char *)bump((size_t)(M + 2) * 0x300); memset(mps, 0, (size_t)(M + 2) * 0x300);
            char *occ_obs = mps + (size_t)M * 0x300, *occ_noobs = mps + (size_t)(M + 1) * 0x300;
            *(int *)(occ_obs + 0x18) = 2; *(int *)(occ_noobs + 0x18) = 0;
            std::vector<void *> fmp(N, nullptr);
            for (int i = 0; i < N; i++) { const float u = uf(); if (u < 0.08f) { fmp[i] = occ_obs; init[i] = -2; } else if (u < 0.12f) fmp[i] = occ_noobs; }
            cv::Mat *mdm = (cv::Mat *)bump(sizeof(cv::Mat));
            std::vector<void *> vp(M);
            for (int m = 0; m < M; m++) {
                char *o = mps + (size_t)m * 0x300;
                *(int *)(o + 0x18) = nobs[m]; *(float *)(o + 0x1c) = px[m]; *(float *)(o + 0x20) = py[m]; *(float *)(o + 0x24) = pxr[m];
                *(int *)(o + 0x28) = lvl[m]; *(float *)(o + 0x2c) = vc[m]; *(bool *)(o + 0x30) = inview[m] != 0; *(bool *)(o + 0x238) = bad[m] != 0;
                mat_init((cv::Mat *)(o + 0x1c8), &mdesc[(size_t)m * 32], 1, 32, 32);
                ((cv::Mat *)(o + 0x1c8))->flags |= 0x4000;
                vp[m] = o;
            }
            (void)mdm;
            // frame
            *(int *)(frame + 0xec) = N;
            void **v;
            v = (void **)(frame + 0x120); v[0] = keys.data(); v[1] = keys.data() + N; v[2] = keys.data() + N;
            v = (void **)(frame + 0x138); v[0] = uright.data(); v[1] = uright.data() + N; v[2] = uright.data() + N;
            mat_init((cv::Mat *)(frame + 0x1c8), desc.data(), N, 32, 32); ((cv::Mat *)(frame + 0x1c8))->flags |= 0x4000;
            v = (void **)(frame + 0x288); v[0] = fmp.data(); v[1] = fmp.data() + N; v[2] = fmp.data() + N;
            v = (void **)(frame + 0x12348); v[0] = scale; v[1] = scale + 8; v[2] = scale + 8;
            Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = 640.f; Frame::mnMaxY = 480.f;
            Frame::mfGridElementWidthInv = 64.f / (Frame::mnMaxX - Frame::mnMinX); Frame::mfGridElementHeightInv = 48.f / (Frame::mnMaxY - Frame::mnMinY);
            // Frame::AssignFeaturesToGrid (so@0xf9120): push_back(i) into cell (round((x - minX) * invW), round((y - minY) * invH))
            std::vector<std::vector<size_t>> cells(64 * 48);
            for (int i = 0; i < N; i++) {
                const int gx = (int)roundf((keys[i].x - Frame::mnMinX) * Frame::mfGridElementWidthInv), gy = (int)roundf((keys[i].y - Frame::mnMinY) * Frame::mfGridElementHeightInv);
                if (gx < 0 || gx >= 64 || gy < 0 || gy >= 48) continue;
                cells[gx * 48 + gy].push_back((size_t)i);
            }
            for (int cidx = 0; cidx < 64 * 48; cidx++) {
                v = (void **)(frame + 0x2c8 + (size_t)cidx * 24);
                v[0] = cells[cidx].data(); v[1] = cells[cidx].data() + cells[cidx].size(); v[2] = v[1];
            }
            ORBmatcher *mt = new ORBmatcher(0.8f, true);
            const std::vector<MapPoint *> &vpr = *(const std::vector<MapPoint *> *)&vp;
            const int nm = mt->SearchByProjection(*(Frame *)frame, vpr, sc[c].th);
            std::vector<int> match(N);
            for (int i = 0; i < N; i++) {
                const char *q = (const char *)fmp[i];
                match[i] = q == nullptr ? -1 : q == occ_obs ? -2 : q == occ_noobs ? -1 : (int)((q - mps) / 0x300);
            }
            // fixture: inputs + result
            std::vector<float> kx(N), ky(N); std::vector<int> ko(N), iv(M), ob(M);
            for (int i = 0; i < N; i++) { kx[i] = keys[i].x; ky[i] = keys[i].y; ko[i] = keys[i].octave; }
            for (int m = 0; m < M; m++) { iv[m] = inview[m] && !bad[m]; ob[m] = nobs[m] > 0; }
            std::vector<float> scv(scale, scale + 8);
            uint32_t thb; memcpy(&thb, &sc[c].th, 4);
            fprintf(JS, "{\"n\": %d, \"m\": %d, \"th_bits\": %u, \"with_uright\": %d, \"nmatches\": %d, ", N, M, thb, sc[c].with_ur, nm);
            J = JS;
            jarr_f("x", kx); jarr_f("y", ky); jarr_i("octave", ko); jarr_f("uright", uright); jarr_f("scale", scv);
            jarr_f("proj_x", px); jarr_f("proj_y", py); jarr_f("proj_xr", pxr); jarr_i("level", lvl); jarr_f("view_cos", vc);
            jarr_i("in_view", iv); jarr_i("obs_positive", ob); jarr_i("init", init); jarr_i("match", match);
            fprintf(JS, "\"desc\": \"");
            for (size_t b = 0; b < desc.size(); b++) fprintf(JS, "%02x", desc[b]);
            fprintf(JS, "\", \"mp_desc\": \"");
            for (size_t b = 0; b < mdesc.size(); b++) fprintf(JS, "%02x", mdesc[b]);
            fprintf(JS, "\"}%s\n", c + 1 < NS ? "," : "");
        }
        fprintf(JS, "]}\n"); fclose(JS);
    }
    // ------------------------------------------------------------ E: ORBmatcher::SearchByBoW(KeyFrame*, Frame&, matches) (glue)
    // KeyFrame (offsets from so@0x80150 and KeyFrame::GetMapPointMatches so@0x9c4c0): mvKeysUn @0x170, mDescriptors @0x1b8,
    // mFeatVec (std::map<unsigned, vector<unsigned>>) @0x248, mvpMapPoints @0x520, mMutexFeatures @0x690.
    // Frame: N @0xec, mvKeys @0xf0, mFeatVec @0x198, mDescriptors @0x1c8.
    {
        path = std::string(outdir) + "/ref_glue_bow.json";
        FILE *JB = fopen(path.c_str(), "w");
        fprintf(JB, "{\"_doc\": \"ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (so@0x80150) executed from the reference binary on hand-laid KeyFrame / "
                    "Frame / MapPoint objects with real std::map feature vectors. floats as uint32 bit patterns; match[j] = keyframe feature whose map point frame "
                    "feature j received, -1 none\", \"cases\": [\n");
        struct { int nkf, nf, nodes; int check; uint64_t seed; } bc[] = {{600, 500, 150, 1, 9301}, {300, 400, 60, 0, 9302}, {800, 900, 300, 1, 9303}};
        const int NBC = 3;
        typedef std::map<unsigned, std::vector<unsigned>> FeatVec;
        for (int c = 0; c < NBC; c++) {
            rng_seed(bc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int NK = bc[c].nkf, NF = bc[c].nf, NN = bc[c].nodes;
            std::vector<cv::KeyPoint> kkeys(NK), fkeys(NF);
            std::vector<uint8_t> kdesc((size_t)NK * 32), fdesc((size_t)NF * 32);
            std::vector<int> has(NK), bad(NK);
            std::vector<unsigned> knode(NK), fnode(NF);
            for (int i = 0; i < NK; i++) {
                kkeys[i].x = uf() * 640.f; kkeys[i].y = uf() * 480.f; kkeys[i].size = 31.f; kkeys[i].angle = uf() * 360.f; kkeys[i].response = 1.f; kkeys[i].octave = 0; kkeys[i].class_id = -1;
                for (int b = 0; b < 32; b++) kdesc[(size_t)i * 32 + b] = (uint8_t)rng_below(256);
                has[i] = uf() < 0.8f; bad[i] = uf() < 0.05f;
                knode[i] = 1000u + 7u * rng_below((uint32_t)NN);
            }
            for (int j = 0; j < NF; j++) {
                fkeys[j].x = uf() * 640.f; fkeys[j].y = uf() * 480.f; fkeys[j].size = 31.f; fkeys[j].response = 1.f; fkeys[j].octave = 0; fkeys[j].class_id = -1;
                const float u = uf();
                if (u < 0.7f) {   // a (noisy) observation of a keyframe feature: same word, similar descriptor, coherent rotation most of the time
                    const int src = (int)rng_below((uint32_t)NK);
                    for (int b = 0; b < 32; b++) fdesc[(size_t)j * 32 + b] = kdesc[(size_t)src * 32 + b];
                    const int flips = (int)rng_below(36);
                    for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); fdesc[(size_t)j * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                    fnode[j] = uf() < 0.9f ? knode[src] : 1000u + 7u * rng_below((uint32_t)NN) + (uf() < 0.3f ? 3u : 0u);
                    float a = kkeys[src].angle - (uf() < 0.8f ? 20.f + uf() * 8.f : uf() * 360.f);
                    if (a < 0.f) a += 360.f;
                    fkeys[j].angle = a;
                } else {
                    for (int b = 0; b < 32; b++) fdesc[(size_t)j * 32 + b] = (uint8_t)rng_below(256);
                    fnode[j] = 1000u + 7u * rng_below((uint32_t)NN) + (uf() < 0.3f ? 3u : 0u);
                    fkeys[j].angle = uf() * 360.f;
                }
            }
            char *kf = (char *)bump(0x800); memset(kf, 0, 0x800);
            char *fr = (char *)bump(0x12800); memset(fr, 0, 0x12800);
            char *mps = (char *)bump((size_t)NK * 0x300); memset(mps, 0, (size_t)NK * 0x300);
            std::vector<void *> kmp(NK, nullptr);
            for (int i = 0; i < NK; i++) if (has[i]) { kmp[i] = mps + (size_t)i * 0x300; *(bool *)(mps + (size_t)i * 0x300 + 0x238) = bad[i] != 0; }
            void **v;
            v = (void **)(kf + 0x170); v[0] = kkeys.data(); v[1] = kkeys.data() + NK; v[2] = v[1];
            mat_init((cv::Mat *)(kf + 0x1b8), kdesc.data(), NK, 32, 32); ((cv::Mat *)(kf + 0x1b8))->flags |= 0x4000;
            FeatVec *kfv = new (kf + 0x248) FeatVec();
            for (int i = 0; i < NK; i++) (*kfv)[knode[i]].push_back((unsigned)i);
            v = (void **)(kf + 0x520); v[0] = kmp.data(); v[1] = kmp.data() + NK; v[2] = v[1];
            *(int *)(fr + 0xec) = NF;
            v = (void **)(fr + 0xf0); v[0] = fkeys.data(); v[1] = fkeys.data() + NF; v[2] = v[1];
            FeatVec *ffv = new (fr + 0x198) FeatVec();
            for (int j = 0; j < NF; j++) (*ffv)[fnode[j]].push_back((unsigned)j);
            mat_init((cv::Mat *)(fr + 0x1c8), fdesc.data(), NF, 32, 32); ((cv::Mat *)(fr + 0x1c8))->flags |= 0x4000;
            ORBmatcher *mt = new ORBmatcher(0.7f, bc[c].check != 0);
            std::vector<MapPoint *> out;
            const int nm = mt->SearchByBoW((KeyFrame *)kf, *(Frame *)fr, out);
            std::vector<int> match(NF, -1);
            for (int j = 0; j < NF && j < (int)out.size(); j++) if (out[j]) match[j] = (int)(((char *)out[j] - mps) / 0x300);
            // fixture: flattened inputs + result
            std::vector<float> ka(NK), fa(NF); std::vector<int> hm(NK);
            for (int i = 0; i < NK; i++) { ka[i] = kkeys[i].angle; hm[i] = has[i] && !bad[i]; }
            for (int j = 0; j < NF; j++) fa[j] = fkeys[j].angle;
            auto flat = [&](FeatVec *fv, std::vector<int> &ids, std::vector<int> &starts, std::vector<int> &feats) {
                for (auto &kv : *fv) { ids.push_back((int)kv.first); starts.push_back((int)feats.size()); for (unsigned x : kv.second) feats.push_back((int)x); }
                starts.push_back((int)feats.size());
            };
            std::vector<int> kid, kst, kfe, fid, fst, ffe;
            flat(kfv, kid, kst, kfe); flat(ffv, fid, fst, ffe);
            fprintf(JB, "{\"n_kf\": %d, \"n_f\": %d, \"nnratio\": 0.7, \"check_orientation\": %d, \"nmatches\": %d, ", NK, NF, bc[c].check, nm);
            J = JB;
            jarr_f("kf_angle", ka); jarr_f("f_angle", fa); jarr_i("kf_has_mp", hm);
            jarr_i("kf_node_id", kid); jarr_i("kf_node_start", kst); jarr_i("kf_feat", kfe);
            jarr_i("f_node_id", fid); jarr_i("f_node_start", fst); jarr_i("f_feat", ffe); jarr_i("match", match);
            fprintf(JB, "\"kf_desc\": \"");
            for (size_t b = 0; b < kdesc.size(); b++) fprintf(JB, "%02x", kdesc[b]);
            fprintf(JB, "\", \"f_desc\": \"");
            for (size_t b = 0; b < fdesc.size(); b++) fprintf(JB, "%02x", fdesc[b]);
            fprintf(JB, "\"}%s\n", c + 1 < NBC ? "," : "");
        }
        fprintf(JB, "]}\n"); fclose(JB);
    }
    // ------------------------------------------------------------ F: ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (glue)
    // Frame (offsets from so@0x80d00): mbf @0xe0, mb @0xe4, N @0xec, mvKeys @0xf0, mvKeysUn @0x120, mvuRight @0x138, mDescriptors @0x1c8,
    // mvpMapPoints @0x288, mvbOutlier (vector<bool>) @0x2a0, mGrid @0x2c8, mTcw (4x4 CV_32F) @0x122c8, mvScaleFactors @0x12348; statics
    // fx, fy, cx, cy, mnMin/Max*, mfGridElement*Inv.  MapPoint: nObs @0x18, mWorldPos (3x1 CV_32F) @0xd8, mDescriptor @0x1c8.
    {
        path = std::string(outdir) + "/ref_glue_search_last.json";
        FILE *JL = fopen(path.c_str(), "w");
        fprintf(JL, "{\"_doc\": \"ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (so@0x80d00) executed from the reference binary on "
                    "hand-laid Frame / MapPoint objects; cv::Mat algebra (gemm) supplied by oracle/refprobe/probe.cpp. floats as uint32 bit patterns; match[k] = "
                    "last-frame index assigned to current key point k, -1 none, -2 occupied before the call\", \"cases\": [\n");
        // cases 5 and 6: localisation mode (Tracking::UpdateLastFrame creates temporal MapPoints with Observations() == 0, include/Tracking.h:152):
        // `noobs` of the last-frame points have nObs = 0, and the second half of the last frame duplicates the first (same place, near-identical
        // descriptor), so that several last-frame points compete for the same current key points and assignments get overwritten
        struct { int n; float th; int mono, check; float dz; uint64_t seed; float noobs; int dup; } lc[] = {
            {500, 7.0f, 0, 1, 0.02f, 9401, 0.f, 0}, {400, 15.0f, 0, 1, -0.6f, 9402, 0.f, 0}, {450, 7.0f, 0, 0, 0.7f, 9403, 0.f, 0}, {350, 15.0f, 1, 1, 0.5f, 9404, 0.f, 0},
            {480, 15.0f, 0, 1, 0.02f, 9405, 0.5f, 1}, {420, 7.0f, 0, 0, 0.3f, 9406, 1.0f, 1}};
        const int NLC = 6;
        for (int c = 0; c < NLC; c++) {
            rng_seed(lc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int NL = lc[c].n;
            const float fx = 520.9f, fy = 521.0f, cx = 325.1f, cy = 249.7f, bf = 40.0f, mb = bf / fx;
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            // poses: small rotations about y / x, translation mostly along z
            auto pose = [&](float ay, float ax, float tx, float ty, float tz, float *T) {
                const float cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float R[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) T[r * 4 + q] = R[r * 3 + q]; }
                T[3] = tx; T[7] = ty; T[11] = tz; T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
            };
            float *Tl = (float *)bump(64), *Tc = (float *)bump(64);
            pose(0.01f, -0.02f, 0.05f, -0.03f, 0.1f, Tl);
            pose(0.03f, -0.01f, 0.02f, 0.01f, 0.1f - lc[c].dz, Tc);
            // last frame: key points with map points at depth z in the last camera; world = Rl^T (Xl - tl)
            std::vector<cv::KeyPoint> lk(NL);
            std::vector<float> wpos((size_t)NL * 3);
            std::vector<int> lhas(NL), lout(NL), lobs(NL, 1);
            std::vector<float> zs(NL);
            std::vector<uint8_t> mdesc((size_t)NL * 32);
            for (int i = 0; i < NL; i++) {
                lk[i].x = 30.f + uf() * 580.f; lk[i].y = 30.f + uf() * 420.f; lk[i].size = 31.f; lk[i].angle = uf() * 360.f; lk[i].response = 1.f;
                lk[i].octave = (int)rng_below(8); lk[i].class_id = -1;
                float z = 1.0f + uf() * 5.0f;
                if (lc[c].dup && i >= NL / 2) {   // a second point at (almost) the same place
                    const int j = i - NL / 2;
                    lk[i].x = lk[j].x + (uf() - 0.5f) * 3.f; lk[i].y = lk[j].y + (uf() - 0.5f) * 3.f; lk[i].octave = lk[j].octave; lk[i].angle = lk[j].angle;
                    z = zs[j] * (1.f + (uf() - 0.5f) * 0.01f);
                }
                zs[i] = z;
                const float Xl[3] = {(lk[i].x - cx) / fx * z, (lk[i].y - cy) / fy * z, z};
                for (int r = 0; r < 3; r++) wpos[(size_t)i * 3 + r] = Tl[0 * 4 + r] * (Xl[0] - Tl[3]) + Tl[1 * 4 + r] * (Xl[1] - Tl[7]) + Tl[2 * 4 + r] * (Xl[2] - Tl[11]);
                lhas[i] = uf() < 0.85f; lout[i] = uf() < 0.08f;
                for (int b = 0; b < 32; b++) mdesc[(size_t)i * 32 + b] = (uint8_t)rng_below(256);
                if (lc[c].dup && i >= NL / 2) {
                    for (int b = 0; b < 32; b++) mdesc[(size_t)i * 32 + b] = mdesc[(size_t)(i - NL / 2) * 32 + b];
                    const int flips = (int)rng_below(12);
                    for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); mdesc[(size_t)i * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                }
            }
            // current frame: noisy re-observations + clutter
            const int NC = NL + NL / 3;
            std::vector<cv::KeyPoint> ck(NC);
            std::vector<float> cur(NC);
            std::vector<uint8_t> cdesc((size_t)NC * 32);
            for (int k = 0; k < NC; k++) {
                ck[k].size = 31.f; ck[k].response = 1.f; ck[k].class_id = -1;
                if (k < NL) {
                    const float *w = &wpos[(size_t)k * 3];
                    const float xc = Tc[0] * w[0] + Tc[1] * w[1] + Tc[2] * w[2] + Tc[3], yc = Tc[4] * w[0] + Tc[5] * w[1] + Tc[6] * w[2] + Tc[7],
                                zc = Tc[8] * w[0] + Tc[9] * w[1] + Tc[10] * w[2] + Tc[11];
                    ck[k].x = fx * xc / zc + cx + (uf() - 0.5f) * 8.f; ck[k].y = fy * yc / zc + cy + (uf() - 0.5f) * 8.f;
                    ck[k].octave = std::max(0, std::min(7, lk[k].octave + (int)rng_below(3) - 1));
                    float a = lk[k].angle - (uf() < 0.8f ? 10.f + uf() * 9.f : uf() * 360.f);
                    if (a < 0.f) a += 360.f;
                    ck[k].angle = a;
                    for (int b = 0; b < 32; b++) cdesc[(size_t)k * 32 + b] = mdesc[(size_t)k * 32 + b];
                    const int flips = (int)rng_below(70);
                    for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); cdesc[(size_t)k * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                    cur[k] = uf() < 0.6f ? ck[k].x - bf / zc + (uf() - 0.5f) * (uf() < 0.8f ? 4.f : 60.f) : -1.f;
                } else {
                    ck[k].x = uf() * 640.f; ck[k].y = uf() * 480.f; ck[k].octave = (int)rng_below(8); ck[k].angle = uf() * 360.f;
                    for (int b = 0; b < 32; b++) cdesc[(size_t)k * 32 + b] = (uint8_t)rng_below(256);
                    cur[k] = uf() < 0.5f ? ck[k].x - uf() * 30.f : -1.f;
                }
            }
            // objects
            char *lf = (char *)bump(0x12800), *cf = (char *)bump(0x12800); memset(lf, 0, 0x12800); memset(cf, 0, 0x12800);
            char *mps = (char *)bump((size_t)(NL + 2) * 0x300); memset(mps, 0, (size_t)(NL + 2) * 0x300);
            char *occ_obs = mps + (size_t)NL * 0x300, *occ_noobs = mps + (size_t)(NL + 1) * 0x300;
            *(int *)(occ_obs + 0x18) = 3; *(int *)(occ_noobs + 0x18) = 0;
            std::vector<void *> lmp(NL, nullptr), cmp_(NC, nullptr);
            std::vector<uint64_t> obits((NL + 63) / 64 + 1, 0);
            std::vector<int> init(NC, -1);
            for (int i = 0; i < NL; i++) {
                char *o = mps + (size_t)i * 0x300;
                *(int *)(o + 0x18) = 1 + (int)rng_below(3);
                if (lc[c].noobs > 0.f && uf() < lc[c].noobs) { *(int *)(o + 0x18) = 0; lobs[i] = 0; }
                mat_init((cv::Mat *)(o + 0xd8), (unsigned char *)&wpos[(size_t)i * 3], 3, 1, 4);
                ((cv::Mat *)(o + 0xd8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0xd8))->step_buf[1] = 4;
                mat_init((cv::Mat *)(o + 0x1c8), &mdesc[(size_t)i * 32], 1, 32, 32); ((cv::Mat *)(o + 0x1c8))->flags |= 0x4000;
                if (lhas[i]) lmp[i] = o;
                if (lout[i]) obits[i / 64] |= 1ull << (i % 64);
            }
            for (int k = 0; k < NC; k++) { const float u = uf(); if (u < 0.06f) { cmp_[k] = occ_obs; init[k] = -2; } else if (u < 0.09f) cmp_[k] = occ_noobs; }
            auto lay = [&](char *f, std::vector<cv::KeyPoint> &keys, int n, float *T) {
                *(float *)(f + 0xe0) = bf; *(float *)(f + 0xe4) = mb; *(int *)(f + 0xec) = n;
                void **v;
                v = (void **)(f + 0xf0); v[0] = keys.data(); v[1] = keys.data() + n; v[2] = v[1];
                v = (void **)(f + 0x120); v[0] = keys.data(); v[1] = keys.data() + n; v[2] = v[1];
                mat_init((cv::Mat *)(f + 0x122c8), (unsigned char *)T, 4, 4, 16);
                ((cv::Mat *)(f + 0x122c8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(f + 0x122c8))->step_buf[1] = 4;
                v = (void **)(f + 0x12348); v[0] = scale; v[1] = scale + 8; v[2] = scale + 8;
            };
            lay(lf, lk, NL, Tl); lay(cf, ck, NC, Tc);
            void **v;
            v = (void **)(lf + 0x288); v[0] = lmp.data(); v[1] = lmp.data() + NL; v[2] = v[1];
            v = (void **)(lf + 0x2a0); v[0] = obits.data();
            v = (void **)(cf + 0x138); v[0] = cur.data(); v[1] = cur.data() + NC; v[2] = v[1];
            mat_init((cv::Mat *)(cf + 0x1c8), cdesc.data(), NC, 32, 32); ((cv::Mat *)(cf + 0x1c8))->flags |= 0x4000;
            v = (void **)(cf + 0x288); v[0] = cmp_.data(); v[1] = cmp_.data() + NC; v[2] = v[1];
            Frame::fx = fx; Frame::fy = fy; Frame::cx = cx; Frame::cy = cy;
            Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = 640.f; Frame::mnMaxY = 480.f;
            Frame::mfGridElementWidthInv = 64.f / 640.f; Frame::mfGridElementHeightInv = 48.f / 480.f;
            std::vector<std::vector<size_t>> cells(64 * 48);
            for (int k = 0; k < NC; k++) {
                const int gx = (int)roundf((ck[k].x - Frame::mnMinX) * Frame::mfGridElementWidthInv), gy = (int)roundf((ck[k].y - Frame::mnMinY) * Frame::mfGridElementHeightInv);
                if (gx < 0 || gx >= 64 || gy < 0 || gy >= 48) continue;
                cells[gx * 48 + gy].push_back((size_t)k);
            }
            for (int cidx = 0; cidx < 64 * 48; cidx++) {
                v = (void **)(cf + 0x2c8 + (size_t)cidx * 24);
                v[0] = cells[cidx].data(); v[1] = cells[cidx].data() + cells[cidx].size(); v[2] = v[1];
            }
            ORBmatcher *mt = new ORBmatcher(0.9f, lc[c].check != 0);
            const int nm = mt->SearchByProjection(*(Frame *)cf, *(const Frame *)lf, lc[c].th, lc[c].mono != 0);
            std::vector<int> match(NC);
            for (int k = 0; k < NC; k++) {
                const char *q = (const char *)cmp_[k];
                match[k] = q == nullptr ? -1 : q == occ_obs ? -2 : q == occ_noobs ? -3 : (int)((q - mps) / 0x300);
            }
            std::vector<float> kx(NC), ky(NC), ka(NC), la(NL), Tlv(Tl, Tl + 16), Tcv(Tc, Tc + 16), scv(scale, scale + 8), cam = {fx, fy, cx, cy, bf, mb, lc[c].th};
            std::vector<int> ko(NC), lo(NL);
            for (int k = 0; k < NC; k++) { kx[k] = ck[k].x; ky[k] = ck[k].y; ka[k] = ck[k].angle; ko[k] = ck[k].octave; }
            for (int i = 0; i < NL; i++) { la[i] = lk[i].angle; lo[i] = lk[i].octave; }
            fprintf(JL, "{\"n_cur\": %d, \"n_last\": %d, \"mono\": %d, \"check_orientation\": %d, \"nmatches\": %d, ", NC, NL, lc[c].mono, lc[c].check, nm);
            J = JL;
            jarr_f("cam", cam); jarr_f("Tcw", Tcv); jarr_f("Tlw", Tlv); jarr_f("scale", scv);
            jarr_f("x", kx); jarr_f("y", ky); jarr_f("angle", ka); jarr_i("octave", ko); jarr_f("uright", cur); jarr_i("init", init);
            jarr_f("last_angle", la); jarr_i("last_octave", lo); jarr_i("last_has_mp", lhas); jarr_i("last_outlier", lout); jarr_i("last_obs_positive", lobs); jarr_f("world_pos", wpos);
            jarr_i("match", match);
            fprintf(JL, "\"desc\": \"");
            for (size_t b = 0; b < cdesc.size(); b++) fprintf(JL, "%02x", cdesc[b]);
            fprintf(JL, "\", \"mp_desc\": \"");
            for (size_t b = 0; b < mdesc.size(); b++) fprintf(JL, "%02x", mdesc[b]);
            fprintf(JL, "\"}%s\n", c + 1 < NLC ? "," : "");
        }
        fprintf(JL, "]}\n"); fclose(JL);
    }
    // ------------------------------------------------------------ G: ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, matches12) (glue)
    {
        path = std::string(outdir) + "/ref_glue_bow_kf.json";
        FILE *JB = fopen(path.c_str(), "w");
        fprintf(JB, "{\"_doc\": \"ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (so@0x82cc0) executed from the reference binary on hand-laid "
                    "KeyFrame / MapPoint objects. match12[i1] = KF2 feature whose map point KF1 feature i1 received, -1 none\", \"cases\": [\n");
        struct { int n1, n2, nodes; int check; uint64_t seed; } bc[] = {{500, 600, 140, 1, 9501}, {700, 400, 80, 0, 9502}};
        const int NBC = 2;
        typedef std::map<unsigned, std::vector<unsigned>> FeatVec;
        for (int c = 0; c < NBC; c++) {
            rng_seed(bc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int N1 = bc[c].n1, N2 = bc[c].n2, NN = bc[c].nodes;
            std::vector<cv::KeyPoint> k1(N1), k2(N2);
            std::vector<uint8_t> d1((size_t)N1 * 32), d2((size_t)N2 * 32);
            std::vector<int> has1(N1), bad1(N1), has2(N2), bad2(N2);
            std::vector<unsigned> nd1(N1), nd2(N2);
            for (int i = 0; i < N1; i++) {
                k1[i].x = uf() * 640.f; k1[i].y = uf() * 480.f; k1[i].size = 31.f; k1[i].angle = uf() * 360.f; k1[i].response = 1.f; k1[i].octave = 0; k1[i].class_id = -1;
                for (int b = 0; b < 32; b++) d1[(size_t)i * 32 + b] = (uint8_t)rng_below(256);
                has1[i] = uf() < 0.8f; bad1[i] = uf() < 0.05f; nd1[i] = 500u + 3u * rng_below((uint32_t)NN);
            }
            for (int j = 0; j < N2; j++) {
                k2[j].x = uf() * 640.f; k2[j].y = uf() * 480.f; k2[j].size = 31.f; k2[j].response = 1.f; k2[j].octave = 0; k2[j].class_id = -1;
                has2[j] = uf() < 0.8f; bad2[j] = uf() < 0.05f;
                if (uf() < 0.7f) {
                    const int src = (int)rng_below((uint32_t)N1);
                    for (int b = 0; b < 32; b++) d2[(size_t)j * 32 + b] = d1[(size_t)src * 32 + b];
                    const int flips = (int)rng_below(36);
                    for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); d2[(size_t)j * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                    nd2[j] = uf() < 0.9f ? nd1[src] : 500u + 3u * rng_below((uint32_t)NN) + (uf() < 0.3f ? 1u : 0u);
                    float a = k1[src].angle - (uf() < 0.8f ? 40.f + uf() * 8.f : uf() * 360.f);
                    if (a < 0.f) a += 360.f;
                    k2[j].angle = a;
                } else {
                    for (int b = 0; b < 32; b++) d2[(size_t)j * 32 + b] = (uint8_t)rng_below(256);
                    nd2[j] = 500u + 3u * rng_below((uint32_t)NN) + (uf() < 0.3f ? 1u : 0u);
                    k2[j].angle = uf() * 360.f;
                }
            }
            char *kfa = (char *)bump(0x800), *kfb = (char *)bump(0x800); memset(kfa, 0, 0x800); memset(kfb, 0, 0x800);
            char *mp1 = (char *)bump((size_t)N1 * 0x300), *mp2 = (char *)bump((size_t)N2 * 0x300); memset(mp1, 0, (size_t)N1 * 0x300); memset(mp2, 0, (size_t)N2 * 0x300);
            std::vector<void *> m1(N1, nullptr), m2(N2, nullptr);
            for (int i = 0; i < N1; i++) if (has1[i]) { m1[i] = mp1 + (size_t)i * 0x300; *(bool *)(mp1 + (size_t)i * 0x300 + 0x238) = bad1[i] != 0; }
            for (int j = 0; j < N2; j++) if (has2[j]) { m2[j] = mp2 + (size_t)j * 0x300; *(bool *)(mp2 + (size_t)j * 0x300 + 0x238) = bad2[j] != 0; }
            auto layk = [&](char *kf, std::vector<cv::KeyPoint> &keys, std::vector<uint8_t> &desc, std::vector<unsigned> &node, std::vector<void *> &mps, int n) {
                void **v;
                v = (void **)(kf + 0x170); v[0] = keys.data(); v[1] = keys.data() + n; v[2] = v[1];
                mat_init((cv::Mat *)(kf + 0x1b8), desc.data(), n, 32, 32); ((cv::Mat *)(kf + 0x1b8))->flags |= 0x4000;
                FeatVec *fv = new (kf + 0x248) FeatVec();
                for (int i = 0; i < n; i++) (*fv)[node[i]].push_back((unsigned)i);
                v = (void **)(kf + 0x520); v[0] = mps.data(); v[1] = mps.data() + n; v[2] = v[1];
                return fv;
            };
            FeatVec *fv1 = layk(kfa, k1, d1, nd1, m1, N1), *fv2 = layk(kfb, k2, d2, nd2, m2, N2);
            ORBmatcher *mt = new ORBmatcher(0.75f, bc[c].check != 0);
            std::vector<MapPoint *> out;
            const int nm = mt->SearchByBoW((KeyFrame *)kfa, (KeyFrame *)kfb, out);
            std::vector<int> match(N1, -1);
            for (int i = 0; i < N1 && i < (int)out.size(); i++) if (out[i]) match[i] = (int)(((char *)out[i] - mp2) / 0x300);
            std::vector<float> a1(N1), a2(N2); std::vector<int> h1(N1), h2(N2);
            for (int i = 0; i < N1; i++) { a1[i] = k1[i].angle; h1[i] = has1[i] && !bad1[i]; }
            for (int j = 0; j < N2; j++) { a2[j] = k2[j].angle; h2[j] = has2[j] && !bad2[j]; }
            auto flat = [&](FeatVec *fv, std::vector<int> &ids, std::vector<int> &starts, std::vector<int> &feats) {
                for (auto &kv : *fv) { ids.push_back((int)kv.first); starts.push_back((int)feats.size()); for (unsigned x : kv.second) feats.push_back((int)x); }
                starts.push_back((int)feats.size());
            };
            std::vector<int> id1, st1, fe1, id2, st2, fe2;
            flat(fv1, id1, st1, fe1); flat(fv2, id2, st2, fe2);
            fprintf(JB, "{\"n1\": %d, \"n2\": %d, \"nnratio\": 0.75, \"check_orientation\": %d, \"nmatches\": %d, ", N1, N2, bc[c].check, nm);
            J = JB;
            jarr_f("angle1", a1); jarr_f("angle2", a2); jarr_i("has_mp1", h1); jarr_i("has_mp2", h2);
            jarr_i("node_id1", id1); jarr_i("node_start1", st1); jarr_i("feat1", fe1);
            jarr_i("node_id2", id2); jarr_i("node_start2", st2); jarr_i("feat2", fe2); jarr_i("match", match);
            fprintf(JB, "\"desc1\": \"");
            for (size_t b = 0; b < d1.size(); b++) fprintf(JB, "%02x", d1[b]);
            fprintf(JB, "\", \"desc2\": \"");
            for (size_t b = 0; b < d2.size(); b++) fprintf(JB, "%02x", d2[b]);
            fprintf(JB, "\"}%s\n", c + 1 < NBC ? "," : "");
        }
        fprintf(JB, "]}\n"); fclose(JB);
    }
    // ------------------------------------------------------------ H: ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (glue)
    // extra offsets (MapPoint::Get{Min,Max}DistanceInvariance so@0x8fa40/0x8fad0, PredictScale(float, Frame*) so@0x8fc20):
    // MapPoint mfMinDistance @0x248, mfMaxDistance @0x24c; Frame mnScaleLevels @0x12338, mfLogScaleFactor @0x12340.
    {
        path = std::string(outdir) + "/ref_glue_search_reloc.json";
        FILE *JR = fopen(path.c_str(), "w");
        fprintf(JR, "{\"_doc\": \"ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist) (so@0x7e8c0, relocalisation) executed from "
                    "the reference binary on hand-laid objects; cv::Mat algebra / cv::norm supplied by oracle/refprobe/probe.cpp. floats as uint32 bit patterns; "
                    "match[k] = keyframe feature assigned to current key point k, -1 none, -2 occupied before the call\", \"cases\": [\n");
        struct { int n; float th; int orbdist, check; uint64_t seed; } rc[] = {{500, 10.0f, 100, 1, 9601}, {400, 3.0f, 64, 1, 9602}, {450, 10.0f, 100, 0, 9603}};
        const int NRC = 3;
        for (int c = 0; c < NRC; c++) {
            rng_seed(rc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int NK = rc[c].n;
            const float fx = 520.9f, fy = 521.0f, cx = 325.1f, cy = 249.7f;
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            const float logsf = logf(1.2f);
            float *Tc = (float *)bump(64);
            {
                const float ay = 0.04f, ax = -0.03f, cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float R[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) Tc[r * 4 + q] = R[r * 3 + q];
                Tc[3] = 0.1f; Tc[7] = -0.05f; Tc[11] = 0.2f; Tc[12] = Tc[13] = Tc[14] = 0.f; Tc[15] = 1.f;
            }
            float Ow[3];
            for (int i = 0; i < 3; i++) Ow[i] = -(Tc[0 * 4 + i] * Tc[3] + Tc[1 * 4 + i] * Tc[7] + Tc[2 * 4 + i] * Tc[11]);
            std::vector<cv::KeyPoint> kk(NK);
            std::vector<float> wpos((size_t)NK * 3), dmin(NK), dmax(NK);
            std::vector<int> has(NK), bad(NK), found(NK), plev(NK);
            std::vector<uint8_t> mdesc((size_t)NK * 32);
            const int NC = NK + NK / 3;
            std::vector<cv::KeyPoint> ck(NC);
            std::vector<uint8_t> cdesc((size_t)NC * 32);
            for (int i = 0; i < NK; i++) {
                // a point in front of the current camera, placed by back-projecting a pixel of the current image
                const float u0 = (uf() < 0.9f ? 20.f + uf() * 600.f : -60.f + uf() * 760.f), v0 = 20.f + uf() * 440.f, z = 0.8f + uf() * 6.f;
                const float Xc[3] = {(u0 - cx) / fx * z - Tc[3], (v0 - cy) / fy * z - Tc[7], z - Tc[11]};
                for (int r = 0; r < 3; r++) wpos[(size_t)i * 3 + r] = Tc[0 * 4 + r] * Xc[0] + Tc[1 * 4 + r] * Xc[1] + Tc[2 * 4 + r] * Xc[2];
                const float dx = wpos[(size_t)i * 3] - Ow[0], dy = wpos[(size_t)i * 3 + 1] - Ow[1], dz = wpos[(size_t)i * 3 + 2] - Ow[2];
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                plev[i] = (int)rng_below(8);
                dmax[i] = dist * powf(1.2f, (float)plev[i] - 0.5f + uf());
                dmin[i] = dmax[i] / powf(1.2f, 7.f);
                if (uf() < 0.07f) { dmax[i] = dist * 0.5f; dmin[i] = dmax[i] / 4.f; }   // out of the scale-invariance range
                kk[i].x = uf() * 640.f; kk[i].y = uf() * 480.f; kk[i].size = 31.f; kk[i].angle = uf() * 360.f; kk[i].response = 1.f; kk[i].octave = plev[i]; kk[i].class_id = -1;
                has[i] = uf() < 0.85f; bad[i] = uf() < 0.05f; found[i] = uf() < 0.1f;
                for (int b = 0; b < 32; b++) mdesc[(size_t)i * 32 + b] = (uint8_t)rng_below(256);
                // its (noisy) observation in the current frame
                ck[i].size = 31.f; ck[i].response = 1.f; ck[i].class_id = -1;
                ck[i].x = u0 + (uf() - 0.5f) * 6.f; ck[i].y = v0 + (uf() - 0.5f) * 6.f;
                ck[i].octave = std::max(0, std::min(7, plev[i] + (int)rng_below(4) - 1));
                float a = kk[i].angle - (uf() < 0.8f ? 30.f + uf() * 9.f : uf() * 360.f);
                if (a < 0.f) a += 360.f;
                ck[i].angle = a;
                for (int b = 0; b < 32; b++) cdesc[(size_t)i * 32 + b] = mdesc[(size_t)i * 32 + b];
                const int flips = (int)rng_below(90);
                for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); cdesc[(size_t)i * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
            }
            for (int k = NK; k < NC; k++) {
                ck[k].size = 31.f; ck[k].response = 1.f; ck[k].class_id = -1;
                ck[k].x = uf() * 640.f; ck[k].y = uf() * 480.f; ck[k].octave = (int)rng_below(8); ck[k].angle = uf() * 360.f;
                for (int b = 0; b < 32; b++) cdesc[(size_t)k * 32 + b] = (uint8_t)rng_below(256);
            }
            char *kf = (char *)bump(0x800), *cf = (char *)bump(0x12800); memset(kf, 0, 0x800); memset(cf, 0, 0x12800);
            char *mps = (char *)bump((size_t)(NK + 1) * 0x300); memset(mps, 0, (size_t)(NK + 1) * 0x300);
            char *occ = mps + (size_t)NK * 0x300;
            std::vector<void *> kmp(NK, nullptr), cmp_(NC, nullptr);
            std::set<MapPoint *> sfound;
            std::vector<int> init(NC, -1);
            for (int i = 0; i < NK; i++) {
                char *o = mps + (size_t)i * 0x300;
                mat_init((cv::Mat *)(o + 0xd8), (unsigned char *)&wpos[(size_t)i * 3], 3, 1, 4);
                ((cv::Mat *)(o + 0xd8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0xd8))->step_buf[1] = 4;
                mat_init((cv::Mat *)(o + 0x1c8), &mdesc[(size_t)i * 32], 1, 32, 32); ((cv::Mat *)(o + 0x1c8))->flags |= 0x4000;
                *(bool *)(o + 0x238) = bad[i] != 0; *(float *)(o + 0x248) = dmin[i]; *(float *)(o + 0x24c) = dmax[i];
                if (has[i]) { kmp[i] = o; if (found[i]) sfound.insert((MapPoint *)o); }
            }
            for (int k = 0; k < NC; k++) if (uf() < 0.08f) { cmp_[k] = occ; init[k] = -2; }
            void **v;
            v = (void **)(kf + 0x170); v[0] = kk.data(); v[1] = kk.data() + NK; v[2] = v[1];
            v = (void **)(kf + 0x520); v[0] = kmp.data(); v[1] = kmp.data() + NK; v[2] = v[1];
            *(int *)(cf + 0xec) = NC;
            v = (void **)(cf + 0xf0); v[0] = ck.data(); v[1] = ck.data() + NC; v[2] = v[1];
            v = (void **)(cf + 0x120); v[0] = ck.data(); v[1] = ck.data() + NC; v[2] = v[1];
            mat_init((cv::Mat *)(cf + 0x1c8), cdesc.data(), NC, 32, 32); ((cv::Mat *)(cf + 0x1c8))->flags |= 0x4000;
            v = (void **)(cf + 0x288); v[0] = cmp_.data(); v[1] = cmp_.data() + NC; v[2] = v[1];
            mat_init((cv::Mat *)(cf + 0x122c8), (unsigned char *)Tc, 4, 4, 16);
            ((cv::Mat *)(cf + 0x122c8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(cf + 0x122c8))->step_buf[1] = 4;
            *(int *)(cf + 0x12338) = 8; *(float *)(cf + 0x12340) = logsf;
            v = (void **)(cf + 0x12348); v[0] = scale; v[1] = scale + 8; v[2] = scale + 8;
            Frame::fx = fx; Frame::fy = fy; Frame::cx = cx; Frame::cy = cy;
            Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = 640.f; Frame::mnMaxY = 480.f;
            Frame::mfGridElementWidthInv = 64.f / 640.f; Frame::mfGridElementHeightInv = 48.f / 480.f;
            std::vector<std::vector<size_t>> cells(64 * 48);
            for (int k = 0; k < NC; k++) {
                const int gx = (int)roundf((ck[k].x - Frame::mnMinX) * Frame::mfGridElementWidthInv), gy = (int)roundf((ck[k].y - Frame::mnMinY) * Frame::mfGridElementHeightInv);
                if (gx < 0 || gx >= 64 || gy < 0 || gy >= 48) continue;
                cells[gx * 48 + gy].push_back((size_t)k);
            }
            for (int cidx = 0; cidx < 64 * 48; cidx++) {
                v = (void **)(cf + 0x2c8 + (size_t)cidx * 24);
                v[0] = cells[cidx].data(); v[1] = cells[cidx].data() + cells[cidx].size(); v[2] = v[1];
            }
            ORBmatcher *mt = new ORBmatcher(0.9f, rc[c].check != 0);
            const int nm = mt->SearchByProjection(*(Frame *)cf, (KeyFrame *)kf, sfound, rc[c].th, rc[c].orbdist);
            std::vector<int> match(NC);
            for (int k = 0; k < NC; k++) { const char *q = (const char *)cmp_[k]; match[k] = q == nullptr ? -1 : q == occ ? -2 : (int)((q - mps) / 0x300); }
            std::vector<float> kx(NC), ky(NC), ka(NC), kfa(NK), Tcv(Tc, Tc + 16), scv(scale, scale + 8), cam = {fx, fy, cx, cy, logsf, rc[c].th};
            std::vector<int> ko(NC), valid(NK);
            for (int k = 0; k < NC; k++) { kx[k] = ck[k].x; ky[k] = ck[k].y; ka[k] = ck[k].angle; ko[k] = ck[k].octave; }
            for (int i = 0; i < NK; i++) { kfa[i] = kk[i].angle; valid[i] = has[i] && !bad[i] && !found[i]; }
            fprintf(JR, "{\"n_cur\": %d, \"n_kf\": %d, \"orbdist\": %d, \"check_orientation\": %d, \"nmatches\": %d, ", NC, NK, rc[c].orbdist, rc[c].check, nm);
            J = JR;
            jarr_f("cam", cam); jarr_f("Tcw", Tcv); jarr_f("scale", scv);
            jarr_f("x", kx); jarr_f("y", ky); jarr_f("angle", ka); jarr_i("octave", ko); jarr_i("init", init);
            jarr_f("kf_angle", kfa); jarr_i("valid", valid); jarr_f("world_pos", wpos); jarr_f("min_dist", dmin); jarr_f("max_dist", dmax);
            jarr_i("match", match);
            fprintf(JR, "\"desc\": \"");
            for (size_t b = 0; b < cdesc.size(); b++) fprintf(JR, "%02x", cdesc[b]);
            fprintf(JR, "\", \"mp_desc\": \"");
            for (size_t b = 0; b < mdesc.size(); b++) fprintf(JR, "%02x", mdesc[b]);
            fprintf(JR, "\"}%s\n", c + 1 < NRC ? "," : "");
        }
        fprintf(JR, "]}\n"); fclose(JR);
    }
    // ------------------------------------------------------------ I: Frame::isInFrustum(MapPoint*, viewingCosLimit) (glue)
    // Frame (so@0xf5190): mbf @0xe0, mRcw @0x123a8, mtcw @0x12408, mOw @0x124c8 (CV_32F), mnScaleLevels @0x12338, mfLogScaleFactor @0x12340;
    // MapPoint: mWorldPos @0xd8, mNormalVector @0x168 (GetNormal so@0x917b0), mfMinDistance/mfMaxDistance @0x248/0x24c; results written to
    // mTrackProjX/Y/XR @0x1c/0x20/0x24, mnTrackScaleLevel @0x28, mTrackViewCos @0x2c, mbTrackInView @0x30.
    {
        path = std::string(outdir) + "/ref_glue_frustum.json";
        FILE *JF = fopen(path.c_str(), "w");
        fprintf(JF, "{\"_doc\": \"Frame::isInFrustum(MapPoint*, float) (so@0xf5190) executed from the reference binary for every map point of a random local map; "
                    "cv::Mat algebra, cv::norm and Mat::dot supplied by oracle/refprobe/probe.cpp. floats as uint32 bit patterns\", \"cases\": [\n");
        const int NFC = 2;
        for (int c = 0; c < NFC; c++) {
            rng_seed(9701 + c);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int M = 3000;
            const float fx = 517.3f, fy = 516.5f, cx = 318.6f, cy = 255.3f, bf = 40.0f, coslim = c == 0 ? 0.5f : 0.9f;
            const float logsf = logf(1.2f);
            float *R = (float *)bump(36), *t = (float *)bump(12), *Ow = (float *)bump(12);
            {
                const float ay = 0.2f - 0.3f * c, ax = 0.1f, cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float Rr[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                memcpy(R, Rr, 36); t[0] = 0.3f; t[1] = -0.1f; t[2] = 0.5f;
                for (int i = 0; i < 3; i++) Ow[i] = (float)(-((double)R[0 * 3 + i] * t[0] + (double)R[1 * 3 + i] * t[1] + (double)R[2 * 3 + i] * t[2]));
            }
            char *fr = (char *)bump(0x12800); memset(fr, 0, 0x12800);
            auto mat32 = [&](char *at, float *data, int rows, int cols) {
                mat_init((cv::Mat *)at, (unsigned char *)data, rows, cols, (size_t)cols * 4);
                ((cv::Mat *)at)->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)at)->step_buf[1] = 4;
            };
            mat32(fr + 0x123a8, R, 3, 3); mat32(fr + 0x12408, t, 3, 1); mat32(fr + 0x124c8, Ow, 3, 1);
            *(float *)(fr + 0xe0) = bf; *(int *)(fr + 0x12338) = 8; *(float *)(fr + 0x12340) = logsf;
            Frame::fx = fx; Frame::fy = fy; Frame::cx = cx; Frame::cy = cy;
            Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = 640.f; Frame::mnMaxY = 480.f;
            std::vector<float> wpos((size_t)M * 3), nrm((size_t)M * 3), dmin(M), dmax(M);
            char *mps = (char *)bump((size_t)M * 0x300); memset(mps, 0, (size_t)M * 0x300);
            std::vector<int> inview(M), level(M);
            std::vector<float> px(M), py(M), pxr(M), vcos(M);
            for (int i = 0; i < M; i++) {
                for (int r = 0; r < 3; r++) wpos[(size_t)i * 3 + r] = (uf() - 0.5f) * (r == 2 ? 16.f : 10.f) + (r == 2 ? 3.f : 0.f);
                float nx = uf() - 0.5f, ny = uf() - 0.5f, nz = uf() - 0.5f;
                if (uf() < 0.7f) { nx = wpos[(size_t)i * 3] - Ow[0] + (uf() - 0.5f); ny = wpos[(size_t)i * 3 + 1] - Ow[1] + (uf() - 0.5f); nz = wpos[(size_t)i * 3 + 2] - Ow[2] + (uf() - 0.5f); }
                const float nl = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-6f;
                nrm[(size_t)i * 3] = nx / nl; nrm[(size_t)i * 3 + 1] = ny / nl; nrm[(size_t)i * 3 + 2] = nz / nl;
                dmax[i] = 1.f + uf() * 14.f; dmin[i] = dmax[i] / (2.f + uf() * 3.f);
                char *o = mps + (size_t)i * 0x300;
                mat32(o + 0xd8, &wpos[(size_t)i * 3], 3, 1); mat32(o + 0x168, &nrm[(size_t)i * 3], 3, 1);
                *(float *)(o + 0x248) = dmin[i]; *(float *)(o + 0x24c) = dmax[i];
                *(bool *)(o + 0x30) = true;   // must be cleared by the call
                const bool in = ((Frame *)fr)->isInFrustum((MapPoint *)o, coslim);
                inview[i] = in ? 1 : 0;
                if ((bool)*(bool *)(o + 0x30) != in) { fprintf(stderr, "refprobe: mbTrackInView disagrees with the return value\n"); abort(); }
                px[i] = *(float *)(o + 0x1c); py[i] = *(float *)(o + 0x20); pxr[i] = *(float *)(o + 0x24); level[i] = *(int *)(o + 0x28); vcos[i] = *(float *)(o + 0x2c);
            }
            std::vector<float> cam = {fx, fy, cx, cy, bf, logsf, coslim}, Rv(R, R + 9), tv(t, t + 3), Owv(Ow, Ow + 3);
            fprintf(JF, "{\"m\": %d, ", M);
            J = JF;
            jarr_f("cam", cam); jarr_f("Rcw", Rv); jarr_f("tcw", tv); jarr_f("Ow", Owv); jarr_f("world_pos", wpos); jarr_f("normal", nrm); jarr_f("min_dist", dmin);
            jarr_f("max_dist", dmax); jarr_i("in_view", inview); jarr_f("proj_x", px); jarr_f("proj_y", py); jarr_f("proj_xr", pxr); jarr_i("level", level);
            jarr_f("view_cos", vcos, true);
            fprintf(JF, "}%s\n", c + 1 < NFC ? "," : "");
        }
        fprintf(JF, "]}\n"); fclose(JF);
    }
    // ------------------------------------------------------------ J: Frame::ComputeStereoFromRGBD(const cv::Mat&) (so@0xf6860)
    // N @0xec, mvKeys @0xf0, mvKeysUn @0x120, mvuRight @0x138, mvDepth @0x150, mbf @0xe0; no OpenCV code is reached.
    {
        path = std::string(outdir) + "/ref_glue_stereo.json";
        FILE *JT = fopen(path.c_str(), "w");
        fprintf(JT, "{\"_doc\": \"Frame::ComputeStereoFromRGBD (so@0xf6860) executed from the reference binary: mvuRight / mvDepth for random key points and a random "
                    "depth image (LCG, see refgen.py: depth_image). floats as uint32 bit patterns\", \"cases\": [\n");
        const int W = 640, H = 480, N = 1200;
        rng_seed(9801);
        auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
        std::vector<float> depth((size_t)W * H);
        for (size_t i = 0; i < depth.size(); i++) { const float u = uf(); depth[i] = u < 0.08f ? 0.f : (u < 0.1f ? -1.f : 0.4f + uf() * 6.f); }
        std::vector<cv::KeyPoint> keys(N), keysun(N);
        for (int i = 0; i < N; i++) {
            keys[i].x = uf() * 639.99f; keys[i].y = uf() * 479.99f; keys[i].size = 31.f; keys[i].angle = 0.f; keys[i].response = 1.f; keys[i].octave = 0; keys[i].class_id = -1;
            keysun[i] = keys[i]; keysun[i].x += (uf() - 0.5f) * 3.f; keysun[i].y += (uf() - 0.5f) * 3.f;
        }
        char *fr = (char *)bump(0x12800); memset(fr, 0, 0x12800);
        *(float *)(fr + 0xe0) = 40.0f; *(int *)(fr + 0xec) = N;
        void **v;
        v = (void **)(fr + 0xf0); v[0] = keys.data(); v[1] = keys.data() + N; v[2] = v[1];
        v = (void **)(fr + 0x120); v[0] = keysun.data(); v[1] = keysun.data() + N; v[2] = v[1];
        cv::Mat dm;
        mat_init(&dm, (unsigned char *)depth.data(), H, W, (size_t)W * 4);
        dm.flags = 0x42FF0000 | 0x4000 | 5; dm.step_buf[1] = 4;
        ((Frame *)fr)->ComputeStereoFromRGBD(dm);
        float **ur = (float **)(fr + 0x138), **dp = (float **)(fr + 0x150);
        if (ur[1] - ur[0] != N || dp[1] - dp[0] != N) { fprintf(stderr, "refprobe: unexpected vector sizes\n"); abort(); }
        std::vector<float> kx(N), ky(N), kux(N), urv(ur[0], ur[0] + N), dpv(dp[0], dp[0] + N);
        for (int i = 0; i < N; i++) { kx[i] = keys[i].x; ky[i] = keys[i].y; kux[i] = keysun[i].x; }
        fprintf(JT, "{\"w\": %d, \"h\": %d, \"n\": %d, \"bf_bits\": %u, \"seed\": 9801, ", W, H, N, 0x42200000u);
        J = JT;
        jarr_f("x", kx); jarr_f("y", ky); jarr_f("x_un", kux); jarr_f("uright", urv); jarr_f("depth_of_kp", dpv, true);
        fprintf(JT, "}\n]}\n"); fclose(JT);
    }
    // ------------------------------------------------------------ K: ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (so@0x7a500, glue)
    // KeyFrame offsets (KeyFrame::GetFeaturesInArea so@0x96fe0, IsInImage so@0x97480, GetRotation/GetTranslation/GetCameraCenter so@0x97be0/0x97f30/0x97900,
    // MapPoint::PredictScale(float, KeyFrame*) so@0x8fb60): mnGridCols/Rows @0x18/0x1c, mfGridElementWidthInv/HeightInv @0x20/0x24, fx, fy, cx, cy @0x130..0x13c,
    // mbf @0x148, N @0x154, mvKeysUn @0x170, mvuRight @0x188, mDescriptors @0x1b8, mnScaleLevels @0x2d8, mfLogScaleFactor @0x2e0, mvScaleFactors @0x2e8,
    // mvInvLevelSigma2 @0x318, mnMinX/MinY/MaxX/MaxY (int) @0x330..0x33c, Tcw @0x3a0, Ow @0x460, mvpMapPoints @0x520, mGrid @0x548, mMutexPose @0x640,
    // mMutexFeatures @0x690.  MapPoint: mObservations (std::map<KeyFrame*, size_t>) @0x138 (IsInKeyFrame so@0x8f970).
    // MapPoint::AddObservation / Replace are the recorders defined above; KeyFrame::GetMapPoint / AddMapPoint run from the binary.
    {
        path = std::string(outdir) + "/ref_glue_fuse.json";
        FILE *JK = fopen(path.c_str(), "w");
        fprintf(JK, "{\"_doc\": \"ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (so@0x7a500) executed from the reference binary on hand-laid objects; "
                    "cv::Mat algebra / cv::norm / Mat::dot supplied by oracle/refprobe/probe.cpp, MapPoint::AddObservation / Replace replaced by recorders. floats as "
                    "uint32 bit patterns; best_idx[i] = keyframe key point map point i was fused with (-1 none)\", \"cases\": [\n");
        struct { int nk, m; float th; uint64_t seed; } kc[] = {{900, 700, 3.0f, 9901}, {600, 800, 4.0f, 9902}, {1000, 500, 2.5f, 9903}};
        const int NKC = 3;
        for (int c = 0; c < NKC; c++) {
            rng_seed(kc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int NK = kc[c].nk, M = kc[c].m;
            const float fx = 517.3f, fy = 516.5f, cx = 318.6f, cy = 255.3f, bf = 40.0f;
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            const float logsf = logf(1.2f);
            float *Tc = (float *)bump(64), *Owp = (float *)bump(16);
            {
                const float ay = -0.05f + 0.02f * c, ax = 0.03f, cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float R[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) Tc[r * 4 + q] = R[r * 3 + q];
                Tc[3] = -0.2f; Tc[7] = 0.07f; Tc[11] = 0.15f; Tc[12] = Tc[13] = Tc[14] = 0.f; Tc[15] = 1.f;
            }
            for (int i = 0; i < 3; i++) Owp[i] = -(Tc[0 * 4 + i] * Tc[3] + Tc[1 * 4 + i] * Tc[7] + Tc[2 * 4 + i] * Tc[11]);
            std::vector<cv::KeyPoint> kk(NK);
            std::vector<float> kur(NK), kz(NK);
            std::vector<uint8_t> kdesc((size_t)NK * 32);
            for (int k = 0; k < NK; k++) {
                kk[k].x = uf() * 640.f; kk[k].y = uf() * 480.f; kk[k].size = 31.f; kk[k].angle = uf() * 360.f; kk[k].response = 1.f; kk[k].octave = (int)rng_below(8); kk[k].class_id = -1;
                kz[k] = 0.6f + uf() * 7.f;
                kur[k] = uf() < 0.7f ? kk[k].x - bf / kz[k] + (uf() - 0.5f) * 1.5f : -1.f;
                for (int b = 0; b < 32; b++) kdesc[(size_t)k * 32 + b] = (uint8_t)rng_below(256);
            }
            std::vector<float> wpos((size_t)M * 3), nrm((size_t)M * 3), dmin(M), dmax(M);
            std::vector<int> bad(M), inkf(M), nobs(M);
            std::vector<uint8_t> mdesc((size_t)M * 32);
            for (int i = 0; i < M; i++) {
                // most points re-observe a key point of the keyframe (a few of them the same one), the rest fall anywhere (some outside / behind)
                const int src = (int)rng_below(NK);
                const bool tied = uf() < 0.85f;
                float u0 = tied ? kk[src].x + (uf() - 0.5f) * 5.f * scale[kk[src].octave] : -80.f + uf() * 800.f;
                float v0 = tied ? kk[src].y + (uf() - 0.5f) * 5.f * scale[kk[src].octave] : -60.f + uf() * 600.f;
                float z = tied ? kz[src] * (1.f + (uf() - 0.5f) * 0.02f) : 0.5f + uf() * 8.f;
                if (uf() < 0.03f) z = -z;
                const float Xc[3] = {(u0 - cx) / fx * z - Tc[3], (v0 - cy) / fy * z - Tc[7], z - Tc[11]};
                for (int r = 0; r < 3; r++) wpos[(size_t)i * 3 + r] = Tc[0 * 4 + r] * Xc[0] + Tc[1 * 4 + r] * Xc[1] + Tc[2 * 4 + r] * Xc[2];
                const float dx = wpos[(size_t)i * 3] - Owp[0], dy = wpos[(size_t)i * 3 + 1] - Owp[1], dz = wpos[(size_t)i * 3 + 2] - Owp[2];
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                // viewing direction: around the ray from the camera, now and then more than 60 degrees off
                float nv[3] = {dx / dist + (uf() - 0.5f) * 0.6f, dy / dist + (uf() - 0.5f) * 0.6f, dz / dist + (uf() - 0.5f) * 0.6f};
                if (uf() < 0.08f) { nv[0] = -nv[0]; nv[2] = -nv[2]; }
                if (uf() < 0.1f) { nv[0] += 1.5f; }
                const float nn = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                for (int r = 0; r < 3; r++) nrm[(size_t)i * 3 + r] = nv[r] / nn;
                const int plev = tied ? kk[src].octave : (int)rng_below(8);
                dmax[i] = dist * powf(1.2f, (float)plev + (uf() < 0.5f ? 0.f : 1.f) - 0.5f + (uf() - 0.5f) * 0.9f);
                dmin[i] = dmax[i] / powf(1.2f, 7.f);
                if (uf() < 0.05f) { dmax[i] = dist * 0.6f; dmin[i] = dmax[i] / 4.f; }
                bad[i] = uf() < 0.04f; inkf[i] = uf() < 0.06f; nobs[i] = 1 + (int)rng_below(6);
                for (int b = 0; b < 32; b++) mdesc[(size_t)i * 32 + b] = tied ? kdesc[(size_t)src * 32 + b] : (uint8_t)rng_below(256);
                const int flips = (int)rng_below(80);
                for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); mdesc[(size_t)i * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
            }
            char *kf = (char *)bump(0x800); memset(kf, 0, 0x800);
            const size_t MPS = 0x300;
            char *mps = (char *)bump((size_t)(M + NK) * MPS); memset(mps, 0, (size_t)(M + NK) * MPS);
            char *occ = mps + (size_t)M * MPS;   // occupant k = the map point key point k already holds (30 % of them)
            std::vector<MapPoint *> list(M, nullptr);
            std::vector<void *> kmp(NK, nullptr);
            std::vector<int> isnull(M);
            for (int i = 0; i < M; i++) {
                char *o = mps + (size_t)i * MPS;
                *(int *)(o + 0x18) = nobs[i];
                new (o + 0x138) std::map<KeyFrame *, size_t>();
                if (inkf[i]) (*(std::map<KeyFrame *, size_t> *)(o + 0x138))[(KeyFrame *)kf] = 0;
                mat_init((cv::Mat *)(o + 0xd8), (unsigned char *)&wpos[(size_t)i * 3], 3, 1, 4);
                ((cv::Mat *)(o + 0xd8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0xd8))->step_buf[1] = 4;
                mat_init((cv::Mat *)(o + 0x168), (unsigned char *)&nrm[(size_t)i * 3], 3, 1, 4);
                ((cv::Mat *)(o + 0x168))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0x168))->step_buf[1] = 4;
                mat_init((cv::Mat *)(o + 0x1c8), &mdesc[(size_t)i * 32], 1, 32, 32); ((cv::Mat *)(o + 0x1c8))->flags |= 0x4000;
                *(bool *)(o + 0x238) = bad[i] != 0; *(float *)(o + 0x248) = dmin[i]; *(float *)(o + 0x24c) = dmax[i];
                isnull[i] = uf() < 0.02f;
                if (!isnull[i]) list[i] = (MapPoint *)o;
            }
            for (int k = 0; k < NK; k++) {
                char *o = occ + (size_t)k * MPS;
                *(int *)(o + 0x18) = 1 + (int)rng_below(6);
                new (o + 0x138) std::map<KeyFrame *, size_t>();
                if (uf() < 0.3f) kmp[k] = o;
            }
            *(int *)(kf + 0x18) = 64; *(int *)(kf + 0x1c) = 48; *(float *)(kf + 0x20) = 64.f / 640.f; *(float *)(kf + 0x24) = 48.f / 480.f;
            *(float *)(kf + 0x130) = fx; *(float *)(kf + 0x134) = fy; *(float *)(kf + 0x138) = cx; *(float *)(kf + 0x13c) = cy; *(float *)(kf + 0x148) = bf;
            *(int *)(kf + 0x154) = NK;
            void **v;
            v = (void **)(kf + 0x170); v[0] = kk.data(); v[1] = kk.data() + NK; v[2] = v[1];
            v = (void **)(kf + 0x188); v[0] = kur.data(); v[1] = kur.data() + NK; v[2] = v[1];
            mat_init((cv::Mat *)(kf + 0x1b8), kdesc.data(), NK, 32, 32); ((cv::Mat *)(kf + 0x1b8))->flags |= 0x4000;
            *(int *)(kf + 0x2d8) = 8; *(float *)(kf + 0x2e0) = logsf;
            v = (void **)(kf + 0x2e8); v[0] = scale; v[1] = scale + 8; v[2] = v[1];
            v = (void **)(kf + 0x318); v[0] = is2; v[1] = is2 + 8; v[2] = v[1];
            *(int *)(kf + 0x330) = 0; *(int *)(kf + 0x334) = 0; *(int *)(kf + 0x338) = 640; *(int *)(kf + 0x33c) = 480;
            mat_init((cv::Mat *)(kf + 0x3a0), (unsigned char *)Tc, 4, 4, 16);
            ((cv::Mat *)(kf + 0x3a0))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(kf + 0x3a0))->step_buf[1] = 4;
            mat_init((cv::Mat *)(kf + 0x460), (unsigned char *)Owp, 3, 1, 4);
            ((cv::Mat *)(kf + 0x460))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(kf + 0x460))->step_buf[1] = 4;
            v = (void **)(kf + 0x520); v[0] = kmp.data(); v[1] = kmp.data() + NK; v[2] = v[1];
            std::vector<std::vector<std::vector<size_t>>> G(64, std::vector<std::vector<size_t>>(48));
            for (int k = 0; k < NK; k++) {
                const int gx = (int)roundf(kk[k].x * (64.f / 640.f)), gy = (int)roundf(kk[k].y * (48.f / 480.f));
                if (gx < 0 || gx >= 64 || gy < 0 || gy >= 48) continue;
                G[gx][gy].push_back((size_t)k);
            }
            memcpy(kf + 0x548, (void *)&G, sizeof(G));
            g_fuse_events.clear();
            ORBmatcher *mt = new ORBmatcher(0.6f, true);
            const int nf = mt->Fuse((KeyFrame *)kf, list, kc[c].th);
            // decode: AddObservation names the key point directly; a Replace pairs the current point with the holder of its key point
            std::vector<int> best(M, -1), valid(M);
            auto mp_index = [&](void *q) { return (int)(((char *)q - mps) / MPS); };
            for (const FuseEvent &e : g_fuse_events) {
                if (e.kind == 0) { best[mp_index(e.a)] = (int)e.idx; continue; }
                const int ia = mp_index(e.a), ib = mp_index(e.b);
                // one of the two is a list point without a key point yet (the one being fused), the other holds the key point
                auto holder_kp = [&](int q) { return q >= M ? q - M : best[q]; };
                const int cur = (ia < M && best[ia] < 0) ? ia : ib, oth = cur == ia ? ib : ia;
                if (cur >= M || holder_kp(oth) < 0) { fprintf(stderr, "refprobe: cannot decode a Replace event\n"); abort(); }
                best[cur] = holder_kp(oth);
            }
            int cnt = 0;
            for (int i = 0; i < M; i++) { valid[i] = !isnull[i] && !bad[i] && !inkf[i]; cnt += best[i] >= 0; }
            if (cnt != nf) { fprintf(stderr, "refprobe: %d fused points decoded, the call returned %d\n", cnt, nf); abort(); }
            std::vector<float> kx(NK), ky(NK), Tcv(Tc, Tc + 16), Owv(Owp, Owp + 3), scv(scale, scale + 8), isv(is2, is2 + 8), cam = {fx, fy, cx, cy, bf, logsf, kc[c].th};
            std::vector<int> ko(NK);
            for (int k = 0; k < NK; k++) { kx[k] = kk[k].x; ky[k] = kk[k].y; ko[k] = kk[k].octave; }
            fprintf(JK, "{\"n_kf\": %d, \"m\": %d, \"nfused\": %d, ", NK, M, nf);
            J = JK;
            jarr_f("cam", cam); jarr_f("Tcw", Tcv); jarr_f("Ow", Owv); jarr_f("scale", scv); jarr_f("inv_sigma2", isv);
            jarr_f("x", kx); jarr_f("y", ky); jarr_i("octave", ko); jarr_f("uright", kur);
            jarr_f("world_pos", wpos); jarr_f("normal", nrm); jarr_f("min_dist", dmin); jarr_f("max_dist", dmax); jarr_i("valid", valid);
            jarr_i("best_idx", best);
            fprintf(JK, "\"desc\": \"");
            for (size_t b = 0; b < kdesc.size(); b++) fprintf(JK, "%02x", kdesc[b]);
            fprintf(JK, "\", \"mp_desc\": \"");
            for (size_t b = 0; b < mdesc.size(); b++) fprintf(JK, "%02x", mdesc[b]);
            fprintf(JK, "\"}%s\n", c + 1 < NKC ? "," : "");
        }
        fprintf(JK, "]}\n"); fclose(JK);
    }
    // ------------------------------------------------------------ L, M: the two Scw overloads of loop closing (glue)
    //   L  ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float th, vector<MapPoint*>& vpReplacePoint)   so@0x7bb20
    //   M  ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>& vpMatched, int th)   so@0x880f0
    // Same hand-laid KeyFrame / MapPoint objects as tier K; the Sim3 algebra (operator/, .t(), unary minus, gemm) is the restatement above.
    for (int tierLM = 0; tierLM < 2; tierLM++) {
        path = std::string(outdir) + (tierLM == 0 ? "/ref_glue_fuse_sim3.json" : "/ref_glue_search_sim3.json");
        FILE *JL = fopen(path.c_str(), "w");
        if (tierLM == 0)
            fprintf(JL, "{\"_doc\": \"ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, th, vpReplacePoint) (so@0x7bb20) executed from the reference binary on "
                        "hand-laid objects; cv::Mat algebra supplied by oracle/refprobe/probe.cpp, MapPoint::AddObservation replaced by a recorder. floats as uint32 bit "
                        "patterns; best_idx[i] = keyframe key point chosen for map point i (-1 none)\", \"cases\": [\n");
        else
            fprintf(JL, "{\"_doc\": \"ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vpMatched, int th) (so@0x880f0) executed from the reference "
                        "binary on hand-laid objects. floats as uint32 bit patterns; init[k] = -2 where vpMatched[k] was occupied on entry; match[k] = map point stored at key "
                        "point k by the call (-1 none, -2 occupied before)\", \"cases\": [\n");
        struct { int nk, m; float th; float s; uint64_t seed; } lc[] = {{900, 700, 4.0f, 1.07f, 10001}, {700, 900, 3.0f, 0.93f, 10002}, {1000, 600, 10.0f, 1.0f, 10003}};
        const int NLC = 3;
        for (int c = 0; c < NLC; c++) {
            rng_seed(lc[c].seed + 100 * tierLM);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int NK = lc[c].nk, M = lc[c].m;
            const float fx = 517.3f, fy = 516.5f, cx = 318.6f, cy = 255.3f, bf = 40.0f, ssc = lc[c].s;
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            const float logsf = logf(1.2f);
            float *Sc = (float *)bump(64);
            float Rm[9], tv[3], Owp[3];
            {
                const float ay = 0.06f - 0.03f * c, ax = -0.02f, cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float R[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                memcpy(Rm, R, sizeof(R)); tv[0] = 0.3f; tv[1] = -0.04f; tv[2] = -0.1f;
                for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) Sc[r * 4 + q] = ssc * R[r * 3 + q]; Sc[r * 4 + 3] = ssc * tv[r]; }
                Sc[12] = Sc[13] = Sc[14] = 0.f; Sc[15] = 1.f;
            }
            for (int i = 0; i < 3; i++) Owp[i] = -(Rm[0 * 3 + i] * tv[0] + Rm[1 * 3 + i] * tv[1] + Rm[2 * 3 + i] * tv[2]);
            std::vector<cv::KeyPoint> kk(NK);
            std::vector<float> kur(NK), kz(NK);
            std::vector<uint8_t> kdesc((size_t)NK * 32);
            for (int k = 0; k < NK; k++) {
                kk[k].x = uf() * 640.f; kk[k].y = uf() * 480.f; kk[k].size = 31.f; kk[k].angle = uf() * 360.f; kk[k].response = 1.f; kk[k].octave = (int)rng_below(8); kk[k].class_id = -1;
                kz[k] = 0.6f + uf() * 7.f; kur[k] = -1.f;
                for (int b = 0; b < 32; b++) kdesc[(size_t)k * 32 + b] = (uint8_t)rng_below(256);
            }
            std::vector<float> wpos((size_t)M * 3), nrm((size_t)M * 3), dmin(M), dmax(M);
            std::vector<int> bad(M), pre(M);
            std::vector<uint8_t> mdesc((size_t)M * 32);
            for (int i = 0; i < M; i++) {
                const int src = (int)rng_below(NK);
                const bool tied = uf() < 0.85f;
                float u0 = tied ? kk[src].x + (uf() - 0.5f) * 6.f * scale[kk[src].octave] : -80.f + uf() * 800.f;
                float v0 = tied ? kk[src].y + (uf() - 0.5f) * 6.f * scale[kk[src].octave] : -60.f + uf() * 600.f;
                float z = tied ? kz[src] : 0.5f + uf() * 8.f;
                if (uf() < 0.03f) z = -z;
                const float Xc[3] = {(u0 - cx) / fx * z - tv[0], (v0 - cy) / fy * z - tv[1], z - tv[2]};
                for (int r = 0; r < 3; r++) wpos[(size_t)i * 3 + r] = Rm[0 * 3 + r] * Xc[0] + Rm[1 * 3 + r] * Xc[1] + Rm[2 * 3 + r] * Xc[2];
                const float dx = wpos[(size_t)i * 3] - Owp[0], dy = wpos[(size_t)i * 3 + 1] - Owp[1], dz = wpos[(size_t)i * 3 + 2] - Owp[2];
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                float nv[3] = {dx / dist + (uf() - 0.5f) * 0.6f, dy / dist + (uf() - 0.5f) * 0.6f, dz / dist + (uf() - 0.5f) * 0.6f};
                if (uf() < 0.08f) { nv[0] = -nv[0]; nv[2] = -nv[2]; }
                if (uf() < 0.1f) { nv[0] += 1.5f; }
                const float nn = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                for (int r = 0; r < 3; r++) nrm[(size_t)i * 3 + r] = nv[r] / nn;
                const int plev = tied ? kk[src].octave : (int)rng_below(8);
                dmax[i] = dist * powf(1.2f, (float)plev + (uf() < 0.5f ? 0.f : 1.f) - 0.5f + (uf() - 0.5f) * 0.9f);
                dmin[i] = dmax[i] / powf(1.2f, 7.f);
                if (uf() < 0.05f) { dmax[i] = dist * 0.6f; dmin[i] = dmax[i] / 4.f; }
                bad[i] = uf() < 0.04f; pre[i] = uf() < 0.05f;   // pre: the point already sits in the keyframe (L) / in vpMatched (M)
                for (int b = 0; b < 32; b++) mdesc[(size_t)i * 32 + b] = tied ? kdesc[(size_t)src * 32 + b] : (uint8_t)rng_below(256);
                const int flips = (int)rng_below(80);
                for (int q = 0; q < flips; q++) { const int bit = (int)rng_below(256); mdesc[(size_t)i * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
            }
            char *kf = (char *)bump(0x800); memset(kf, 0, 0x800);
            const size_t MPS = 0x300;
            char *mps = (char *)bump((size_t)(M + NK) * MPS); memset(mps, 0, (size_t)(M + NK) * MPS);
            char *occ = mps + (size_t)M * MPS;
            std::vector<MapPoint *> list(M, nullptr);
            std::vector<void *> kmp(NK, nullptr);
            for (int i = 0; i < M; i++) {
                char *o = mps + (size_t)i * MPS;
                *(int *)(o + 0x18) = 1 + (int)rng_below(6);
                new (o + 0x138) std::map<KeyFrame *, size_t>();
                mat_init((cv::Mat *)(o + 0xd8), (unsigned char *)&wpos[(size_t)i * 3], 3, 1, 4);
                ((cv::Mat *)(o + 0xd8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0xd8))->step_buf[1] = 4;
                mat_init((cv::Mat *)(o + 0x168), (unsigned char *)&nrm[(size_t)i * 3], 3, 1, 4);
                ((cv::Mat *)(o + 0x168))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0x168))->step_buf[1] = 4;
                mat_init((cv::Mat *)(o + 0x1c8), &mdesc[(size_t)i * 32], 1, 32, 32); ((cv::Mat *)(o + 0x1c8))->flags |= 0x4000;
                *(bool *)(o + 0x238) = bad[i] != 0; *(float *)(o + 0x248) = dmin[i]; *(float *)(o + 0x24c) = dmax[i];
                list[i] = (MapPoint *)o;
            }
            // key points already holding a map point: an occupant object (25 %), or one of the listed points flagged `pre`
            std::vector<int> init(NK, -1);
            for (int k = 0; k < NK; k++) {
                char *o = occ + (size_t)k * MPS;
                *(int *)(o + 0x18) = 1 + (int)rng_below(6);
                new (o + 0x138) std::map<KeyFrame *, size_t>();
                if (uf() < 0.25f) { kmp[k] = o; init[k] = -2; }
            }
            for (int i = 0; i < M; i++) if (pre[i]) { const int k = (int)rng_below(NK); if (kmp[k] == nullptr && !bad[i]) {   // (a bad holder would make the choice unobservable)
                kmp[k] = mps + (size_t)i * MPS; init[k] = -2; } else pre[i] = 0; }
            *(int *)(kf + 0x18) = 64; *(int *)(kf + 0x1c) = 48; *(float *)(kf + 0x20) = 64.f / 640.f; *(float *)(kf + 0x24) = 48.f / 480.f;
            *(float *)(kf + 0x130) = fx; *(float *)(kf + 0x134) = fy; *(float *)(kf + 0x138) = cx; *(float *)(kf + 0x13c) = cy; *(float *)(kf + 0x148) = bf;
            *(int *)(kf + 0x154) = NK;
            void **v;
            v = (void **)(kf + 0x170); v[0] = kk.data(); v[1] = kk.data() + NK; v[2] = v[1];
            v = (void **)(kf + 0x188); v[0] = kur.data(); v[1] = kur.data() + NK; v[2] = v[1];
            mat_init((cv::Mat *)(kf + 0x1b8), kdesc.data(), NK, 32, 32); ((cv::Mat *)(kf + 0x1b8))->flags |= 0x4000;
            *(int *)(kf + 0x2d8) = 8; *(float *)(kf + 0x2e0) = logsf;
            v = (void **)(kf + 0x2e8); v[0] = scale; v[1] = scale + 8; v[2] = v[1];
            v = (void **)(kf + 0x318); v[0] = is2; v[1] = is2 + 8; v[2] = v[1];
            *(int *)(kf + 0x330) = 0; *(int *)(kf + 0x334) = 0; *(int *)(kf + 0x338) = 640; *(int *)(kf + 0x33c) = 480;
            v = (void **)(kf + 0x520); v[0] = kmp.data(); v[1] = kmp.data() + NK; v[2] = v[1];
            std::vector<std::vector<std::vector<size_t>>> G(64, std::vector<std::vector<size_t>>(48));
            for (int k = 0; k < NK; k++) {
                const int gx = (int)roundf(kk[k].x * (64.f / 640.f)), gy = (int)roundf(kk[k].y * (48.f / 480.f));
                if (gx < 0 || gx >= 64 || gy < 0 || gy >= 48) continue;
                G[gx][gy].push_back((size_t)k);
            }
            memcpy(kf + 0x548, (void *)&G, sizeof(G));
            cv::Mat Scw;
            mat_init(&Scw, (unsigned char *)Sc, 4, 4, 16);
            Scw.flags = 0x42FF0000 | 0x4000 | 5; Scw.step_buf[1] = 4;
            ORBmatcher *mt = new ORBmatcher(0.75f, true);
            auto mp_index = [&](void *q) { return (int)(((char *)q - mps) / MPS); };
            std::vector<int> valid(M), out;
            int ret = 0;
            if (tierLM == 0) {
                g_fuse_events.clear();
                std::vector<MapPoint *> repl(M, nullptr);
                ret = mt->Fuse((KeyFrame *)kf, Scw, list, lc[c].th, repl);
                std::vector<int> best(M, -1);
                for (const FuseEvent &e : g_fuse_events) {
                    if (e.kind != 0) { fprintf(stderr, "refprobe: unexpected Replace in Fuse(Scw)\n"); abort(); }
                    best[mp_index(e.a)] = (int)e.idx;
                }
                // a replace candidate names the holder of the chosen key point: an occupant (its own index), a `pre` point or a point added earlier in this call
                std::vector<int> kp_of_pre(M, -1);
                for (int k = 0; k < NK; k++) if (kmp[k] && mp_index(kmp[k]) < M && init[k] == -2) kp_of_pre[mp_index(kmp[k])] = k;
                for (int i = 0; i < M; i++) if (repl[i]) {
                    const int q = mp_index(repl[i]);
                    const int k = q >= M ? q - M : (kp_of_pre[q] >= 0 ? kp_of_pre[q] : best[q]);
                    if (k < 0 || best[i] >= 0) { fprintf(stderr, "refprobe: cannot decode vpReplacePoint\n"); abort(); }
                    best[i] = k;
                }
                int cnt = 0;
                for (int i = 0; i < M; i++) { valid[i] = !bad[i] && !pre[i]; cnt += best[i] >= 0; }
                if (cnt != ret) { fprintf(stderr, "refprobe: Fuse(Scw): %d decoded, %d returned\n", cnt, ret); abort(); }
                out = best;
            } else {
                std::vector<MapPoint *> matched(NK, nullptr);
                for (int k = 0; k < NK; k++) matched[k] = (MapPoint *)kmp[k];
                ret = mt->SearchByProjection((KeyFrame *)kf, Scw, list, matched, (int)lc[c].th);
                out.assign(NK, -1);
                int cnt = 0;
                for (int k = 0; k < NK; k++) {
                    if (init[k] == -2) { out[k] = -2; if ((void *)matched[k] != kmp[k]) { fprintf(stderr, "refprobe: occupied slot overwritten\n"); abort(); } continue; }
                    if (matched[k]) { out[k] = mp_index(matched[k]); cnt++; }
                }
                if (cnt != ret) { fprintf(stderr, "refprobe: SearchByProjection(Scw): %d decoded, %d returned\n", cnt, ret); abort(); }
                for (int i = 0; i < M; i++) valid[i] = !bad[i] && !pre[i];
            }
            std::vector<float> kx(NK), ky(NK), Scv(Sc, Sc + 16), scv(scale, scale + 8), cam = {fx, fy, cx, cy, bf, logsf, lc[c].th};
            std::vector<int> ko(NK);
            for (int k = 0; k < NK; k++) { kx[k] = kk[k].x; ky[k] = kk[k].y; ko[k] = kk[k].octave; }
            fprintf(JL, "{\"n_kf\": %d, \"m\": %d, \"ret\": %d, ", NK, M, ret);
            J = JL;
            jarr_f("cam", cam); jarr_f("Scw", Scv); jarr_f("scale", scv);
            jarr_f("x", kx); jarr_f("y", ky); jarr_i("octave", ko); jarr_i("init", init);
            jarr_f("world_pos", wpos); jarr_f("normal", nrm); jarr_f("min_dist", dmin); jarr_f("max_dist", dmax); jarr_i("valid", valid);
            jarr_i("out", out);
            fprintf(JL, "\"desc\": \"");
            for (size_t b = 0; b < kdesc.size(); b++) fprintf(JL, "%02x", kdesc[b]);
            fprintf(JL, "\", \"mp_desc\": \"");
            for (size_t b = 0; b < mdesc.size(); b++) fprintf(JL, "%02x", mdesc[b]);
            fprintf(JL, "\"}%s\n", c + 1 < NLC ? "," : "");
        }
        fprintf(JL, "]}\n"); fclose(JL);
    }
    // ------------------------------------------------------------ N: ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (so@0x838b0, glue)
    // KeyFrame::GetMapPointMatches and MapPoint::GetIndexInKeyFrame (so@0x91cd0) run from the binary on the hand-laid objects.
    {
        path = std::string(outdir) + "/ref_glue_sim3.json";
        FILE *JN = fopen(path.c_str(), "w");
        fprintf(JN, "{\"_doc\": \"ORBmatcher::SearchBySim3 (so@0x838b0) executed from the reference binary on two hand-laid keyframes; cv::Mat algebra supplied by "
                    "oracle/refprobe/probe.cpp. floats as uint32 bit patterns; valid1/valid2 = map point present, not bad, not matched on entry; match12[i1] = key point of "
                    "keyframe 2 stored by the call (-1 none)\", \"cases\": [\n");
        struct { int n; float th, s12; uint64_t seed; } nc[] = {{800, 7.5f, 1.04f, 10101}, {600, 10.0f, 0.97f, 10102}, {900, 5.0f, 1.0f, 10103}};
        const int NNC = 3;
        for (int c = 0; c < NNC; c++) {
            rng_seed(nc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int N = nc[c].n;
            const float fxy[2][4] = {{517.3f, 516.5f, 318.6f, 255.3f}, {520.9f, 521.0f, 325.1f, 249.7f}};
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            const float logsf = logf(1.2f);
            float *T[2] = {(float *)bump(64), (float *)bump(64)};
            for (int q = 0; q < 2; q++) {
                const float ay = q == 0 ? 0.05f : -0.04f, ax = q == 0 ? -0.02f : 0.03f, cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float R[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) T[q][r * 4 + k] = R[r * 3 + k];
                T[q][3] = q == 0 ? 0.1f : -0.25f; T[q][7] = q == 0 ? 0.02f : -0.03f; T[q][11] = q == 0 ? -0.05f : 0.12f;
                T[q][12] = T[q][13] = T[q][14] = 0.f; T[q][15] = 1.f;
            }
            // S12: camera 2 -> camera 1, R12 = R1w R2w^T, t12 = t1w - s R12 t2w (plus the scale under test)
            float *R12 = (float *)bump(48), *t12 = (float *)bump(16);
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { float a = 0; for (int j = 0; j < 3; j++) a += T[0][r * 4 + j] * T[1][k * 4 + j]; R12[r * 3 + k] = a; }
            for (int r = 0; r < 3; r++) t12[r] = T[0][r * 4 + 3] - nc[c].s12 * (R12[r * 3] * T[1][3] + R12[r * 3 + 1] * T[1][7] + R12[r * 3 + 2] * T[1][11]);
            std::vector<cv::KeyPoint> kk[2] = {std::vector<cv::KeyPoint>(N), std::vector<cv::KeyPoint>(N)};
            std::vector<uint8_t> kdesc[2] = {std::vector<uint8_t>((size_t)N * 32), std::vector<uint8_t>((size_t)N * 32)}, mdesc[2] = {kdesc[0], kdesc[0]};
            std::vector<float> wpos[2] = {std::vector<float>((size_t)N * 3), std::vector<float>((size_t)N * 3)}, dmin[2] = {std::vector<float>(N), std::vector<float>(N)}, dmax[2] = {dmin[0], dmin[0]};
            std::vector<int> has[2] = {std::vector<int>(N), std::vector<int>(N)}, bad[2] = {has[0], has[0]}, pre1(N, -1);
            auto cam_of = [&](int q, const float *xw, float *xc) { for (int r = 0; r < 3; r++) xc[r] = T[q][r * 4] * xw[0] + T[q][r * 4 + 1] * xw[1] + T[q][r * 4 + 2] * xw[2] + T[q][r * 4 + 3]; };
            for (int i = 0; i < N; i++) {
                // a world point seen by keyframe 1 at key point i; keyframe 2 gets its re-projection at key point i as well (then shuffled by index offset)
                const int i2 = (i * 7 + 3) % N;   // N is not a multiple of 7: a permutation
                const float u1 = 10.f + uf() * 620.f, v1 = 10.f + uf() * 460.f, z1 = 0.8f + uf() * 6.f;
                const float Xc1[3] = {(u1 - fxy[0][2]) / fxy[0][0] * z1 - T[0][3], (v1 - fxy[0][3]) / fxy[0][1] * z1 - T[0][7], z1 - T[0][11]};
                float Xw[3];
                for (int r = 0; r < 3; r++) Xw[r] = T[0][0 * 4 + r] * Xc1[0] + T[0][1 * 4 + r] * Xc1[1] + T[0][2 * 4 + r] * Xc1[2];
                float Xc2[3]; cam_of(1, Xw, Xc2);
                const float u2 = fxy[1][0] * Xc2[0] / Xc2[2] + fxy[1][2], v2 = fxy[1][1] * Xc2[1] / Xc2[2] + fxy[1][3];
                const int o1 = (int)rng_below(8), o2 = std::max(0, std::min(7, o1 + (int)rng_below(3) - 1));
                cv::KeyPoint a = {u1 + (uf() - 0.5f) * 4.f, v1 + (uf() - 0.5f) * 4.f, 31.f, uf() * 360.f, 1.f, o1, -1};
                cv::KeyPoint b = {u2 + (uf() - 0.5f) * 4.f, v2 + (uf() - 0.5f) * 4.f, 31.f, uf() * 360.f, 1.f, o2, -1};
                if (uf() < 0.1f) { b.x = uf() * 640.f; b.y = uf() * 480.f; }
                kk[0][i] = a; kk[1][i2] = b;
                for (int k = 0; k < 32; k++) { kdesc[0][(size_t)i * 32 + k] = (uint8_t)rng_below(256); kdesc[1][(size_t)i2 * 32 + k] = kdesc[0][(size_t)i * 32 + k]; }
                for (int q = 0, nf = (int)rng_below(70); q < nf; q++) { const int bit = (int)rng_below(256); kdesc[1][(size_t)i2 * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                // the two map points (one per keyframe) sit near the world point; each carries the descriptor of its own keyframe's key point, a little noisy
                const float d1 = sqrtf(Xc1[0] * Xc1[0] + Xc1[1] * Xc1[1] + Xc1[2] * Xc1[2] + 2 * (Xc1[0] * T[0][3] + Xc1[1] * T[0][7] + Xc1[2] * T[0][11]) + T[0][3] * T[0][3] + T[0][7] * T[0][7] + T[0][11] * T[0][11]);
                const float d2 = sqrtf(Xc2[0] * Xc2[0] + Xc2[1] * Xc2[1] + Xc2[2] * Xc2[2]);
                for (int q = 0; q < 2; q++) {
                    const int idx = q == 0 ? i : i2;
                    for (int r = 0; r < 3; r++) wpos[q][(size_t)idx * 3 + r] = Xw[r] + (uf() - 0.5f) * 0.01f;
                    // invariance range judged from the OTHER camera (the one the point is projected into), around that key point's octave
                    const float dd = q == 0 ? d2 : d1; const int oo = q == 0 ? o2 : o1;
                    dmax[q][idx] = dd * powf(1.2f, (float)oo + (uf() < 0.5f ? 0.f : 1.f) - 0.5f + (uf() - 0.5f) * 0.9f);
                    dmin[q][idx] = dmax[q][idx] / powf(1.2f, 7.f);
                    if (uf() < 0.05f) { dmax[q][idx] = dd * 0.6f; dmin[q][idx] = dmax[q][idx] / 4.f; }
                    has[q][idx] = uf() < 0.85f; bad[q][idx] = uf() < 0.04f;
                    for (int k = 0; k < 32; k++) mdesc[q][(size_t)idx * 32 + k] = kdesc[q][(size_t)idx * 32 + k];
                    for (int w = 0, nf = (int)rng_below(30); w < nf; w++) { const int bit = (int)rng_below(256); mdesc[q][(size_t)idx * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                }
                if (uf() < 0.06f) pre1[i] = i2;   // already matched on entry (vpMatches12[i] = map point of keyframe 2 at i2)
            }
            const size_t MPS = 0x300;
            char *kfs[2] = {(char *)bump(0x800), (char *)bump(0x800)};
            char *mps[2] = {(char *)bump((size_t)N * MPS), (char *)bump((size_t)N * MPS)};
            std::vector<void *> kmp[2] = {std::vector<void *>(N, nullptr), std::vector<void *>(N, nullptr)};
            std::vector<std::vector<std::vector<size_t>>> G[2];
            for (int q = 0; q < 2; q++) {
                char *kf = kfs[q]; memset(kf, 0, 0x800); memset(mps[q], 0, (size_t)N * MPS);
                for (int i = 0; i < N; i++) {
                    char *o = mps[q] + (size_t)i * MPS;
                    *(int *)(o + 0x18) = 2;
                    new (o + 0x138) std::map<KeyFrame *, size_t>();
                    (*(std::map<KeyFrame *, size_t> *)(o + 0x138))[(KeyFrame *)kf] = (size_t)i;
                    mat_init((cv::Mat *)(o + 0xd8), (unsigned char *)&wpos[q][(size_t)i * 3], 3, 1, 4);
                    ((cv::Mat *)(o + 0xd8))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(o + 0xd8))->step_buf[1] = 4;
                    mat_init((cv::Mat *)(o + 0x1c8), &mdesc[q][(size_t)i * 32], 1, 32, 32); ((cv::Mat *)(o + 0x1c8))->flags |= 0x4000;
                    *(bool *)(o + 0x238) = bad[q][i] != 0; *(float *)(o + 0x248) = dmin[q][i]; *(float *)(o + 0x24c) = dmax[q][i];
                    if (has[q][i]) kmp[q][i] = o;
                }
                *(int *)(kf + 0x18) = 64; *(int *)(kf + 0x1c) = 48; *(float *)(kf + 0x20) = 64.f / 640.f; *(float *)(kf + 0x24) = 48.f / 480.f;
                *(float *)(kf + 0x130) = fxy[q][0]; *(float *)(kf + 0x134) = fxy[q][1]; *(float *)(kf + 0x138) = fxy[q][2]; *(float *)(kf + 0x13c) = fxy[q][3]; *(float *)(kf + 0x148) = 40.f;
                *(int *)(kf + 0x154) = N;
                void **v;
                v = (void **)(kf + 0x170); v[0] = kk[q].data(); v[1] = kk[q].data() + N; v[2] = v[1];
                mat_init((cv::Mat *)(kf + 0x1b8), kdesc[q].data(), N, 32, 32); ((cv::Mat *)(kf + 0x1b8))->flags |= 0x4000;
                *(int *)(kf + 0x2d8) = 8; *(float *)(kf + 0x2e0) = logsf;
                v = (void **)(kf + 0x2e8); v[0] = scale; v[1] = scale + 8; v[2] = v[1];
                v = (void **)(kf + 0x318); v[0] = is2; v[1] = is2 + 8; v[2] = v[1];
                *(int *)(kf + 0x330) = 0; *(int *)(kf + 0x334) = 0; *(int *)(kf + 0x338) = 640; *(int *)(kf + 0x33c) = 480;
                mat_init((cv::Mat *)(kf + 0x3a0), (unsigned char *)T[q], 4, 4, 16);
                ((cv::Mat *)(kf + 0x3a0))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(kf + 0x3a0))->step_buf[1] = 4;
                v = (void **)(kf + 0x520); v[0] = kmp[q].data(); v[1] = kmp[q].data() + N; v[2] = v[1];
                G[q].assign(64, std::vector<std::vector<size_t>>(48));
                for (int k = 0; k < N; k++) {
                    const int gx = (int)roundf(kk[q][k].x * (64.f / 640.f)), gy = (int)roundf(kk[q][k].y * (48.f / 480.f));
                    if (gx < 0 || gx >= 64 || gy < 0 || gy >= 48) continue;
                    G[q][gx][gy].push_back((size_t)k);
                }
                memcpy(kf + 0x548, (void *)&G[q], sizeof(G[q]));
            }
            std::vector<MapPoint *> m12(N, nullptr);
            std::vector<int> valid[2] = {std::vector<int>(N), std::vector<int>(N)};
            std::vector<int> am2(N, 0);
            for (int i = 0; i < N; i++) if (pre1[i] >= 0 && has[1][pre1[i]]) { m12[i] = (MapPoint *)(mps[1] + (size_t)pre1[i] * MPS); am2[pre1[i]] = 1; } else pre1[i] = -1;
            for (int i = 0; i < N; i++) { valid[0][i] = has[0][i] && !bad[0][i] && pre1[i] < 0; valid[1][i] = has[1][i] && !bad[1][i] && !am2[i]; }
            cv::Mat R12m, t12m;
            mat_init(&R12m, (unsigned char *)R12, 3, 3, 12); R12m.flags = 0x42FF0000 | 0x4000 | 5; R12m.step_buf[1] = 4;
            mat_init(&t12m, (unsigned char *)t12, 3, 1, 4); t12m.flags = 0x42FF0000 | 0x4000 | 5; t12m.step_buf[1] = 4;
            ORBmatcher *mt = new ORBmatcher(0.75f, true);
            const float s12 = nc[c].s12;
            const int nfound = mt->SearchBySim3((KeyFrame *)kfs[0], (KeyFrame *)kfs[1], m12, s12, R12m, t12m, nc[c].th);
            std::vector<int> match(N, -1);
            int cnt = 0;
            for (int i = 0; i < N; i++) if (m12[i] && pre1[i] < 0) { match[i] = (int)(((char *)m12[i] - mps[1]) / MPS); cnt++; }
            if (cnt != nfound) { fprintf(stderr, "refprobe: SearchBySim3: %d decoded, %d returned\n", cnt, nfound); abort(); }
            fprintf(JN, "{\"n\": %d, \"nfound\": %d, ", N, nfound);
            J = JN;
            std::vector<float> cam = {fxy[0][0], fxy[0][1], fxy[0][2], fxy[0][3], fxy[1][0], fxy[1][1], fxy[1][2], fxy[1][3], logsf, nc[c].th, s12};
            jarr_f("cam", cam); jarr_f("T1w", std::vector<float>(T[0], T[0] + 16)); jarr_f("T2w", std::vector<float>(T[1], T[1] + 16));
            jarr_f("R12", std::vector<float>(R12, R12 + 9)); jarr_f("t12", std::vector<float>(t12, t12 + 3)); jarr_f("scale", std::vector<float>(scale, scale + 8));
            for (int q = 0; q < 2; q++) {
                std::vector<float> kx(N), ky(N); std::vector<int> ko(N);
                for (int k = 0; k < N; k++) { kx[k] = kk[q][k].x; ky[k] = kk[q][k].y; ko[k] = kk[q][k].octave; }
                const std::string sfx = q == 0 ? "1" : "2";
                jarr_f(("x" + sfx).c_str(), kx); jarr_f(("y" + sfx).c_str(), ky); jarr_i(("octave" + sfx).c_str(), ko);
                jarr_f(("world_pos" + sfx).c_str(), wpos[q]); jarr_f(("min_dist" + sfx).c_str(), dmin[q]); jarr_f(("max_dist" + sfx).c_str(), dmax[q]);
                jarr_i(("valid" + sfx).c_str(), valid[q]);
                fprintf(JN, "\"desc%s\": \"", sfx.c_str());
                for (size_t b = 0; b < kdesc[q].size(); b++) fprintf(JN, "%02x", kdesc[q][b]);
                fprintf(JN, "\", \"mp_desc%s\": \"", sfx.c_str());
                for (size_t b = 0; b < mdesc[q].size(); b++) fprintf(JN, "%02x", mdesc[q][b]);
                fprintf(JN, "\", ");
            }
            jarr_i("match12", match, true);
            fprintf(JN, "}%s\n", c + 1 < NNC ? "," : "");
        }
        fprintf(JN, "]}\n"); fclose(JN);
    }
    // ------------------------------------------------------------ O: ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (so@0x86b30, glue)
    // KeyFrame mvLevelSigma2 @0x300 (CheckDistEpipolarLine so@0x79b90); KeyFrame::GetMapPoint / GetCameraCenter / GetRotation / GetTranslation run from the binary.
    {
        path = std::string(outdir) + "/ref_glue_triangulation.json";
        FILE *JO = fopen(path.c_str(), "w");
        fprintf(JO, "{\"_doc\": \"ORBmatcher::SearchForTriangulation (so@0x86b30) executed from the reference binary on two hand-laid keyframes. floats as uint32 bit "
                    "patterns; has_mp = GetMapPoint(i) != NULL (such features are skipped); match12[i1] = keyframe-2 feature paired with keyframe-1 feature i1, -1 none\", "
                    "\"cases\": [\n");
        struct { int n, nodes, only_stereo, check; float tz; uint64_t seed; } oc[] = {{710, 120, 0, 1, 0.02f, 10201}, {600, 90, 0, 1, 0.45f, 10202}, {650, 100, 1, 0, 0.1f, 10203}};
        const int NOC = 3;
        typedef std::map<unsigned, std::vector<unsigned>> FeatVec;
        for (int c = 0; c < NOC; c++) {
            rng_seed(oc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int N = oc[c].n, NN = oc[c].nodes;
            const float fx = 517.3f, fy = 516.5f, cx = 318.6f, cy = 255.3f, bf = 40.f;
            float scale[16], inv[16], s2[16], is2[16]; int per[16], um[16];
            orc_orb_tables(1000, 1.2f, 8, scale, inv, s2, is2, per, um);
            float *T[2] = {(float *)bump(64), (float *)bump(64)}, *Ow1 = (float *)bump(16);
            for (int q = 0; q < 2; q++) {
                const float ay = q == 0 ? 0.02f : -0.03f, ax = q == 0 ? -0.01f : 0.02f, cyw = cosf(ay), syw = sinf(ay), cxw = cosf(ax), sxw = sinf(ax);
                const float R[9] = {cyw, syw * sxw, syw * cxw, 0.f, cxw, -sxw, -syw, cyw * sxw, cyw * cxw};
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) T[q][r * 4 + k] = R[r * 3 + k];
                T[q][3] = q == 0 ? 0.0f : -0.12f; T[q][7] = q == 0 ? 0.0f : 0.03f; T[q][11] = q == 0 ? 0.0f : oc[c].tz;
                T[q][12] = T[q][13] = T[q][14] = 0.f; T[q][15] = 1.f;
            }
            for (int i = 0; i < 3; i++) Ow1[i] = -(T[0][0 * 4 + i] * T[0][3] + T[0][1 * 4 + i] * T[0][7] + T[0][2 * 4 + i] * T[0][11]);
            // F12 = K1^-T [t12]x R12 K2^-1 (LocalMapping::ComputeF12), same intrinsics for both keyframes
            float *F12 = (float *)bump(48);
            {
                double R12[9], t12[3];
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { double a = 0; for (int j = 0; j < 3; j++) a += (double)T[0][r * 4 + j] * T[1][k * 4 + j]; R12[r * 3 + k] = a; }
                for (int r = 0; r < 3; r++) t12[r] = T[0][r * 4 + 3] - (R12[r * 3] * T[1][3] + R12[r * 3 + 1] * T[1][7] + R12[r * 3 + 2] * T[1][11]);
                const double tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
                double E[9];
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { double a = 0; for (int j = 0; j < 3; j++) a += tx[r * 3 + j] * R12[j * 3 + k]; E[r * 3 + k] = a; }
                const double Ki[9] = {1.0 / fx, 0, -cx / (double)fx, 0, 1.0 / fy, -cy / (double)fy, 0, 0, 1};
                double M[9];
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { double a = 0; for (int j = 0; j < 3; j++) a += Ki[j * 3 + r] * E[j * 3 + k]; M[r * 3 + k] = a; }   // K^-T E
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { double a = 0; for (int j = 0; j < 3; j++) a += M[r * 3 + j] * Ki[j * 3 + k]; F12[r * 3 + k] = (float)a; }
            }
            std::vector<cv::KeyPoint> kk[2] = {std::vector<cv::KeyPoint>(N), std::vector<cv::KeyPoint>(N)};
            std::vector<uint8_t> kdesc[2] = {std::vector<uint8_t>((size_t)N * 32), std::vector<uint8_t>((size_t)N * 32)};
            std::vector<float> ur[2] = {std::vector<float>(N), std::vector<float>(N)};
            std::vector<int> has[2] = {std::vector<int>(N), std::vector<int>(N)};
            std::vector<unsigned> nd[2] = {std::vector<unsigned>(N), std::vector<unsigned>(N)};
            for (int i = 0; i < N; i++) {
                const int i2 = (i * 7 + 5) % N;
                const float u1 = 10.f + uf() * 620.f, v1 = 10.f + uf() * 460.f, z1 = 0.8f + uf() * 6.f;
                const float Xc1[3] = {(u1 - cx) / fx * z1 - T[0][3], (v1 - cy) / fy * z1 - T[0][7], z1 - T[0][11]};
                float Xw[3], Xc2[3];
                for (int r = 0; r < 3; r++) Xw[r] = T[0][0 * 4 + r] * Xc1[0] + T[0][1 * 4 + r] * Xc1[1] + T[0][2 * 4 + r] * Xc1[2];
                for (int r = 0; r < 3; r++) Xc2[r] = T[1][r * 4] * Xw[0] + T[1][r * 4 + 1] * Xw[1] + T[1][r * 4 + 2] * Xw[2] + T[1][r * 4 + 3];
                const float u2 = fx * Xc2[0] / Xc2[2] + cx, v2 = fy * Xc2[1] / Xc2[2] + cy;
                const int o1 = (int)rng_below(8), o2 = std::max(0, std::min(7, o1 + (int)rng_below(3) - 1));
                const float noise = uf() < 0.8f ? 1.5f : 12.f;   // some pairs violate the epipolar constraint
                cv::KeyPoint a = {u1, v1, 31.f, uf() * 360.f, 1.f, o1, -1};
                float ang2 = a.angle - (uf() < 0.8f ? 20.f + uf() * 8.f : uf() * 360.f);
                if (ang2 < 0.f) ang2 += 360.f;
                cv::KeyPoint b = {u2 + (uf() - 0.5f) * noise * scale[o2], v2 + (uf() - 0.5f) * noise * scale[o2], 31.f, ang2, 1.f, o2, -1};
                kk[0][i] = a; kk[1][i2] = b;
                ur[0][i] = uf() < 0.5f ? u1 - bf / z1 : -1.f; ur[1][i2] = uf() < 0.5f ? b.x - bf / Xc2[2] : -1.f;
                has[0][i] = uf() < 0.35f; has[1][i2] = uf() < 0.35f;
                for (int k = 0; k < 32; k++) { kdesc[0][(size_t)i * 32 + k] = (uint8_t)rng_below(256); kdesc[1][(size_t)i2 * 32 + k] = kdesc[0][(size_t)i * 32 + k]; }
                for (int q = 0, nf = (int)rng_below(64); q < nf; q++) { const int bit = (int)rng_below(256); kdesc[1][(size_t)i2 * 32 + bit / 8] ^= (uint8_t)(1u << (bit & 7)); }
                nd[0][i] = 700u + 2u * rng_below((uint32_t)NN);
                nd[1][i2] = uf() < 0.9f ? nd[0][i] : 700u + 2u * rng_below((uint32_t)NN) + (uf() < 0.3f ? 1u : 0u);
            }
            // a second look-alike of some keyframe-1 features inside the same node (ties / equal distances)
            for (int t = 0; t < N / 10; t++) {
                const int src = (int)rng_below(N), dst = (int)rng_below(N);
                if (src == dst) continue;
                kk[1][dst].x = kk[1][src].x + (uf() - 0.5f); kk[1][dst].y = kk[1][src].y + (uf() - 0.5f); kk[1][dst].octave = kk[1][src].octave;
                memcpy(&kdesc[1][(size_t)dst * 32], &kdesc[1][(size_t)src * 32], 32); nd[1][dst] = nd[1][src]; ur[1][dst] = ur[1][src]; has[1][dst] = has[1][src];
            }
            char *kfs[2] = {(char *)bump(0x800), (char *)bump(0x800)};
            char *mpb = (char *)bump(0x300); memset(mpb, 0, 0x300);
            std::vector<void *> kmp[2] = {std::vector<void *>(N, nullptr), std::vector<void *>(N, nullptr)};
            FeatVec *fv[2];
            for (int q = 0; q < 2; q++) {
                char *kf = kfs[q]; memset(kf, 0, 0x800);
                for (int i = 0; i < N; i++) if (has[q][i]) kmp[q][i] = mpb;
                *(float *)(kf + 0x130) = fx; *(float *)(kf + 0x134) = fy; *(float *)(kf + 0x138) = cx; *(float *)(kf + 0x13c) = cy; *(float *)(kf + 0x148) = bf;
                *(int *)(kf + 0x154) = N;
                void **v;
                v = (void **)(kf + 0x170); v[0] = kk[q].data(); v[1] = kk[q].data() + N; v[2] = v[1];
                v = (void **)(kf + 0x188); v[0] = ur[q].data(); v[1] = ur[q].data() + N; v[2] = v[1];
                mat_init((cv::Mat *)(kf + 0x1b8), kdesc[q].data(), N, 32, 32); ((cv::Mat *)(kf + 0x1b8))->flags |= 0x4000;
                fv[q] = new (kf + 0x248) FeatVec();
                for (int i = 0; i < N; i++) (*fv[q])[nd[q][i]].push_back((unsigned)i);
                *(int *)(kf + 0x2d8) = 8;
                v = (void **)(kf + 0x2e8); v[0] = scale; v[1] = scale + 8; v[2] = v[1];
                v = (void **)(kf + 0x300); v[0] = s2; v[1] = s2 + 8; v[2] = v[1];
                mat_init((cv::Mat *)(kf + 0x3a0), (unsigned char *)T[q], 4, 4, 16);
                ((cv::Mat *)(kf + 0x3a0))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(kf + 0x3a0))->step_buf[1] = 4;
                if (q == 0) { mat_init((cv::Mat *)(kf + 0x460), (unsigned char *)Ow1, 3, 1, 4); ((cv::Mat *)(kf + 0x460))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(kf + 0x460))->step_buf[1] = 4; }
                v = (void **)(kf + 0x520); v[0] = kmp[q].data(); v[1] = kmp[q].data() + N; v[2] = v[1];
            }
            cv::Mat Fm;
            mat_init(&Fm, (unsigned char *)F12, 3, 3, 12); Fm.flags = 0x42FF0000 | 0x4000 | 5; Fm.step_buf[1] = 4;
            ORBmatcher *mt = new ORBmatcher(0.6f, oc[c].check != 0);
            std::vector<std::pair<size_t, size_t>> pairs;
            const int nm = mt->SearchForTriangulation((KeyFrame *)kfs[0], (KeyFrame *)kfs[1], Fm, pairs, oc[c].only_stereo != 0);
            if ((int)pairs.size() != nm) { fprintf(stderr, "refprobe: SearchForTriangulation: %zu pairs, %d returned\n", pairs.size(), nm); abort(); }
            std::vector<int> match(N, -1);
            for (auto &pr : pairs) match[pr.first] = (int)pr.second;
            auto flat = [&](FeatVec *f, std::vector<int> &ids, std::vector<int> &starts, std::vector<int> &feats) {
                for (auto &kv : *f) { ids.push_back((int)kv.first); starts.push_back((int)feats.size()); for (unsigned x : kv.second) feats.push_back((int)x); }
                starts.push_back((int)feats.size());
            };
            fprintf(JO, "{\"n\": %d, \"only_stereo\": %d, \"check_orientation\": %d, \"nmatches\": %d, ", N, oc[c].only_stereo, oc[c].check, nm);
            J = JO;
            std::vector<float> cam = {fx, fy, cx, cy};
            jarr_f("cam", cam); jarr_f("Ow1", std::vector<float>(Ow1, Ow1 + 3)); jarr_f("T2w", std::vector<float>(T[1], T[1] + 16));
            jarr_f("F12", std::vector<float>(F12, F12 + 9)); jarr_f("scale", std::vector<float>(scale, scale + 8)); jarr_f("sigma2", std::vector<float>(s2, s2 + 8));
            for (int q = 0; q < 2; q++) {
                std::vector<float> kx(N), ky(N), ka(N); std::vector<int> ko(N), ids, st, fe;
                for (int k = 0; k < N; k++) { kx[k] = kk[q][k].x; ky[k] = kk[q][k].y; ka[k] = kk[q][k].angle; ko[k] = kk[q][k].octave; }
                flat(fv[q], ids, st, fe);
                const std::string sfx = q == 0 ? "1" : "2";
                jarr_f(("x" + sfx).c_str(), kx); jarr_f(("y" + sfx).c_str(), ky); jarr_f(("angle" + sfx).c_str(), ka); jarr_i(("octave" + sfx).c_str(), ko);
                jarr_f(("uright" + sfx).c_str(), ur[q]); jarr_i(("has_mp" + sfx).c_str(), has[q]);
                jarr_i(("node_id" + sfx).c_str(), ids); jarr_i(("node_start" + sfx).c_str(), st); jarr_i(("feat" + sfx).c_str(), fe);
                fprintf(JO, "\"desc%s\": \"", sfx.c_str());
                for (size_t b = 0; b < kdesc[q].size(); b++) fprintf(JO, "%02x", kdesc[q][b]);
                fprintf(JO, "\", ");
            }
            jarr_i("match12", match, true);
            fprintf(JO, "}%s\n", c + 1 < NOC ? "," : "");
        }
        fprintf(JO, "]}\n"); fclose(JO);
    }
    // ------------------------------------------------------------ P: Frame::AssignFeaturesToGrid() (so@0xf9120) with Frame::PosInGrid (so@0xf5fa0)
    // N @0xec, mvKeysUn @0x120, mGrid[64][48] @0x2c8 (std::vector<size_t>, 24 B each); statics mnMinX/Y, mfGridElementWidthInv/HeightInv.
    {
        path = std::string(outdir) + "/ref_glue_grid.json";
        FILE *JP = fopen(path.c_str(), "w");
        fprintf(JP, "{\"_doc\": \"Frame::AssignFeaturesToGrid (so@0xf9120) executed from the reference binary: the 64x48 cell lists (CSR over cell = ix*48+iy) for key "
                    "points on and around the image, incl. half-cell and border positions. floats as uint32 bit patterns\", \"cases\": [\n");
        struct { int n; float minx, miny, maxx, maxy; uint64_t seed; } pc[] = {{1500, 0.f, 0.f, 640.f, 480.f, 10301}, {900, -11.4f, -8.7f, 652.3f, 489.1f, 10302}};
        for (int c = 0; c < 2; c++) {
            rng_seed(pc[c].seed);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int N = pc[c].n;
            Frame::mnMinX = pc[c].minx; Frame::mnMinY = pc[c].miny; Frame::mnMaxX = pc[c].maxx; Frame::mnMaxY = pc[c].maxy;
            Frame::mfGridElementWidthInv = 64.f / (pc[c].maxx - pc[c].minx); Frame::mfGridElementHeightInv = 48.f / (pc[c].maxy - pc[c].miny);
            std::vector<cv::KeyPoint> ku(N);
            for (int i = 0; i < N; i++) {
                float x = pc[c].minx - 6.f + uf() * (pc[c].maxx - pc[c].minx + 12.f), y = pc[c].miny - 6.f + uf() * (pc[c].maxy - pc[c].miny + 12.f);
                const float wcell = (pc[c].maxx - pc[c].minx) / 64.f, hcell = (pc[c].maxy - pc[c].miny) / 48.f;
                const float u = uf();
                if (u < 0.15f) x = pc[c].minx + wcell * ((float)rng_below(65) + 0.5f);            // exactly between two cells (round-half cases)
                else if (u < 0.3f) y = pc[c].miny + hcell * ((float)rng_below(49) + 0.5f);
                else if (u < 0.35f) { x = uf() < 0.5f ? pc[c].minx : pc[c].maxx; }
                else if (u < 0.4f) { y = uf() < 0.5f ? pc[c].miny : pc[c].maxy; }
                ku[i].x = x; ku[i].y = y; ku[i].size = 31.f; ku[i].angle = 0.f; ku[i].response = 1.f; ku[i].octave = 0; ku[i].class_id = -1;
            }
            char *fr = (char *)bump(0x12800); memset(fr, 0, 0x12800);
            *(int *)(fr + 0xec) = N;
            void **v = (void **)(fr + 0x120); v[0] = ku.data(); v[1] = ku.data() + N; v[2] = v[1];
            ((Frame *)fr)->AssignFeaturesToGrid();
            std::vector<int> start(64 * 48 + 1, 0), idx;
            for (int cell = 0; cell < 64 * 48; cell++) {
                size_t **g = (size_t **)(fr + 0x2c8 + (size_t)cell * 24);
                start[cell] = (int)idx.size();
                for (size_t *q = g[0]; q != g[1]; q++) idx.push_back((int)*q);
            }
            start[64 * 48] = (int)idx.size();
            std::vector<float> kx(N), ky(N), bnd = {pc[c].minx, pc[c].miny, pc[c].maxx, pc[c].maxy};
            for (int i = 0; i < N; i++) { kx[i] = ku[i].x; ky[i] = ku[i].y; }
            fprintf(JP, "{\"n\": %d, ", N);
            J = JP;
            jarr_f("bounds", bnd); jarr_f("x", kx); jarr_f("y", ky); jarr_i("cell_start", start); jarr_i("cell_idx", idx, true);
            fprintf(JP, "}%s\n", c == 0 ? "," : "");
        }
        fprintf(JP, "]}\n"); fclose(JP);
    }
    // ------------------------------------------------------------ Q: Frame::UndistortKeyPoints() (so@0xf8630, glue): mK @0x20, mDistCoef @0x80, N @0xec, mvKeys @0xf0, mvKeysUn @0x120
    {
        path = std::string(outdir) + "/ref_glue_undistort.json";
        FILE *JQ = fopen(path.c_str(), "w");
        fprintf(JQ, "{\"_doc\": \"Frame::UndistortKeyPoints (so@0xf8630) executed from the reference binary (its Mat fill / reshape / copy-back glue; cv::undistortPoints is "
                    "the restatement in oracle/frame_oracle.c). floats as uint32 bit patterns; cam = fx, fy, cx, cy, k1, k2, p1, p2, k3 (TUM1.yaml / no distortion)\", \"cases\": [\n");
        const float cams[2][9] = {{517.306408f, 516.469215f, 318.643040f, 255.313989f, 0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f},
                                  {535.4f, 539.2f, 320.1f, 247.6f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        for (int c = 0; c < 2; c++) {
            rng_seed(10401 + c);
            auto uf = [&]() { return (float)(rng_u32() >> 8) * (1.0f / 16777216.0f); };
            const int N = 800;
            std::vector<cv::KeyPoint> keys(N);
            for (int i = 0; i < N; i++) { keys[i].x = uf() * 640.f; keys[i].y = uf() * 480.f; keys[i].size = 31.f + (float)rng_below(80); keys[i].angle = uf() * 360.f; keys[i].response = (float)rng_below(200); keys[i].octave = (int)rng_below(8); keys[i].class_id = -1; }
            float *Km = (float *)bump(48), *Dm = (float *)bump(32);
            const float Kv[9] = {cams[c][0], 0, cams[c][2], 0, cams[c][1], cams[c][3], 0, 0, 1};
            memcpy(Km, Kv, sizeof(Kv)); memcpy(Dm, &cams[c][4], 5 * sizeof(float));
            char *fr = (char *)bump(0x12800); memset(fr, 0, 0x12800);
            mat_init((cv::Mat *)(fr + 0x20), (unsigned char *)Km, 3, 3, 12); ((cv::Mat *)(fr + 0x20))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(fr + 0x20))->step_buf[1] = 4;
            mat_init((cv::Mat *)(fr + 0x80), (unsigned char *)Dm, 5, 1, 4); ((cv::Mat *)(fr + 0x80))->flags = 0x42FF0000 | 0x4000 | 5; ((cv::Mat *)(fr + 0x80))->step_buf[1] = 4;
            *(int *)(fr + 0xec) = N;
            void **v = (void **)(fr + 0xf0); v[0] = keys.data(); v[1] = keys.data() + N; v[2] = v[1];
            ((Frame *)fr)->UndistortKeyPoints();
            cv::KeyPoint **un = (cv::KeyPoint **)(fr + 0x120);
            if (un[1] - un[0] != N) { fprintf(stderr, "refprobe: mvKeysUn has %ld entries\n", (long)(un[1] - un[0])); abort(); }
            std::vector<float> kx(N), ky(N), ux(N), uy(N), cam(cams[c], cams[c] + 9);
            int same = 1;
            for (int i = 0; i < N; i++) {
                kx[i] = keys[i].x; ky[i] = keys[i].y; ux[i] = un[0][i].x; uy[i] = un[0][i].y;
                same &= un[0][i].size == keys[i].size && un[0][i].angle == keys[i].angle && un[0][i].response == keys[i].response && un[0][i].octave == keys[i].octave && un[0][i].class_id == keys[i].class_id;
            }
            fprintf(JQ, "{\"n\": %d, \"other_fields_copied\": %d, ", N, same);
            J = JQ;
            jarr_f("cam", cam); jarr_f("x", kx); jarr_f("y", ky); jarr_f("x_un", ux); jarr_f("y_un", uy, true);
            fprintf(JQ, "}%s\n", c == 0 ? "," : "");
        }
        fprintf(JQ, "]}\n"); fclose(JQ);
    }
    printf("refprobe: fixtures written to %s\n", outdir);
    return 0;
}
