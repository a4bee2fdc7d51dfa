// plf_common.h -- shared host/device helpers of the MI355X (gfx950) feature front-end.
// CDNA4 only: wavefront = 64 lanes is hard-coded throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/plf.h"

#define PLF_WAVE 64

#define PLF_HIP_TRY(expr)                                                                             \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            fprintf(stderr, "[plf] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e), __FILE__, __LINE__, \
                    hipGetErrorString(_e));                                                           \
            return PLF_E_HIP;                                                                         \
        }                                                                                             \
    } while (0)

static inline size_t plf_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#ifdef __HIPCC__
// ---- wave-level helpers (64 lanes)
__device__ __forceinline__ int plf_lane() { return threadIdx.x & 63; }

// sum over the 64 lanes (all of them active), result wave-uniform: four DPP steps leave every lane of a row of 16 with its row's sum, four v_readlane add the
// rows on the scalar unit.  (The butterfly over __shfl_xor was six ds_bpermute round trips with their address arithmetic and waits.)
__device__ __forceinline__ int plf_wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// exclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int plf_wave_excl_scan(int v)
{
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (plf_lane() >= o) x += y;
    }
    return x - v;
}

__device__ __forceinline__ int plf_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

// Block-wide exclusive scan of a[0..n) (ints, LDS or global), in place; returns the total.
// `tmp` must hold blockDim.x + 1 ints of LDS.  All threads of the block must call it.
__device__ inline int plf_block_excl_scan(int *a, int n, int *tmp)
{
    const int T = blockDim.x, t = threadIdx.x;
    const int per = (n + T - 1) / T;
    const int b = t * per, e = min(b + per, n);
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
    tmp[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < T; i++) { int v = tmp[i]; tmp[i] = run; run += v; }
        tmp[T] = run;
    }
    __syncthreads();
    int run = tmp[t];
    for (int i = b; i < e; i++) { int v = a[i]; a[i] = run; run += v; }
    const int total = tmp[T];
    __syncthreads();
    return total;
}

// cv::fastAtan2 (OpenCV 3.3 scalar path), degrees in [0,360).  Plain mul/add, no FMA (-ffp-contract=off).
__device__ __forceinline__ float plf_fast_atan2(float y, float x)
{
    const float p1 = (float)(0.9997878412794807 * (180 / 3.14159265358979323846));
    const float p3 = (float)(-0.3258083974640975 * (180 / 3.14159265358979323846));
    const float p5 = (float)(0.1555786518463281 * (180 / 3.14159265358979323846));
    const float p7 = (float)(-0.04432655554792128 * (180 / 3.14159265358979323846));
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
// the same arithmetic with ONE division (the two branches above each carry their own ~12-instruction IEEE division): numerator = the smaller of |x|, |y|,
// denominator = the larger + eps; for |x| == |y| both forms divide the same numbers
__device__ __forceinline__ float plf_fast_atan2_1div(float y, float x)
{
    const float p1 = (float)(0.9997878412794807 * (180 / 3.14159265358979323846));
    const float p3 = (float)(-0.3258083974640975 * (180 / 3.14159265358979323846));
    const float p5 = (float)(0.1555786518463281 * (180 / 3.14159265358979323846));
    const float p7 = (float)(-0.04432655554792128 * (180 / 3.14159265358979323846));
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool xge = ax >= ay;
    const float c = __fdiv_rn(xge ? ay : ax, (xge ? ax : ay) + eps);
    const float c2 = c * c;
    const float t = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = xge ? t : 90.f - t;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// sincosf as the reference's libm computes it (glibc >= 2.28 sysdeps/ieee754/flt-32/s_sincosf.c: quadrant
// reduction and two degree-7/8 polynomials evaluated in double, result rounded to float).  The reference
// binary calls sincosf@plt for the BRIEF steering angle (so@0x77803); a merely "correctly rounded" sin/cos
// differs from it by 1 ulp for a few percent of the angles, which can move a sample by one pixel.  This is the
// same sequence of IEEE double operations (no FMA), verified bit-identical to glibc 2.35 on 2*10^8 angles
// (oracle/orb_oracle.c: orc_sincosf_glibc, tests/test_oracle_props.py).  Valid for 0 <= |y| < 120.
__device__ __forceinline__ void plf_sincosf_glibc(float y, float *sinp, float *cosp)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    double x = (double)y;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ff;
    int n = 0;
    double sgn = 1.0, flip = 1.0;  // flip = -1 selects the negated cosine table (quadrants 2,3)
    if (top < ((0x3f490fdbu >> 20) & 0x7ff)) {  // |y| < pi/4 (compared on the top 12 bits, as glibc does)
        if (top < ((0x39800000u >> 20) & 0x7ff)) { *sinp = y; *cosp = 1.0f; return; }  // |y| < 2^-12
    } else {
        const double r = x * hpi_inv;
        n = ((int)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
        const int q = n & 3;
        sgn = (q == 1 || q == 2) ? -1.0 : 1.0;
        if (n & 2) flip = -1.0;
    }
    const double xr = x;      // reduced argument (x*x uses the unsigned one)
    const double xs = x * sgn;
    const double x2 = xr * xr;
    const double c0 = C0 * flip, c1k = C1 * flip, c2k = C2 * flip, c3k = C3 * flip, c4k = C4 * flip;  // exact sign flips
    const double x4 = x2 * x2;
    const double x3 = x2 * xs;
    const double c2 = c3k + x2 * c4k;
    const double s1 = S2 + x2 * S3;
    const double c1 = c0 + x2 * c1k;
    const double x5 = x3 * x2;
    const double x6 = x4 * x2;
    const double s = xs + x3 * S1;
    const double c = c1 + x4 * c2k;
    const float sv = (float)(s + x5 * s1), cv = (float)(c + x6 * c2);
    if (n & 1) { *sinp = cv; *cosp = sv; } else { *sinp = sv; *cosp = cv; }
}

// 256-bit Hamming distance of two 32-byte descriptors held as 8 dwords
__device__ __forceinline__ int plf_hamming8(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
    return d;
}
#endif

// Per-handle ordering of calls that arrive on different streams (include/plf.h, "Streams"): every call records an event on its stream when it has enqueued its
// work; a call on ANOTHER stream makes that stream wait for the event (device side).  No host wait, and the previous call's stream handle is never touched again --
// round 2 synchronised the cached hipStream_t of the previous call, which dangles once the caller has destroyed that stream (easy from Python,
// where torch streams are garbage-collected).  Plain data: the handles are calloc'ed.
struct PlfStreamOrder { hipEvent_t ev; bool set; hipStream_t last; };
static inline void plf_order_begin(PlfStreamOrder &o, hipStream_t s)
{
    if (o.set && o.last != s && o.ev) { if (hipStreamWaitEvent(s, o.ev, 0) != hipSuccess) (void)hipGetLastError(); }
    o.last = s;
}
static inline void plf_order_end(PlfStreamOrder &o, hipStream_t s)
{
    if (!o.ev && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); o.ev = nullptr; return; }
    if (hipEventRecord(o.ev, s) == hipSuccess) o.set = true; else (void)hipGetLastError();
}
static inline void plf_order_free(PlfStreamOrder &o) { if (o.ev) (void)hipEventDestroy(o.ev); o.ev = nullptr; o.set = false; }
struct PlfOrderGuard { PlfStreamOrder &o; hipStream_t s; ~PlfOrderGuard() { plf_order_end(o, s); } };
