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

__device__ __forceinline__ int plf_wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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

// 256-bit Hamming distance of two 32-byte descriptors held as 8 dwords
__device__ __forceinline__ int plf_hamming8(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
    return d;
}
#endif
