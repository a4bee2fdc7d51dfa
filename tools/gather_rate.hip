// gather_rate.hip -- what does a scattered gather cost a CU that holds 32 latency-bound chains?  Every wave runs a DEPENDENT chain of gathers (the next
// addresses depend on the loaded words) with L active lanes; 8192 waves (32 per CU) like a launch of k_lsd_regions2.  If the time per step grows with L the
// chains are bound by the CU's address pipeline (lines per clock), not by the HBM round trip alone.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int WIDE>
__global__ void __launch_bounds__(512) k_chain(const uint32_t *__restrict__ a, size_t words_per_wave, int iters, int L, uint32_t *out)
{
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t *base = a + (size_t)wv * words_per_wave;          // each wave gathers inside its own 1.2 MB (a frame's angle map)
    uint32_t idx = lane * 977u + wv * 131u, acc = 0;
    if (lane >= L) return;
    for (int i = 0; i < iters; i++) {
        const uint32_t p = (idx % (uint32_t)(words_per_wave - 4)) & ~3u;
        uint32_t v;
        if (WIDE) { const uint4 q = *reinterpret_cast<const uint4 *>(base + p); v = q.x ^ q.w; }
        else v = base[p];
        acc += v;
        idx = idx * 1664525u + 1013904223u + (v & 1u);               // (dependent on the loaded word)
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main()
{
    const int waves = 8192; const size_t wpw = 300000;               // 1.2 MB per wave, 9.8 GB in all: nothing stays in L2 / MALL
    uint32_t *d_a, *d_out;
    CHECK(hipMalloc(&d_a, waves * wpw * 4)); CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemset(d_a, 1, waves * wpw * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\n");
    for (int wide = 0; wide <= 1; wide++)
        for (int wpc : {4, 16, 32})
            for (int L : {1, 8, 16, 32, 64}) {
                const int nw = 256 * wpc, iters = 2000;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    if (wide) hipLaunchKernelGGL(k_chain<1>, dim3(nw / 8), dim3(512), 0, 0, d_a, wpw, iters, L, d_out);
                    else hipLaunchKernelGGL(k_chain<0>, dim3(nw / 8), dim3(512), 0, 0, d_a, wpw, iters, L, d_out);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                }
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf(" \"%s_waves_per_cu_%d_lanes_%d\": {\"us_per_dependent_gather\": %.3f, \"G_lane_loads_per_s\": %.2f},\n", wide ? "dwordx4" : "dword", wpc, L,
                       ms * 1e3 / iters, (double)nw * L * iters / (ms * 1e-3) / 1e9);
            }
    printf(" \"note\": \"dependent chain of scattered gathers per wave, each wave inside its own 1.2 MB; 256 CUs\"\n}\n");
    return 0;
}
