// simd_placement.hip -- where do the waves of a workgroup land?  Workgroups of 256 threads in which ONE wave runs a dependent chain (the other three leave at
// once), 1 / 2 / 4 workgroups per CU: if the first wave of every workgroup is started on the same SIMD the chains of co-resident workgroups share its issue
// slots and the launch takes 2x / 4x as long; rotating the working wave with the workgroup index avoids that if wave i goes to SIMD i.
//   hipcc --offload-arch=gfx950 -O3 tools/simd_placement.hip -o /tmp/simd_placement && /tmp/simd_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_chain(uint32_t *out, int iters, int rotate, int *simd_of)
{
    __shared__ char pad[24 * 1024];      // (a band wave's LDS footprint: bounds the workgroups per CU like the real kernel)
    const int wv = threadIdx.x >> 6;
    const int worker = rotate ? (int)(blockIdx.x & 3) : 0;
    if (wv != worker) return;
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) simd_of[blockIdx.x] = (int)hwid;
    uint32_t a = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = a * 1664525u + 1013904223u;
    }
    if (a == 0x12345u) { out[0] = a; pad[threadIdx.x] = 1; }
}

int main()
{
    uint32_t *d_out; int *d_simd;
    CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_simd, 4096 * sizeof(int)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    static int h_simd[4096];
    printf("{\n");
    for (int rotate = 0; rotate <= 1; rotate++)
        for (int blocks : {64, 256, 512, 1024}) {
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, d_out, 1000, rotate, d_simd);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, d_out, 20000, rotate, d_simd);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h_simd, d_simd, blocks * sizeof(int), hipMemcpyDeviceToHost));
            int hist[4] = {0, 0, 0, 0};
            for (int b = 0; b < blocks; b++) hist[(h_simd[b] >> 4) & 3]++;     // HW_ID bits 5:4 = SIMD
            printf(" \"workgroups_%d_rotate_%d\": {\"ms\": %.3f, \"working_waves_per_simd_id\": [%d, %d, %d, %d]},\n", blocks, rotate, ms, hist[0], hist[1], hist[2], hist[3]);
        }
    printf(" \"note\": \"one working wave per 256-thread workgroup, 24 KB of LDS each; 256 CUs\"\n}\n");
    return 0;
}
