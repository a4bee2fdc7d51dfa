// Where does the hardware put the waves of k_lsd_regions2's launch shape (1024 workgroups of 8 waves, 38 KB of LDS each, 64 VGPRs: 4 workgroups per CU, 8 waves per SIMD)?
// Every wave records HW_ID / XCC_ID; the host prints which (workgroup, wave) pairs share a SIMD.   hipcc --offload-arch=gfx950 -O2 tools/wave_placement.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ void __launch_bounds__(512) k(unsigned *out, int spin)
{
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[(blockIdx.x * 8 + wave) * 2] = hw; out[(blockIdx.x * 8 + wave) * 2 + 1] = xcc;
    }
    // stay resident so that the whole grid is placed before anything retires
    volatile char *p = lds; float acc = 0.f;
    for (int i = 0; i < spin; i++) acc += __sinf((float)i + p[threadIdx.x]);
    if (acc == 123.456f) out[0] = 0;
}
int main()
{
    const int G = 1024;
    unsigned *d; hipMalloc(&d, G * 8 * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(k, dim3(G), dim3(512), 38 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 8 * 2); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> simd, cu;
    for (int i = 0; i < G * 8; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const unsigned simd_id = (hw >> 4) & 3, cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned long long cukey = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu_id;
        simd[(cukey << 2) | simd_id].push_back(i); cu[cukey].push_back(i);
    }
    printf("%zu CUs, %zu SIMDs used\n", cu.size(), simd.size());
    int shown = 0;
    for (auto &kv : cu) { if (shown++ >= 6) break; printf("CU %05llx: workgroups", kv.first); std::vector<int> w; for (int i : kv.second) w.push_back(i / 8); std::sort(w.begin(), w.end()); w.erase(std::unique(w.begin(), w.end()), w.end()); for (int x : w) printf(" %d", x); printf("\n"); }
    shown = 0;
    for (auto &kv : simd) { if (shown++ >= 12) break; printf("SIMD %06llx: (workgroup.wave)", kv.first); for (int i : kv.second) printf(" %d.%d", i / 8, i % 8); printf("\n"); }
    // histogram: which wave indices of a workgroup share a SIMD
    int pair[8][8] = {};
    for (auto &kv : simd) for (int a : kv.second) for (int b : kv.second) if (a / 8 == b / 8 && a != b) pair[a % 8][b % 8]++;
    printf("waves of one workgroup that share a SIMD (row = wave, col = other wave, count):\n");
    for (int a = 0; a < 8; a++) { for (int b = 0; b < 8; b++) printf(" %5d", pair[a][b]); printf("\n"); }
    // workgroup -> CU: difference between workgroup ids on one CU
    std::map<int, int> diff;
    for (auto &kv : cu) { std::vector<int> w; for (int i : kv.second) w.push_back(i / 8); std::sort(w.begin(), w.end()); w.erase(std::unique(w.begin(), w.end()), w.end()); for (size_t j = 1; j < w.size(); j++) diff[w[j] - w[j - 1]]++; }
    printf("differences between consecutive workgroup ids on one CU:"); for (auto &kv : diff) printf(" %d:%d", kv.first, kv.second); printf("\n");
    return 0;
}
