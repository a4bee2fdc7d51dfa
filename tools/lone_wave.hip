// lone_wave.hip -- what ONE wave per SIMD pays per instruction for the patterns k_lsd_regions8 is made of (dependent chains, VALU <-> SALU hand-offs, taken
// branches, DPP, LDS round trips).  1024 workgroups of 64 threads (one wave per SIMD), cycles from s_memtime around N repetitions of each pattern.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lone_wave tools/lone_wave.hip && /tmp/lone_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP 256
#define STR2(x) #x
#define STR(x) STR2(x)
#define PAT(name, init, body)                                                                                          \
    __global__ void __launch_bounds__(64) name(long long *out, float seed)                                             \
    {                                                                                                                  \
        __shared__ float lds[256];                                                                                     \
        lds[threadIdx.x] = seed;                                                                                       \
        float a = seed + threadIdx.x, b = seed * 2.f, c = 1.f, d = 2.f;                                                \
        unsigned ldsaddr = threadIdx.x * 4;                                                                            \
        (void)ldsaddr;                                                                                                 \
        init;                                                                                                          \
        long long t0 = __builtin_readcyclecounter();                                                                   \
        for (int it = 0; it < 64; it++) {                                                                              \
            asm volatile(".rept " STR(REP) "\n" body "\n.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(ldsaddr) : "vcc", "s20", "s21", "s22", "s23", "scc", "memory"); \
        }                                                                                                              \
        long long t1 = __builtin_readcyclecounter();                                                                   \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                               \
        if (a + b + c + d == 12345.f) out[0] = 0;                                                                      \
    }

PAT(k_dep_add, , "v_add_f32 %0, %0, %1")
PAT(k_indep_add, , "v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1")
PAT(k_dep_f64ish, , "v_cvt_f64_f32 v[10:11], %0\n v_add_f64 v[10:11], v[10:11], v[10:11]\n v_cvt_f32_f64 %0, v[10:11]")
PAT(k_cmp_sand_cnd, , "v_cmp_lt_f32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], exec\n v_cndmask_b32 %0, %0, %2, s[20:21]")
PAT(k_cmp_cnd, , "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
PAT(k_salu_dep, , "s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[20:21], s[20:21], s[22:23]")
PAT(k_dpp_dep, , "s_nop 1\n v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
PAT(k_branch_taken, , "s_cbranch_scc1 1f\n s_nop 0\n 1: v_add_f32 %0, %0, %1\n s_cmp_eq_u32 s20, s20")
PAT(k_branch_not, , "s_cmp_lg_u32 s20, s20\n s_cbranch_scc1 1f\n 1: v_add_f32 %0, %0, %1")
PAT(k_lds_rt, , "ds_read_b32 %0, %4\n s_waitcnt lgkmcnt(0)")
PAT(k_bperm_rt, , "ds_bpermute_b32 %0, %4, %0\n s_waitcnt lgkmcnt(0)")
PAT(k_readlane_use, , "v_readlane_b32 s20, %0, 3\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0")
PAT(k_vcmp_scmp_br, , "v_cmp_lt_f32 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 1f\n 1: v_add_f32 %0, %0, %1")
PAT(k_saveexec, , "v_cmp_lt_f32 vcc, %1, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]")

int main()
{
    long long *d;
    hipMalloc(&d, 1024 * 8);
    std::vector<long long> h(1024);
    struct { const char *name; void (*k)(long long *, float); int ninst; } T[] = {
        {"dependent v_add_f32 chain", k_dep_add, 1}, {"3 independent v_add_f32", k_indep_add, 3}, {"cvt f64 / add f64 / cvt f32 chain", k_dep_f64ish, 3},
        {"v_cmp -> s_and -> v_cndmask chain", k_cmp_sand_cnd, 3}, {"v_cmp vcc -> v_cndmask chain", k_cmp_cnd, 2}, {"dependent s_and / s_or", k_salu_dep, 2},
        {"dependent v_or_b32_dpp (+s_nop 1)", k_dpp_dep, 2}, {"taken s_cbranch + v_add + s_cmp", k_branch_taken, 3}, {"not-taken s_cbranch + s_cmp + v_add", k_branch_not, 3},
        {"ds_read_b32 round trip", k_lds_rt, 1}, {"ds_bpermute round trip", k_bperm_rt, 1}, {"v_readlane -> s_add -> v_add", k_readlane_use, 3},
        {"v_cmp -> s_cmp vcc -> branch(not taken) -> v_add", k_vcmp_scmp_br, 4}, {"v_cmp -> saveexec -> v_add -> restore", k_saveexec, 4}};
    for (auto &t : T) {
        for (int waves = 1; waves <= 2; waves++) {
            hipLaunchKernelGGL(t.k, dim3(1024 * waves), dim3(64), 0, 0, d, 1.5f);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < 1024; i++) s += (double)h[i];
            s /= 1024.0 * 64 * REP;
            printf("%-52s %d wave(s)/SIMD: %7.1f cycles per repetition (%d instr) = %5.1f per instruction\n", t.name, waves, s, t.ninst, s / t.ninst);
        }
    }
    return 0;
}
