// valu_issue.hip -- micro-benchmark: how many wave64 VALU instructions per second does one MI355X retire, per instruction class and per
// number of resident waves per SIMD?  (DESIGN.md priced the step against "one wave64 VALU instruction per 4 cycles per
// SIMD"; /opt/skills/guides/MI355X_MICROARCH.md says 2 cycles on gfx950.)  Every wave runs ITERS iterations of 64 back-to-back instructions
// of one class on 8 independent accumulator registers (no instruction depends on one of the previous 7), written in inline asm so that the
// compiler can neither fold nor reorder them.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue > profiles/r03_valu_issue.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY64(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

// 32-bit classes: accumulators a0..a7 (VGPR), operands b, c
#define K32(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, uint32_t b, uint32_t c, unsigned long long *clk) \
    {                                                                                                      \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const unsigned long long t0 = clock64(), w0 = wall_clock64();                                      \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(BODY64(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                                  \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }                   \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;                              \
    }

#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
K32(k_add_u32, A_ADD)
K32(k_perm_b32, A_PERM)
K32(k_pk_max_i16, A_PKMAX)
K32(k_pk_add_u16, A_PKADD)
K32(k_fma_f32, A_FMA32)
K32(k_dot4_u32_u8, A_DOT4)
K32(k_mul_lo_u32, A_MULLO)
K32(k_alignbit_b32, A_ALIGNBIT)
K32(k_xor_b32, A_XOR)
K32(k_bcnt_u32, A_BCNT)
K32(k_cndmask_b32, A_CNDMASK)
K32(k_lshl_add_u32, A_LSHLADD)
K32(k_sad_u8, A_SAD)

#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_EXPF(i) "v_exp_f32 %" #i ", %" #i "\n"
#define A_RCPF(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
K32(k_and_b32, A_AND)
K32(k_lshlrev_b32, A_LSHL)
K32(k_min_u32, A_MIN)
K32(k_mul_u32_u24, A_MUL24)
K32(k_mad_u32_u24, A_MAD24)
K32(k_add3_u32, A_ADD3)
K32(k_bfe_u32, A_BFE)
K32(k_mov_b32, A_MOV)
K32(k_add_f32, A_ADDF)
K32(k_mul_f32, A_MULF)
K32(k_exp_f32, A_EXPF)
K32(k_rcp_f32, A_RCPF)
K32(k_cvt_f32_u32, A_CVTFU)
K32(k_cmp_lt_u32, A_CMP)

// VALU instructions with a SCALAR source (the compiler feeds wave-uniform values to VALU instructions this way all the time) and the select on a mask
#define KS32(NAME, ASM)                                                                                    \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, uint32_t b, uint32_t c, unsigned long long *clk) \
    {                                                                                                      \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long m = __ballot((threadIdx.x & 1) != 0);                                           \
        const unsigned long long t0 = clock64(), w0 = wall_clock64();                                      \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(BODY64(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "s"(m), "v"(c) : "vcc"); \
        }                                                                                                  \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }                   \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;                              \
    }
#define A_ADD_S(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define A_XOR_S(i) "v_xor_b32 %" #i ", %8, %" #i "\n"
#define A_FMA_S(i) "v_fma_f32 %" #i ", %" #i ", %8, %10\n"
#define A_CND_S(i) "v_cndmask_b32 %" #i ", %" #i ", %10, %9\n"
#define A_CND_VCC(i) "v_cndmask_b32 %" #i ", %" #i ", %10, vcc\n"
#define A_RDL(i) "v_readlane_b32 s20, %" #i ", 3\n"
KS32(k_add_u32_sgpr, A_ADD_S)
KS32(k_xor_b32_sgpr, A_XOR_S)
KS32(k_fma_f32_sgpr, A_FMA_S)
KS32(k_cndmask_sgprmask, A_CND_S)
#define A_CND_E64VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %10, vcc\n"
#define A_CMPCND_VCC(i) "v_cmp_lt_u32_e32 vcc, %10, %" #i "\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\n"
#define A_CMPCND_SGPR(i) "v_cmp_lt_u32_e64 s[20:21], %10, %" #i "\nv_cndmask_b32_e64 %" #i ", %" #i ", %10, s[20:21]\n"
#define A_ADDCO(i) "v_add_co_u32_e32 %" #i ", vcc, %10, %" #i "\n"
#define A_ADDC(i) "v_addc_co_u32_e32 %" #i ", vcc, %10, %" #i ", vcc\n"
#define A_ADDCO_SGPR(i) "v_add_co_u32_e64 %" #i ", s[20:21], %10, %" #i "\n"
#define KSX(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, uint32_t b, uint32_t c, unsigned long long *clk) \
    {                                                                                                      \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long m = __ballot((threadIdx.x & 1) != 0);                                           \
        const unsigned long long t0 = clock64(), w0 = wall_clock64();                                      \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(BODY64(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "s"(m), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                                  \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }                   \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;                              \
    }
KSX(k_cnd_e64_vcc, A_CND_E64VCC)
KSX(k_cmpcnd_vcc, A_CMPCND_VCC)
KSX(k_cmpcnd_sgpr, A_CMPCND_SGPR)
KSX(k_addco_vcc, A_ADDCO)
KSX(k_addc_vcc, A_ADDC)
KSX(k_addco_sgpr, A_ADDCO_SGPR)
#define A_CMP2CND_VCC(i) "v_cmp_lt_u32_e32 vcc, %10, %" #i "\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\n"
#define A_CMP4CND_VCC(i) "v_cmp_lt_u32_e32 vcc, %10, %" #i "\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\n"
#define A_CMP2CND_MIX(i) "v_cmp_lt_u32_e32 vcc, %10, %" #i "\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\nv_add_u32 %" #i ", %" #i ", %10\nv_cndmask_b32_e32 %" #i ", %" #i ", %10, vcc\n"
KSX(k_cmp2cnd_vcc, A_CMP2CND_VCC)
KSX(k_cmp4cnd_vcc, A_CMP4CND_VCC)
KSX(k_cmp2cnd_mix, A_CMP2CND_MIX)
__global__ void __launch_bounds__(1024) k_cndmask_vcc_init(uint32_t *out, int iters, uint32_t b, uint32_t c, unsigned long long *clk)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        asm volatile("s_mov_b64 vcc, 0x5555\n" BODY64(A_CND_VCC) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "s"(b), "v"(c) : "vcc");
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;
}
__global__ void __launch_bounds__(1024) k_readlane_b32(uint32_t *out, int iters, uint32_t b, uint32_t c, unsigned long long *clk)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        asm volatile(BODY64(A_RDL) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "s"(b), "v"(c) : "s20");
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;
}

// 64-bit classes: accumulators are VGPR pairs
#define K64(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, double b, double c, unsigned long long *clk)     \
    {                                                                                                      \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;   \
        const unsigned long long t0 = clock64(), w0 = wall_clock64();                                      \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(BODY64(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                                  \
        const unsigned long long t1 = clock64(), w1 = wall_clock64();                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }                   \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.12345) out[0] = 1;                                  \
    }
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
K64(k_add_f64, A_ADD64)
K64(k_fma_f64, A_FMA64)
K64(k_mul_f64, A_MUL64)
K64(k_pk_fma_f32, A_PKFMA32)
#define A_ADD64S(i) "v_add_f64 %" #i ", %" #i ", 1.0\n"
#define A_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_SQRT64(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define A_CVT64(i) "v_cvt_f32_f64 %" #i ", %" #i "\n"
K64(k_add_f64_const, A_ADD64S)
K64(k_rcp_f64, A_RCP64)
K64(k_sqrt_f64, A_SQRT64)

// scalar ALU for comparison (the region chain is ~45 % SALU): 8 independent SGPR accumulators
__global__ void __launch_bounds__(1024) k_s_add_u32(uint32_t *out, int iters, uint32_t b, uint32_t c, unsigned long long *clk)
{
    uint32_t a0 = __builtin_amdgcn_readfirstlane(b), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
#define A_SADD(i) "s_add_u32 %" #i ", %" #i ", %8\n"
    for (int i = 0; i < iters; i++) {
        asm volatile(BODY64(A_SADD) : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7) : "s"(c) : "scc");
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;
}

struct Case { const char *name; void *fn; bool f64; };

int main()
{
    hipDeviceProp_t P;
    CHECK(hipGetDeviceProperties(&P, 0));
    const int cus = P.multiProcessorCount, simds = cus * 4;
    uint32_t *d_out; unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_clk, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Case cases[] = {
        {"v_add_u32", (void *)k_add_u32, false}, {"v_xor_b32", (void *)k_xor_b32, false}, {"v_lshl_add_u32", (void *)k_lshl_add_u32, false},
        {"v_cndmask_b32", (void *)k_cndmask_b32, false}, {"v_perm_b32", (void *)k_perm_b32, false}, {"v_alignbit_b32", (void *)k_alignbit_b32, false},
        {"v_bcnt_u32_b32", (void *)k_bcnt_u32, false}, {"v_pk_max_i16", (void *)k_pk_max_i16, false}, {"v_pk_add_u16", (void *)k_pk_add_u16, false},
        {"v_dot4_u32_u8", (void *)k_dot4_u32_u8, false}, {"v_sad_u8", (void *)k_sad_u8, false}, {"v_mul_lo_u32", (void *)k_mul_lo_u32, false},
        {"v_fma_f32", (void *)k_fma_f32, false}, {"v_pk_fma_f32", (void *)k_pk_fma_f32, true}, {"v_add_f64", (void *)k_add_f64, true},
        {"v_mul_f64", (void *)k_mul_f64, true}, {"v_fma_f64", (void *)k_fma_f64, true}, {"s_add_u32", (void *)k_s_add_u32, false},
        {"v_and_b32", (void *)k_and_b32, false}, {"v_lshlrev_b32", (void *)k_lshlrev_b32, false}, {"v_min_u32", (void *)k_min_u32, false},
        {"v_mul_u32_u24", (void *)k_mul_u32_u24, false}, {"v_mad_u32_u24", (void *)k_mad_u32_u24, false}, {"v_add3_u32", (void *)k_add3_u32, false},
        {"v_bfe_u32", (void *)k_bfe_u32, false}, {"v_mov_b32", (void *)k_mov_b32, false}, {"v_add_f32", (void *)k_add_f32, false},
        {"v_mul_f32", (void *)k_mul_f32, false}, {"v_exp_f32", (void *)k_exp_f32, false}, {"v_rcp_f32", (void *)k_rcp_f32, false},
        {"v_cvt_f32_u32", (void *)k_cvt_f32_u32, false}, {"v_cmp_lt_u32 (vcc)", (void *)k_cmp_lt_u32, false},
        {"v_add_u32 (sgpr src)", (void *)k_add_u32_sgpr, false}, {"v_xor_b32 (sgpr src)", (void *)k_xor_b32_sgpr, false},
        {"v_fma_f32 (sgpr src)", (void *)k_fma_f32_sgpr, false}, {"v_cndmask_b32 (sgpr-pair mask)", (void *)k_cndmask_sgprmask, false},
        {"v_cndmask_b32 (vcc set once per 64)", (void *)k_cndmask_vcc_init, false}, {"v_readlane_b32", (void *)k_readlane_b32, false},
        {"v_cndmask_b32_e64 (vcc as explicit operand)", (void *)k_cnd_e64_vcc, false},
        {"PAIR v_cmp_lt_u32_e32 vcc + v_cndmask_b32_e32 vcc (per pair)", (void *)k_cmpcnd_vcc, false},
        {"PAIR v_cmp_lt_u32_e64 s[20:21] + v_cndmask_b32_e64 s[20:21] (per pair)", (void *)k_cmpcnd_sgpr, false},
        {"TRIPLE v_cmp_e32 vcc + 2 x v_cndmask_b32_e32 vcc (per triple)", (void *)k_cmp2cnd_vcc, false},
        {"QUINT v_cmp_e32 vcc + 4 x v_cndmask_b32_e32 vcc (per five)", (void *)k_cmp4cnd_vcc, false},
        {"QUAD v_cmp_e32 vcc + v_cndmask_e32 + v_add_u32 + v_cndmask_e32 (per four)", (void *)k_cmp2cnd_mix, false},
        {"v_add_co_u32_e32 (writes vcc)", (void *)k_addco_vcc, false}, {"v_addc_co_u32_e32 (reads + writes vcc)", (void *)k_addc_vcc, false},
        {"v_add_co_u32_e64 (writes s[20:21])", (void *)k_addco_sgpr, false},
        {"v_add_f64 (inline const)", (void *)k_add_f64_const, true}, {"v_rcp_f64", (void *)k_rcp_f64, true}, {"v_sqrt_f64", (void *)k_sqrt_f64, true},
    };
    const int waves_per_simd[] = {1, 2, 4, 8};
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"simds\": %d, \"clock_rate_khz\": %d,\n", P.name, P.gcnArchName, cus, simds, P.clockRate);
    printf(" \"method\": \"each wave: ITERS x 64 back-to-back instructions of one class on 8 independent accumulators (inline asm); grid = CUs x 4 SIMDs x W waves; "
           "rate = wave-instructions / HIP-event time; cycles_per_instr_per_simd = simds * measured_shader_clock / rate\",\n \"results\": [\n");
    bool first = true;
    for (const Case &c : cases) {
        for (int W : waves_per_simd) {
            const int block = 256 * (W < 4 ? W : 4);            // 4 / 8 / 16 waves per workgroup: the waves of a workgroup spread over the CU's 4 SIMDs
            const int grid = cus * (W <= 4 ? 1 : W / 4);
            const int iters = 20000 / W;
            uint32_t b = 0x01020304u, cc = 0x07060504u; double bd = 1.0000001, cd = 1e-9;
            void *a32[] = {&d_out, (void *)&iters, &b, &cc, &d_clk};
            void *a64[] = {&d_out, (void *)&iters, &bd, &cd, &d_clk};
            void **args = c.f64 ? a64 : a32;
            CHECK(hipLaunchKernel(c.fn, dim3(grid), dim3(block), args, 0, 0));   // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CHECK(hipEventRecord(e0, 0));
                CHECK(hipLaunchKernel(c.fn, dim3(grid), dim3(block), args, 0, 0));
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            unsigned long long clk[2];
            CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
            const double waves = (double)grid * block / 64.0, instr = waves * (double)iters * 64.0;
            const double rate = instr / (best * 1e-3);
            // shader clock measured by one wave: clock64() ticks per wall_clock64() tick (100 MHz)
            const double shader_hz = clk[1] ? (double)clk[0] / (double)clk[1] * 1e8 : 0.0;
            const double cyc_wave = (double)clk[0] / ((double)iters * 64.0);   // cycles one wave spent per instruction (W waves share the SIMD)
            printf("%s  {\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4e, \"clock64_hz\": %.4e, "
                   "\"cycles_per_instr_per_simd_at_clock64\": %.3f, \"cycles_per_instr_per_simd_at_2.4GHz\": %.3f, \"wave_cycles_per_instr\": %.3f}",
                   first ? "" : ",\n", c.name, W, best, rate, shader_hz, shader_hz > 0 ? simds * shader_hz / rate : 0.0, simds * 2.4e9 / rate, cyc_wave);
            first = false;
        }
    }
    printf("\n ]}\n");
    return 0;
}
