// valu_issue2.hip -- round 6 addendum to valu_issue.hip: per-SIMD issue interval of the instruction forms the round-6 rewrites of k_orb_level choose between
// (fp32 column sums + v_cvt_pk_u8_f32 packing against mul24 / bfe / min / lshl_or chains; SDWA byte operands; 3-input integer forms).  8 waves per SIMD only.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue2.hip -o /tmp/valu_issue2 && /tmp/valu_issue2 > profiles/r06_valu_issue2.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY64(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)
#define K32(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, uint32_t b, uint32_t c)        \
    {                                                                                                      \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(BODY64(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                                  \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;                              \
    }
#define X(NAME, STR) K32(NAME, STR)
#define I(i) "%" #i
#define A1(i) "v_sub_u32 " I(i) ", " I(i) ", %8\n"
#define A2(i) "v_or_b32 " I(i) ", " I(i) ", %8\n"
#define A3(i) "v_min_f32 " I(i) ", " I(i) ", %8\n"
#define A4(i) "v_max_f32 " I(i) ", " I(i) ", %8\n"
#define A5(i) "v_sub_f32 " I(i) ", " I(i) ", %8\n"
#define A6(i) "v_cvt_pk_u8_f32 " I(i) ", %8, 1, " I(i) "\n"
#define A7(i) "v_cvt_f32_ubyte0 " I(i) ", " I(i) "\n"
#define A8(i) "v_cvt_u32_f32 " I(i) ", " I(i) "\n"
#define A9(i) "v_lshl_or_b32 " I(i) ", " I(i) ", 8, %8\n"
#define A10(i) "v_and_or_b32 " I(i) ", " I(i) ", %8, %9\n"
#define A11(i) "v_or3_b32 " I(i) ", " I(i) ", %8, %9\n"
#define A12(i) "v_bfi_b32 " I(i) ", " I(i) ", %8, %9\n"
#define A13(i) "v_dot2_i32_i16 " I(i) ", %8, %9, " I(i) "\n"
#define A14(i) "v_dot2_u32_u16 " I(i) ", %8, %9, " I(i) "\n"
#define A15(i) "v_mul_hi_u32 " I(i) ", " I(i) ", %8\n"
#define A16(i) "v_mad_i32_i24 " I(i) ", " I(i) ", %8, %9\n"
#define A17(i) "v_fmac_f32 " I(i) ", %8, %9\n"
#define A18(i) "v_add_u32 " I(i) ", 5, " I(i) "\n"
#define A19(i) "v_add_u32 " I(i) ", 0x7fff, " I(i) "\n"
#define A20(i) "v_and_b32 " I(i) ", 0x1ff, " I(i) "\n"
#define A21(i) "v_med3_i32 " I(i) ", " I(i) ", %8, %9\n"
#define A22(i) "v_min3_i32 " I(i) ", " I(i) ", %8, %9\n"
#define A23(i) "v_pk_min_i16 " I(i) ", " I(i) ", %8\n"
#define A24(i) "v_pk_sub_i16 " I(i) ", " I(i) ", %8\n"
#define A25(i) "v_pk_mul_lo_u16 " I(i) ", " I(i) ", %8\n"
#define A26(i) "v_pk_mad_u16 " I(i) ", " I(i) ", %8, %9\n"
#define A27(i) "v_add_u32_sdwa " I(i) ", " I(i) ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define A28(i) "v_sub_u32_sdwa " I(i) ", " I(i) ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2\n"
#define A29(i) "v_not_b32 " I(i) ", " I(i) "\n"
#define A30(i) "v_ashrrev_i32 " I(i) ", 4, " I(i) "\n"
#define A31(i) "v_lshrrev_b32 " I(i) ", 16, " I(i) "\n"
#define A32(i) "v_fma_f32 " I(i) ", " I(i) ", %8, 1.0\n"
#define A33(i) "v_fmaak_f32 " I(i) ", " I(i) ", %8, 0x4b000000\n"
#define A34(i) "v_add_f32 " I(i) ", 0x4b000000, " I(i) "\n"
#define A35(i) "v_max_i32 " I(i) ", " I(i) ", %8\n"
#define A36(i) "v_max3_i32 " I(i) ", " I(i) ", %8, %9\n"
#define A37(i) "v_mul_i32_i24 " I(i) ", " I(i) ", %8\n"
#define A38(i) "v_xad_u32 " I(i) ", " I(i) ", %8, %9\n"
#define A39(i) "v_add_lshl_u32 " I(i) ", " I(i) ", %8, 2\n"
#define A40(i) "v_alignbyte_b32 " I(i) ", " I(i) ", %8, 1\n"
#define A41(i) "v_cvt_f32_i32 " I(i) ", " I(i) "\n"
#define A42(i) "v_rndne_f32 " I(i) ", " I(i) "\n"
#define A43(i) "v_mul_f32 " I(i) ", 0x37800000, " I(i) "\n"
#define A44(i) "v_subrev_u32 " I(i) ", " I(i) ", %8\n"
#define A45(i) "v_sad_u16 " I(i) ", " I(i) ", %8, %9\n"
#define A46(i) "v_msad_u8 " I(i) ", " I(i) ", %8, %9\n"
#define A47(i) "v_cmp_gt_i32 vcc, " I(i) ", %8\n"
#define A48(i) "v_cmp_gt_i32 s[20:21], " I(i) ", %8\n"
#define A49(i) "v_pk_add_f32 " I(i) ", " I(i) ", " I(i) "\n"
X(k1, A1) X(k2, A2) X(k3, A3) X(k4, A4) X(k5, A5) X(k6, A6) X(k7, A7) X(k8, A8) X(k9, A9) X(k10, A10) X(k11, A11) X(k12, A12) X(k13, A13) X(k14, A14) X(k15, A15)
X(k16, A16) X(k17, A17) X(k18, A18) X(k19, A19) X(k20, A20) X(k21, A21) X(k22, A22) X(k23, A23) X(k24, A24) X(k25, A25) X(k26, A26) X(k27, A27) X(k28, A28) X(k29, A29)
X(k30, A30) X(k31, A31) X(k32, A32) X(k33, A33) X(k34, A34) X(k35, A35) X(k36, A36) X(k37, A37) X(k38, A38) X(k39, A39) X(k40, A40) X(k41, A41) X(k42, A42) X(k43, A43)
X(k44, A44) X(k45, A45) X(k46, A46) X(k47, A47)
__global__ void __launch_bounds__(1024) k48(uint32_t *out, int iters, uint32_t b, uint32_t c)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; i++) {
        asm volatile(BODY64(A48) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;
}
struct Case { const char *name; void *fn; };
int main()
{
    hipDeviceProp_t P;
    CHECK(hipGetDeviceProperties(&P, 0));
    const int cus = P.multiProcessorCount, simds = cus * 4;
    uint32_t *d_out; CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Case cases[] = {
        {"v_sub_u32", (void *)k1}, {"v_or_b32", (void *)k2}, {"v_min_f32", (void *)k3}, {"v_max_f32", (void *)k4}, {"v_sub_f32", (void *)k5}, {"v_cvt_pk_u8_f32", (void *)k6},
        {"v_cvt_f32_ubyte0", (void *)k7}, {"v_cvt_u32_f32", (void *)k8}, {"v_lshl_or_b32", (void *)k9}, {"v_and_or_b32", (void *)k10}, {"v_or3_b32", (void *)k11},
        {"v_bfi_b32", (void *)k12}, {"v_dot2_i32_i16", (void *)k13}, {"v_dot2_u32_u16", (void *)k14}, {"v_mul_hi_u32", (void *)k15}, {"v_mad_i32_i24", (void *)k16},
        {"v_fmac_f32", (void *)k17}, {"v_add_u32 (inline const 5)", (void *)k18}, {"v_add_u32 (literal 0x7fff)", (void *)k19}, {"v_and_b32 (literal 0x1ff)", (void *)k20},
        {"v_med3_i32", (void *)k21}, {"v_min3_i32", (void *)k22}, {"v_pk_min_i16", (void *)k23}, {"v_pk_sub_i16", (void *)k24}, {"v_pk_mul_lo_u16", (void *)k25},
        {"v_pk_mad_u16", (void *)k26}, {"v_add_u32_sdwa (src1 BYTE_1)", (void *)k27}, {"v_sub_u32_sdwa (BYTE_0, BYTE_2)", (void *)k28}, {"v_not_b32", (void *)k29},
        {"v_ashrrev_i32", (void *)k30}, {"v_lshrrev_b32", (void *)k31}, {"v_fma_f32 (inline const 1.0)", (void *)k32}, {"v_fmaak_f32 (literal)", (void *)k33},
        {"v_add_f32 (literal)", (void *)k34}, {"v_max_i32", (void *)k35}, {"v_max3_i32", (void *)k36}, {"v_mul_i32_i24", (void *)k37}, {"v_xad_u32", (void *)k38},
        {"v_add_lshl_u32", (void *)k39}, {"v_alignbyte_b32", (void *)k40}, {"v_cvt_f32_i32", (void *)k41}, {"v_rndne_f32", (void *)k42}, {"v_mul_f32 (literal)", (void *)k43},
        {"v_subrev_u32", (void *)k44}, {"v_sad_u16", (void *)k45}, {"v_msad_u8", (void *)k46}, {"v_cmp_gt_i32 (vcc)", (void *)k47}, {"v_cmp_gt_i32 (sgpr pair)", (void *)k48},
    };
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"simds\": %d, \"waves_per_simd\": 8,\n \"method\": \"as tools/valu_issue.hip: 64 back-to-back instructions of one form "
           "on 8 independent accumulators, 8 waves per SIMD on every SIMD; cycles = simds * 2.4e9 / (wave-instructions per second)\",\n \"results\": {\n", P.name, P.gcnArchName, cus, simds);
    bool first = true;
    for (const Case &c : cases) {
        const int block = 1024, grid = cus * 2, iters = 2500;
        uint32_t b = 0x01020304u, cc = 0x07060504u;
        void *args[] = {&d_out, (void *)&iters, &b, &cc};
        CHECK(hipLaunchKernel(c.fn, dim3(grid), dim3(block), args, 0, 0));
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
        const double instr = (double)grid * block / 64.0 * (double)iters * 64.0, rate = instr / (best * 1e-3);
        printf("%s  \"%s\": %.2f", first ? "" : ",\n", c.name, simds * 2.4e9 / rate);
        first = false;
    }
    printf("\n }}\n");
    return 0;
}
