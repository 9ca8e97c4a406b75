// VALU issue-rate microbenchmark for gfx950 (round 3 rewrite; the round-2 version was flawed: sub-millisecond kernels from an idle GPU,
// cycles taken from the NOMINAL clock, one-wave blocks whose SIMD placement was the dispatcher's, loop overhead charged to the FMAs).
//
// Question: how many SIMD cycles does one wave64 VALU instruction occupy the issue port, with k = 1..8 waves resident per SIMD?
//   /opt/skills/guides/MI355X_MICROARCH.md says 2 (several waves; 4 for a lone wave); round 2 measured 3.0-3.6 and concluded that wf_extend
//   was at the issue limit. This version:
//   * 256-thread blocks, k blocks per CU (grid = CUs * k): a block's four waves go to the four SIMDs of one CU, so every SIMD holds exactly k
//     waves IF the dispatcher places k blocks on every CU -- verified, not assumed: every wave records HW_REG_HW_ID / XCC_ID and the host
//     prints the census of waves per (xcc, se, cu, simd);
//   * kernels of >= 50 ms after a warm-up of the same kernel (clock ramped, DVFS settled), iteration count calibrated per case;
//   * cycles two ways: in-kernel s_memtime (shader-clock ticks) around the loop, per wave (median / max over waves), and wall time x the
//     MEASURED clock = delta s_memtime / delta s_memrealtime x 100 MHz of the same launch (guide, "DVFS give-back" item 6);
//   * 128 independent-enough instructions per loop trip (16 accumulators x 8 rounds) against 2-3 SALU loop instructions: < 2.5 % overhead.
// Also times the other VALU forms the traversal kernel is made of (VOP2 add / mul / min / max, v_cndmask, v_cmp, integer ops, v_pk_fma_f32,
// v_rcp_f32) so that an instruction budget can be priced per opcode class.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run: ./valu_issue [ms per case, default 60]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct WaveRecord { unsigned long long ticks, realticks; unsigned int hwid, xcc; };

// 16 accumulators a0..a15; OP16 expands to 16 instructions, one per accumulator
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define ACC_OPERANDS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])

enum Op { OP_FMA, OP_FMAC, OP_ADD, OP_SUB, OP_MUL, OP_MAX, OP_MIN3, OP_MED3, OP_CNDMASK, OP_CNDMASK_SWAP, OP_CNDMASK_SGPR, OP_CMP, OP_CMP_SGPR, OP_CMP_CNDMASK, OP_CMP_4CND, OP_CMPS_4CND, OP_CMP_4CND_DEP, OP_CMP_3FMA_CND, OP_CND_FMA_ALT, OP_CMP_CND_FMA_CND, OP_CMP_CND_3FMA_CND, OP_CND_E64_VCC, OP_CND_NOP, OP_CND_E32_E64_ALT, OP_CMP_4CND_E64VCC, OP_CMP_CND_NOP4, OP_DIVFMAS, OP_ADDC, OP_CND2_FMA2, OP_CND3_FMA, OP_CND2_MAX2, OP_CMP_CND2_FMA, OP_AND, OP_OR, OP_XOR, OP_LSHL, OP_ADDU, OP_SUBU, OP_MINU, OP_MAXI, OP_LSHL_ADD, OP_AND_OR, OP_BFI, OP_BFE, OP_PERM, OP_MUL24, OP_MAD24, OP_MULLO, OP_CVT, OP_MOV, OP_MOV_DPP, OP_READLANE, OP_PKFMA, OP_PKMUL, OP_RCP, OP_MIX_FMA_MAX, OP_MIX_FMA_CMP, OP_DSREAD, OP_BITOP3, OP_ASHR, OP_MIX_BITOP3_MAX, OP_MIX_AND_MAX, OP_MIX_ADDU_MAX, OP_MIX_MULF_MAX, OP_MIX_XOR_CND, OP_COUNT };
static const char* kOpName[OP_COUNT] = {
    "v_fma_f32 (VOP3, 3 VGPR srcs)",
    "v_fmac_f32 (VOP2)",
    "v_add_f32 (VOP2)",
    "v_sub_f32 (VOP2)",
    "v_mul_f32 (VOP2)",
    "v_max_f32 (VOP2)",
    "v_min3_f32 (VOP3)",
    "v_med3_f32 (VOP3)",
    "v_cndmask_b32 dst=src0, vcc (VOP2)",
    "v_cndmask_b32 dst=src1, vcc (VOP2)",
    "v_cndmask_b32 e64, mask in s[20:21]",
    "v_cmp_lt_f32 -> vcc (VOPC)",
    "v_cmp_lt_f32 -> s[20:21] (VOP3)",
    "pair: v_cmp_lt_f32 vcc + v_cndmask vcc (per INSTRUCTION)",
    "1 v_cmp vcc + 4 independent v_cndmask vcc (per INSTRUCTION; 120 per trip)",
    "1 v_cmp_e64 s[20:21] + 4 independent v_cndmask_e64 (per INSTRUCTION; 120)",
    "cswap as compiled: v_cmp vcc + 4 v_cndmask vcc, next cmp reads them (120)",
    "v_cmp vcc; 2 x v_fma; v_cndmask vcc (x4 per block = 16)",
    "no cmp: v_cndmask vcc; v_fma alternating (16)",
    "v_cmp vcc; v_cndmask vcc; v_fma; v_cndmask vcc (x4 = 16)",
    "v_cmp vcc; v_cndmask vcc; 3 x v_fma; v_cndmask vcc; 2 x v_fma (x2 = 16)",
    "v_cndmask_b32_e64 ..., vcc back to back (VOP3 encoding, mask = vcc)",
    "v_cndmask_b32_e32 vcc; s_nop 0 alternating (per VALU instruction: 64 per trip)",
    "v_cndmask_e32 vcc; v_cndmask_e64 s[20:21] alternating",
    "1 v_cmp vcc + 4 v_cndmask_b32_e64 ..., vcc (120 per trip)",
    "1 v_cmp vcc + 4 x (v_cndmask_e32 vcc; s_nop 0) (per VALU instruction: 120 per trip)",
    "v_div_fmas_f32 back to back (reads vcc)",
    "v_addc_co_u32 back to back (reads + writes vcc)",
    "2 x v_cndmask_e32 vcc; 2 x v_fma (x4 = 16)",
    "3 x v_cndmask_e32 vcc; 1 x v_fma (x4 = 16)",
    "2 x v_cndmask_e32 vcc; 2 x v_max_f32 (x4 = 16)",
    "v_cmp vcc; 2 x v_cndmask_e32 vcc; v_fma (x4 = 16)",
    "v_and_b32 (VOP2)",
    "v_or_b32 (VOP2)",
    "v_xor_b32 (VOP2)",
    "v_lshlrev_b32 (VOP2)",
    "v_add_u32 (VOP2)",
    "v_sub_u32 (VOP2)",
    "v_min_u32 (VOP2)",
    "v_max_i32 (VOP2)",
    "v_lshl_add_u32 (VOP3)",
    "v_and_or_b32 (VOP3)",
    "v_bfi_b32 (VOP3)",
    "v_bfe_u32 (VOP3)",
    "v_perm_b32 (VOP3)",
    "v_mul_u32_u24 (VOP2)",
    "v_mad_u32_u24 (VOP3)",
    "v_mul_lo_u32 (VOP3)",
    "v_cvt_f32_u32 (VOP1)",
    "v_mov_b32 (VOP1)",
    "v_mov_b32 dpp row_shr:1",
    "v_readfirstlane_b32 -> s20",
    "v_pk_fma_f32 (VOP3P, 2 x 64 lanes)",
    "v_pk_mul_f32 (VOP3P)",
    "v_rcp_f32 (transcendental)",
    "mix: 8 x (v_fma_f32, v_max_f32) alternating",
    "mix: 8 x (v_fma_f32, v_cmp_lt_f32) alternating",
    "ds_read_b32 (per-lane column, conflict-free) + waitcnt per 16",
    "v_bitop3_b32 (VOP3, three-input bit operation 0x6c)",
    "v_ashrrev_i32 (VOP2)",
    "mix: 8 x (v_bitop3_b32, v_max_f32) alternating",
    "mix: 8 x (v_and_b32, v_max_f32) alternating",
    "mix: 8 x (v_add_u32, v_max_f32) alternating",
    "mix: 8 x (v_mul_f32, v_max_f32) alternating",
    "mix: 8 x (v_xor_b32, v_cndmask_b32_e64 s[20:21]) alternating" };

template <int OP>
__global__ __launch_bounds__(256) void issue_loop(WaveRecord* rec, float* sink, int iters, float fb, float fc)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    float a[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = (float)(threadIdx.x + i) * 1.0e-3f; p[i] = v2f{ a[i], a[i] + 1.0f }; }
    const float b = fb, c = fc;
    const v2f pb = { fb, fb }, pc = { fc, fc };
    unsigned int ib = __float_as_uint(fb) | 1u, ic = __float_as_uint(fc) | 0x10u;
    __shared__ float ldsBuf[16 * 256];
    ldsBuf[threadIdx.x] = fb;
    const unsigned int ldsAddr = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) float*)ldsBuf + threadIdx.x * 4u;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int round = 0; round < 8; ++round) {
            if constexpr (OP == OP_FMA) {
                asm volatile("v_fma_f32 %0, %0, %16, %17\n" "v_fma_f32 %1, %1, %16, %17\n" "v_fma_f32 %2, %2, %16, %17\n" "v_fma_f32 %3, %3, %16, %17\n" "v_fma_f32 %4, %4, %16, %17\n" "v_fma_f32 %5, %5, %16, %17\n" "v_fma_f32 %6, %6, %16, %17\n" "v_fma_f32 %7, %7, %16, %17\n" "v_fma_f32 %8, %8, %16, %17\n" "v_fma_f32 %9, %9, %16, %17\n" "v_fma_f32 %10, %10, %16, %17\n" "v_fma_f32 %11, %11, %16, %17\n" "v_fma_f32 %12, %12, %16, %17\n" "v_fma_f32 %13, %13, %16, %17\n" "v_fma_f32 %14, %14, %16, %17\n" "v_fma_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_FMAC) {
                asm volatile("v_fmac_f32 %0, %16, %17\n" "v_fmac_f32 %1, %16, %17\n" "v_fmac_f32 %2, %16, %17\n" "v_fmac_f32 %3, %16, %17\n" "v_fmac_f32 %4, %16, %17\n" "v_fmac_f32 %5, %16, %17\n" "v_fmac_f32 %6, %16, %17\n" "v_fmac_f32 %7, %16, %17\n" "v_fmac_f32 %8, %16, %17\n" "v_fmac_f32 %9, %16, %17\n" "v_fmac_f32 %10, %16, %17\n" "v_fmac_f32 %11, %16, %17\n" "v_fmac_f32 %12, %16, %17\n" "v_fmac_f32 %13, %16, %17\n" "v_fmac_f32 %14, %16, %17\n" "v_fmac_f32 %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_ADD) {
                asm volatile("v_add_f32 %0, %0, %16\n" "v_add_f32 %1, %1, %16\n" "v_add_f32 %2, %2, %16\n" "v_add_f32 %3, %3, %16\n" "v_add_f32 %4, %4, %16\n" "v_add_f32 %5, %5, %16\n" "v_add_f32 %6, %6, %16\n" "v_add_f32 %7, %7, %16\n" "v_add_f32 %8, %8, %16\n" "v_add_f32 %9, %9, %16\n" "v_add_f32 %10, %10, %16\n" "v_add_f32 %11, %11, %16\n" "v_add_f32 %12, %12, %16\n" "v_add_f32 %13, %13, %16\n" "v_add_f32 %14, %14, %16\n" "v_add_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_SUB) {
                asm volatile("v_sub_f32 %0, %0, %16\n" "v_sub_f32 %1, %1, %16\n" "v_sub_f32 %2, %2, %16\n" "v_sub_f32 %3, %3, %16\n" "v_sub_f32 %4, %4, %16\n" "v_sub_f32 %5, %5, %16\n" "v_sub_f32 %6, %6, %16\n" "v_sub_f32 %7, %7, %16\n" "v_sub_f32 %8, %8, %16\n" "v_sub_f32 %9, %9, %16\n" "v_sub_f32 %10, %10, %16\n" "v_sub_f32 %11, %11, %16\n" "v_sub_f32 %12, %12, %16\n" "v_sub_f32 %13, %13, %16\n" "v_sub_f32 %14, %14, %16\n" "v_sub_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MUL) {
                asm volatile("v_mul_f32 %0, %0, %16\n" "v_mul_f32 %1, %1, %16\n" "v_mul_f32 %2, %2, %16\n" "v_mul_f32 %3, %3, %16\n" "v_mul_f32 %4, %4, %16\n" "v_mul_f32 %5, %5, %16\n" "v_mul_f32 %6, %6, %16\n" "v_mul_f32 %7, %7, %16\n" "v_mul_f32 %8, %8, %16\n" "v_mul_f32 %9, %9, %16\n" "v_mul_f32 %10, %10, %16\n" "v_mul_f32 %11, %11, %16\n" "v_mul_f32 %12, %12, %16\n" "v_mul_f32 %13, %13, %16\n" "v_mul_f32 %14, %14, %16\n" "v_mul_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MAX) {
                asm volatile("v_max_f32 %0, %0, %16\n" "v_max_f32 %1, %1, %16\n" "v_max_f32 %2, %2, %16\n" "v_max_f32 %3, %3, %16\n" "v_max_f32 %4, %4, %16\n" "v_max_f32 %5, %5, %16\n" "v_max_f32 %6, %6, %16\n" "v_max_f32 %7, %7, %16\n" "v_max_f32 %8, %8, %16\n" "v_max_f32 %9, %9, %16\n" "v_max_f32 %10, %10, %16\n" "v_max_f32 %11, %11, %16\n" "v_max_f32 %12, %12, %16\n" "v_max_f32 %13, %13, %16\n" "v_max_f32 %14, %14, %16\n" "v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIN3) {
                asm volatile("v_min3_f32 %0, %0, %16, %17\n" "v_min3_f32 %1, %1, %16, %17\n" "v_min3_f32 %2, %2, %16, %17\n" "v_min3_f32 %3, %3, %16, %17\n" "v_min3_f32 %4, %4, %16, %17\n" "v_min3_f32 %5, %5, %16, %17\n" "v_min3_f32 %6, %6, %16, %17\n" "v_min3_f32 %7, %7, %16, %17\n" "v_min3_f32 %8, %8, %16, %17\n" "v_min3_f32 %9, %9, %16, %17\n" "v_min3_f32 %10, %10, %16, %17\n" "v_min3_f32 %11, %11, %16, %17\n" "v_min3_f32 %12, %12, %16, %17\n" "v_min3_f32 %13, %13, %16, %17\n" "v_min3_f32 %14, %14, %16, %17\n" "v_min3_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MED3) {
                asm volatile("v_med3_f32 %0, %0, %16, %17\n" "v_med3_f32 %1, %1, %16, %17\n" "v_med3_f32 %2, %2, %16, %17\n" "v_med3_f32 %3, %3, %16, %17\n" "v_med3_f32 %4, %4, %16, %17\n" "v_med3_f32 %5, %5, %16, %17\n" "v_med3_f32 %6, %6, %16, %17\n" "v_med3_f32 %7, %7, %16, %17\n" "v_med3_f32 %8, %8, %16, %17\n" "v_med3_f32 %9, %9, %16, %17\n" "v_med3_f32 %10, %10, %16, %17\n" "v_med3_f32 %11, %11, %16, %17\n" "v_med3_f32 %12, %12, %16, %17\n" "v_med3_f32 %13, %13, %16, %17\n" "v_med3_f32 %14, %14, %16, %17\n" "v_med3_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CNDMASK) {
                asm volatile("v_cndmask_b32 %0, %0, %16, vcc\n" "v_cndmask_b32 %1, %1, %16, vcc\n" "v_cndmask_b32 %2, %2, %16, vcc\n" "v_cndmask_b32 %3, %3, %16, vcc\n" "v_cndmask_b32 %4, %4, %16, vcc\n" "v_cndmask_b32 %5, %5, %16, vcc\n" "v_cndmask_b32 %6, %6, %16, vcc\n" "v_cndmask_b32 %7, %7, %16, vcc\n" "v_cndmask_b32 %8, %8, %16, vcc\n" "v_cndmask_b32 %9, %9, %16, vcc\n" "v_cndmask_b32 %10, %10, %16, vcc\n" "v_cndmask_b32 %11, %11, %16, vcc\n" "v_cndmask_b32 %12, %12, %16, vcc\n" "v_cndmask_b32 %13, %13, %16, vcc\n" "v_cndmask_b32 %14, %14, %16, vcc\n" "v_cndmask_b32 %15, %15, %16, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CNDMASK_SWAP) {
                asm volatile("v_cndmask_b32 %0, %16, %0, vcc\n" "v_cndmask_b32 %1, %16, %1, vcc\n" "v_cndmask_b32 %2, %16, %2, vcc\n" "v_cndmask_b32 %3, %16, %3, vcc\n" "v_cndmask_b32 %4, %16, %4, vcc\n" "v_cndmask_b32 %5, %16, %5, vcc\n" "v_cndmask_b32 %6, %16, %6, vcc\n" "v_cndmask_b32 %7, %16, %7, vcc\n" "v_cndmask_b32 %8, %16, %8, vcc\n" "v_cndmask_b32 %9, %16, %9, vcc\n" "v_cndmask_b32 %10, %16, %10, vcc\n" "v_cndmask_b32 %11, %16, %11, vcc\n" "v_cndmask_b32 %12, %16, %12, vcc\n" "v_cndmask_b32 %13, %16, %13, vcc\n" "v_cndmask_b32 %14, %16, %14, vcc\n" "v_cndmask_b32 %15, %16, %15, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CNDMASK_SGPR) {
                asm volatile("v_cndmask_b32 %0, %0, %16, s[20:21]\n" "v_cndmask_b32 %1, %1, %16, s[20:21]\n" "v_cndmask_b32 %2, %2, %16, s[20:21]\n" "v_cndmask_b32 %3, %3, %16, s[20:21]\n" "v_cndmask_b32 %4, %4, %16, s[20:21]\n" "v_cndmask_b32 %5, %5, %16, s[20:21]\n" "v_cndmask_b32 %6, %6, %16, s[20:21]\n" "v_cndmask_b32 %7, %7, %16, s[20:21]\n" "v_cndmask_b32 %8, %8, %16, s[20:21]\n" "v_cndmask_b32 %9, %9, %16, s[20:21]\n" "v_cndmask_b32 %10, %10, %16, s[20:21]\n" "v_cndmask_b32 %11, %11, %16, s[20:21]\n" "v_cndmask_b32 %12, %12, %16, s[20:21]\n" "v_cndmask_b32 %13, %13, %16, s[20:21]\n" "v_cndmask_b32 %14, %14, %16, s[20:21]\n" "v_cndmask_b32 %15, %15, %16, s[20:21]\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CMP) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n" "v_cmp_lt_f32 vcc, %1, %16\n" "v_cmp_lt_f32 vcc, %2, %16\n" "v_cmp_lt_f32 vcc, %3, %16\n" "v_cmp_lt_f32 vcc, %4, %16\n" "v_cmp_lt_f32 vcc, %5, %16\n" "v_cmp_lt_f32 vcc, %6, %16\n" "v_cmp_lt_f32 vcc, %7, %16\n" "v_cmp_lt_f32 vcc, %8, %16\n" "v_cmp_lt_f32 vcc, %9, %16\n" "v_cmp_lt_f32 vcc, %10, %16\n" "v_cmp_lt_f32 vcc, %11, %16\n" "v_cmp_lt_f32 vcc, %12, %16\n" "v_cmp_lt_f32 vcc, %13, %16\n" "v_cmp_lt_f32 vcc, %14, %16\n" "v_cmp_lt_f32 vcc, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CMP_SGPR) {
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %16\n" "v_cmp_lt_f32 s[20:21], %1, %16\n" "v_cmp_lt_f32 s[20:21], %2, %16\n" "v_cmp_lt_f32 s[20:21], %3, %16\n" "v_cmp_lt_f32 s[20:21], %4, %16\n" "v_cmp_lt_f32 s[20:21], %5, %16\n" "v_cmp_lt_f32 s[20:21], %6, %16\n" "v_cmp_lt_f32 s[20:21], %7, %16\n" "v_cmp_lt_f32 s[20:21], %8, %16\n" "v_cmp_lt_f32 s[20:21], %9, %16\n" "v_cmp_lt_f32 s[20:21], %10, %16\n" "v_cmp_lt_f32 s[20:21], %11, %16\n" "v_cmp_lt_f32 s[20:21], %12, %16\n" "v_cmp_lt_f32 s[20:21], %13, %16\n" "v_cmp_lt_f32 s[20:21], %14, %16\n" "v_cmp_lt_f32 s[20:21], %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CMP_CNDMASK) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32 %0, %0, %17, vcc\n" "v_cmp_lt_f32 vcc, %1, %16\n v_cndmask_b32 %1, %1, %17, vcc\n" "v_cmp_lt_f32 vcc, %2, %16\n v_cndmask_b32 %2, %2, %17, vcc\n" "v_cmp_lt_f32 vcc, %3, %16\n v_cndmask_b32 %3, %3, %17, vcc\n" "v_cmp_lt_f32 vcc, %4, %16\n v_cndmask_b32 %4, %4, %17, vcc\n" "v_cmp_lt_f32 vcc, %5, %16\n v_cndmask_b32 %5, %5, %17, vcc\n" "v_cmp_lt_f32 vcc, %6, %16\n v_cndmask_b32 %6, %6, %17, vcc\n" "v_cmp_lt_f32 vcc, %7, %16\n v_cndmask_b32 %7, %7, %17, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMP_4CND) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32 %1, %1, %17, vcc\n v_cndmask_b32 %2, %2, %17, vcc\n v_cndmask_b32 %3, %3, %17, vcc\n v_cndmask_b32 %4, %4, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %5, %16\n v_cndmask_b32 %6, %6, %17, vcc\n v_cndmask_b32 %7, %7, %17, vcc\n v_cndmask_b32 %8, %8, %17, vcc\n v_cndmask_b32 %9, %9, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %10, %16\n v_cndmask_b32 %11, %11, %17, vcc\n v_cndmask_b32 %12, %12, %17, vcc\n v_cndmask_b32 %13, %13, %17, vcc\n v_cndmask_b32 %14, %14, %17, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMPS_4CND) {
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %16\n v_cndmask_b32 %1, %1, %17, s[20:21]\n v_cndmask_b32 %2, %2, %17, s[20:21]\n v_cndmask_b32 %3, %3, %17, s[20:21]\n v_cndmask_b32 %4, %4, %17, s[20:21]\n"
                             "v_cmp_lt_f32 s[22:23], %5, %16\n v_cndmask_b32 %6, %6, %17, s[22:23]\n v_cndmask_b32 %7, %7, %17, s[22:23]\n v_cndmask_b32 %8, %8, %17, s[22:23]\n v_cndmask_b32 %9, %9, %17, s[22:23]\n"
                             "v_cmp_lt_f32 s[20:21], %10, %16\n v_cndmask_b32 %11, %11, %17, s[20:21]\n v_cndmask_b32 %12, %12, %17, s[20:21]\n v_cndmask_b32 %13, %13, %17, s[20:21]\n v_cndmask_b32 %14, %14, %17, s[20:21]\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23");
            } else if constexpr (OP == OP_CMP_4CND_DEP) {
                // three compare-exchanges of (t, r) pairs as the compiler writes them: (a0,a1)x(a2,a3), (a4,a5)x(a6,a7), then (a0,a1)x(a4,a5)
                asm volatile("v_cmp_lt_f32 vcc, %2, %0\n v_cndmask_b32 %8, %0, %2, vcc\n v_cndmask_b32 %2, %2, %0, vcc\n v_cndmask_b32 %9, %1, %3, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n"
                             "v_cmp_lt_f32 vcc, %6, %4\n v_cndmask_b32 %10, %4, %6, vcc\n v_cndmask_b32 %6, %6, %4, vcc\n v_cndmask_b32 %11, %5, %7, vcc\n v_cndmask_b32 %7, %7, %5, vcc\n"
                             "v_cmp_lt_f32 vcc, %10, %8\n v_cndmask_b32 %0, %8, %10, vcc\n v_cndmask_b32 %4, %10, %8, vcc\n v_cndmask_b32 %1, %9, %11, vcc\n v_cndmask_b32 %5, %11, %9, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMP_3FMA_CND) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_cndmask_b32 %3, %3, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %16\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_cndmask_b32 %7, %7, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %8, %16\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_cndmask_b32 %11, %11, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %12, %16\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_cndmask_b32 %15, %15, %17, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CND_FMA_ALT) {
                asm volatile("v_cndmask_b32 %0, %0, %17, vcc\n v_fma_f32 %1, %1, %16, %17\n v_cndmask_b32 %2, %2, %17, vcc\n v_fma_f32 %3, %3, %16, %17\n"
                             "v_cndmask_b32 %4, %4, %17, vcc\n v_fma_f32 %5, %5, %16, %17\n v_cndmask_b32 %6, %6, %17, vcc\n v_fma_f32 %7, %7, %16, %17\n"
                             "v_cndmask_b32 %8, %8, %17, vcc\n v_fma_f32 %9, %9, %16, %17\n v_cndmask_b32 %10, %10, %17, vcc\n v_fma_f32 %11, %11, %16, %17\n"
                             "v_cndmask_b32 %12, %12, %17, vcc\n v_fma_f32 %13, %13, %16, %17\n v_cndmask_b32 %14, %14, %17, vcc\n v_fma_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMP_CND_FMA_CND) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32 %1, %1, %17, vcc\n v_fma_f32 %2, %2, %16, %17\n v_cndmask_b32 %3, %3, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %16\n v_cndmask_b32 %5, %5, %17, vcc\n v_fma_f32 %6, %6, %16, %17\n v_cndmask_b32 %7, %7, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %8, %16\n v_cndmask_b32 %9, %9, %17, vcc\n v_fma_f32 %10, %10, %16, %17\n v_cndmask_b32 %11, %11, %17, vcc\n"
                             "v_cmp_lt_f32 vcc, %12, %16\n v_cndmask_b32 %13, %13, %17, vcc\n v_fma_f32 %14, %14, %16, %17\n v_cndmask_b32 %15, %15, %17, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMP_CND_3FMA_CND) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32 %1, %1, %17, vcc\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n v_fma_f32 %4, %4, %16, %17\n v_cndmask_b32 %5, %5, %17, vcc\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                             "v_cmp_lt_f32 vcc, %8, %16\n v_cndmask_b32 %9, %9, %17, vcc\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n v_fma_f32 %12, %12, %16, %17\n v_cndmask_b32 %13, %13, %17, vcc\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CND_E64_VCC) {
                asm volatile("v_cndmask_b32_e64 %0, %0, %16, vcc\n" "v_cndmask_b32_e64 %1, %1, %16, vcc\n" "v_cndmask_b32_e64 %2, %2, %16, vcc\n" "v_cndmask_b32_e64 %3, %3, %16, vcc\n" "v_cndmask_b32_e64 %4, %4, %16, vcc\n" "v_cndmask_b32_e64 %5, %5, %16, vcc\n" "v_cndmask_b32_e64 %6, %6, %16, vcc\n" "v_cndmask_b32_e64 %7, %7, %16, vcc\n" "v_cndmask_b32_e64 %8, %8, %16, vcc\n" "v_cndmask_b32_e64 %9, %9, %16, vcc\n" "v_cndmask_b32_e64 %10, %10, %16, vcc\n" "v_cndmask_b32_e64 %11, %11, %16, vcc\n" "v_cndmask_b32_e64 %12, %12, %16, vcc\n" "v_cndmask_b32_e64 %13, %13, %16, vcc\n" "v_cndmask_b32_e64 %14, %14, %16, vcc\n" "v_cndmask_b32_e64 %15, %15, %16, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CND_NOP) {
                asm volatile("v_cndmask_b32_e32 %0, %0, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %1, %1, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %2, %2, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %3, %3, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %4, %4, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %5, %5, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %6, %6, %16, vcc\n s_nop 0\n" "v_cndmask_b32_e32 %7, %7, %16, vcc\n s_nop 0\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CND_E32_E64_ALT) {
                asm volatile("v_cndmask_b32_e32 %0, %0, %16, vcc\n v_cndmask_b32_e64 %8, %8, %16, s[20:21]\n" "v_cndmask_b32_e32 %1, %1, %16, vcc\n v_cndmask_b32_e64 %9, %9, %16, s[20:21]\n" "v_cndmask_b32_e32 %2, %2, %16, vcc\n v_cndmask_b32_e64 %10, %10, %16, s[20:21]\n" "v_cndmask_b32_e32 %3, %3, %16, vcc\n v_cndmask_b32_e64 %11, %11, %16, s[20:21]\n" "v_cndmask_b32_e32 %4, %4, %16, vcc\n v_cndmask_b32_e64 %12, %12, %16, s[20:21]\n" "v_cndmask_b32_e32 %5, %5, %16, vcc\n v_cndmask_b32_e64 %13, %13, %16, s[20:21]\n" "v_cndmask_b32_e32 %6, %6, %16, vcc\n v_cndmask_b32_e64 %14, %14, %16, s[20:21]\n" "v_cndmask_b32_e32 %7, %7, %16, vcc\n v_cndmask_b32_e64 %15, %15, %16, s[20:21]\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CMP_4CND_E64VCC) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32_e64 %1, %1, %17, vcc\n v_cndmask_b32_e64 %2, %2, %17, vcc\n v_cndmask_b32_e64 %3, %3, %17, vcc\n v_cndmask_b32_e64 %4, %4, %17, vcc\n" "v_cmp_lt_f32 vcc, %5, %16\n v_cndmask_b32_e64 %6, %6, %17, vcc\n v_cndmask_b32_e64 %7, %7, %17, vcc\n v_cndmask_b32_e64 %8, %8, %17, vcc\n v_cndmask_b32_e64 %9, %9, %17, vcc\n" "v_cmp_lt_f32 vcc, %10, %16\n v_cndmask_b32_e64 %11, %11, %17, vcc\n v_cndmask_b32_e64 %12, %12, %17, vcc\n v_cndmask_b32_e64 %13, %13, %17, vcc\n v_cndmask_b32_e64 %14, %14, %17, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMP_CND_NOP4) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32_e32 %1, %1, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %2, %2, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %3, %3, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %4, %4, %17, vcc\n" "v_cmp_lt_f32 vcc, %5, %16\n v_cndmask_b32_e32 %6, %6, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %7, %7, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %8, %8, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %9, %9, %17, vcc\n" "v_cmp_lt_f32 vcc, %10, %16\n v_cndmask_b32_e32 %11, %11, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %12, %12, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %13, %13, %17, vcc\n s_nop 0\n v_cndmask_b32_e32 %14, %14, %17, vcc\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_DIVFMAS) {
                asm volatile("v_div_fmas_f32 %0, %0, %16, %17\n" "v_div_fmas_f32 %1, %1, %16, %17\n" "v_div_fmas_f32 %2, %2, %16, %17\n" "v_div_fmas_f32 %3, %3, %16, %17\n" "v_div_fmas_f32 %4, %4, %16, %17\n" "v_div_fmas_f32 %5, %5, %16, %17\n" "v_div_fmas_f32 %6, %6, %16, %17\n" "v_div_fmas_f32 %7, %7, %16, %17\n" "v_div_fmas_f32 %8, %8, %16, %17\n" "v_div_fmas_f32 %9, %9, %16, %17\n" "v_div_fmas_f32 %10, %10, %16, %17\n" "v_div_fmas_f32 %11, %11, %16, %17\n" "v_div_fmas_f32 %12, %12, %16, %17\n" "v_div_fmas_f32 %13, %13, %16, %17\n" "v_div_fmas_f32 %14, %14, %16, %17\n" "v_div_fmas_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_ADDC) {
                asm volatile("v_addc_co_u32 %0, vcc, %0, %16, vcc\n" "v_addc_co_u32 %1, vcc, %1, %16, vcc\n" "v_addc_co_u32 %2, vcc, %2, %16, vcc\n" "v_addc_co_u32 %3, vcc, %3, %16, vcc\n" "v_addc_co_u32 %4, vcc, %4, %16, vcc\n" "v_addc_co_u32 %5, vcc, %5, %16, vcc\n" "v_addc_co_u32 %6, vcc, %6, %16, vcc\n" "v_addc_co_u32 %7, vcc, %7, %16, vcc\n" "v_addc_co_u32 %8, vcc, %8, %16, vcc\n" "v_addc_co_u32 %9, vcc, %9, %16, vcc\n" "v_addc_co_u32 %10, vcc, %10, %16, vcc\n" "v_addc_co_u32 %11, vcc, %11, %16, vcc\n" "v_addc_co_u32 %12, vcc, %12, %16, vcc\n" "v_addc_co_u32 %13, vcc, %13, %16, vcc\n" "v_addc_co_u32 %14, vcc, %14, %16, vcc\n" "v_addc_co_u32 %15, vcc, %15, %16, vcc\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc");
            } else if constexpr (OP == OP_CND2_FMA2) {
                asm volatile("v_cndmask_b32_e32 %0, %0, %16, vcc\n v_cndmask_b32_e32 %1, %1, %16, vcc\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n" "v_cndmask_b32_e32 %4, %4, %16, vcc\n v_cndmask_b32_e32 %5, %5, %16, vcc\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n" "v_cndmask_b32_e32 %8, %8, %16, vcc\n v_cndmask_b32_e32 %9, %9, %16, vcc\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n" "v_cndmask_b32_e32 %12, %12, %16, vcc\n v_cndmask_b32_e32 %13, %13, %16, vcc\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CND3_FMA) {
                asm volatile("v_cndmask_b32_e32 %0, %0, %16, vcc\n v_cndmask_b32_e32 %1, %1, %16, vcc\n v_cndmask_b32_e32 %2, %2, %16, vcc\n v_fma_f32 %3, %3, %16, %17\n" "v_cndmask_b32_e32 %4, %4, %16, vcc\n v_cndmask_b32_e32 %5, %5, %16, vcc\n v_cndmask_b32_e32 %6, %6, %16, vcc\n v_fma_f32 %7, %7, %16, %17\n" "v_cndmask_b32_e32 %8, %8, %16, vcc\n v_cndmask_b32_e32 %9, %9, %16, vcc\n v_cndmask_b32_e32 %10, %10, %16, vcc\n v_fma_f32 %11, %11, %16, %17\n" "v_cndmask_b32_e32 %12, %12, %16, vcc\n v_cndmask_b32_e32 %13, %13, %16, vcc\n v_cndmask_b32_e32 %14, %14, %16, vcc\n v_fma_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CND2_MAX2) {
                asm volatile("v_cndmask_b32_e32 %0, %0, %16, vcc\n v_cndmask_b32_e32 %1, %1, %16, vcc\n v_max_f32 %2, %2, %16\n v_max_f32 %3, %3, %16\n" "v_cndmask_b32_e32 %4, %4, %16, vcc\n v_cndmask_b32_e32 %5, %5, %16, vcc\n v_max_f32 %6, %6, %16\n v_max_f32 %7, %7, %16\n" "v_cndmask_b32_e32 %8, %8, %16, vcc\n v_cndmask_b32_e32 %9, %9, %16, vcc\n v_max_f32 %10, %10, %16\n v_max_f32 %11, %11, %16\n" "v_cndmask_b32_e32 %12, %12, %16, vcc\n v_cndmask_b32_e32 %13, %13, %16, vcc\n v_max_f32 %14, %14, %16\n v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_CMP_CND2_FMA) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %16\n v_cndmask_b32_e32 %1, %1, %17, vcc\n v_cndmask_b32_e32 %2, %2, %17, vcc\n v_fma_f32 %3, %3, %16, %17\n" "v_cmp_lt_f32 vcc, %4, %16\n v_cndmask_b32_e32 %5, %5, %17, vcc\n v_cndmask_b32_e32 %6, %6, %17, vcc\n v_fma_f32 %7, %7, %16, %17\n" "v_cmp_lt_f32 vcc, %8, %16\n v_cndmask_b32_e32 %9, %9, %17, vcc\n v_cndmask_b32_e32 %10, %10, %17, vcc\n v_fma_f32 %11, %11, %16, %17\n" "v_cmp_lt_f32 vcc, %12, %16\n v_cndmask_b32_e32 %13, %13, %17, vcc\n v_cndmask_b32_e32 %14, %14, %17, vcc\n v_fma_f32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_AND) {
                asm volatile("v_and_b32 %0, %0, %16\n" "v_and_b32 %1, %1, %16\n" "v_and_b32 %2, %2, %16\n" "v_and_b32 %3, %3, %16\n" "v_and_b32 %4, %4, %16\n" "v_and_b32 %5, %5, %16\n" "v_and_b32 %6, %6, %16\n" "v_and_b32 %7, %7, %16\n" "v_and_b32 %8, %8, %16\n" "v_and_b32 %9, %9, %16\n" "v_and_b32 %10, %10, %16\n" "v_and_b32 %11, %11, %16\n" "v_and_b32 %12, %12, %16\n" "v_and_b32 %13, %13, %16\n" "v_and_b32 %14, %14, %16\n" "v_and_b32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_OR) {
                asm volatile("v_or_b32 %0, %0, %16\n" "v_or_b32 %1, %1, %16\n" "v_or_b32 %2, %2, %16\n" "v_or_b32 %3, %3, %16\n" "v_or_b32 %4, %4, %16\n" "v_or_b32 %5, %5, %16\n" "v_or_b32 %6, %6, %16\n" "v_or_b32 %7, %7, %16\n" "v_or_b32 %8, %8, %16\n" "v_or_b32 %9, %9, %16\n" "v_or_b32 %10, %10, %16\n" "v_or_b32 %11, %11, %16\n" "v_or_b32 %12, %12, %16\n" "v_or_b32 %13, %13, %16\n" "v_or_b32 %14, %14, %16\n" "v_or_b32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_XOR) {
                asm volatile("v_xor_b32 %0, %0, %16\n" "v_xor_b32 %1, %1, %16\n" "v_xor_b32 %2, %2, %16\n" "v_xor_b32 %3, %3, %16\n" "v_xor_b32 %4, %4, %16\n" "v_xor_b32 %5, %5, %16\n" "v_xor_b32 %6, %6, %16\n" "v_xor_b32 %7, %7, %16\n" "v_xor_b32 %8, %8, %16\n" "v_xor_b32 %9, %9, %16\n" "v_xor_b32 %10, %10, %16\n" "v_xor_b32 %11, %11, %16\n" "v_xor_b32 %12, %12, %16\n" "v_xor_b32 %13, %13, %16\n" "v_xor_b32 %14, %14, %16\n" "v_xor_b32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_LSHL) {
                asm volatile("v_lshlrev_b32 %0, 1, %0\n" "v_lshlrev_b32 %1, 1, %1\n" "v_lshlrev_b32 %2, 1, %2\n" "v_lshlrev_b32 %3, 1, %3\n" "v_lshlrev_b32 %4, 1, %4\n" "v_lshlrev_b32 %5, 1, %5\n" "v_lshlrev_b32 %6, 1, %6\n" "v_lshlrev_b32 %7, 1, %7\n" "v_lshlrev_b32 %8, 1, %8\n" "v_lshlrev_b32 %9, 1, %9\n" "v_lshlrev_b32 %10, 1, %10\n" "v_lshlrev_b32 %11, 1, %11\n" "v_lshlrev_b32 %12, 1, %12\n" "v_lshlrev_b32 %13, 1, %13\n" "v_lshlrev_b32 %14, 1, %14\n" "v_lshlrev_b32 %15, 1, %15\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_ADDU) {
                asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_SUBU) {
                asm volatile("v_sub_u32 %0, %0, %16\n" "v_sub_u32 %1, %1, %16\n" "v_sub_u32 %2, %2, %16\n" "v_sub_u32 %3, %3, %16\n" "v_sub_u32 %4, %4, %16\n" "v_sub_u32 %5, %5, %16\n" "v_sub_u32 %6, %6, %16\n" "v_sub_u32 %7, %7, %16\n" "v_sub_u32 %8, %8, %16\n" "v_sub_u32 %9, %9, %16\n" "v_sub_u32 %10, %10, %16\n" "v_sub_u32 %11, %11, %16\n" "v_sub_u32 %12, %12, %16\n" "v_sub_u32 %13, %13, %16\n" "v_sub_u32 %14, %14, %16\n" "v_sub_u32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MINU) {
                asm volatile("v_min_u32 %0, %0, %16\n" "v_min_u32 %1, %1, %16\n" "v_min_u32 %2, %2, %16\n" "v_min_u32 %3, %3, %16\n" "v_min_u32 %4, %4, %16\n" "v_min_u32 %5, %5, %16\n" "v_min_u32 %6, %6, %16\n" "v_min_u32 %7, %7, %16\n" "v_min_u32 %8, %8, %16\n" "v_min_u32 %9, %9, %16\n" "v_min_u32 %10, %10, %16\n" "v_min_u32 %11, %11, %16\n" "v_min_u32 %12, %12, %16\n" "v_min_u32 %13, %13, %16\n" "v_min_u32 %14, %14, %16\n" "v_min_u32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MAXI) {
                asm volatile("v_max_i32 %0, %0, %16\n" "v_max_i32 %1, %1, %16\n" "v_max_i32 %2, %2, %16\n" "v_max_i32 %3, %3, %16\n" "v_max_i32 %4, %4, %16\n" "v_max_i32 %5, %5, %16\n" "v_max_i32 %6, %6, %16\n" "v_max_i32 %7, %7, %16\n" "v_max_i32 %8, %8, %16\n" "v_max_i32 %9, %9, %16\n" "v_max_i32 %10, %10, %16\n" "v_max_i32 %11, %11, %16\n" "v_max_i32 %12, %12, %16\n" "v_max_i32 %13, %13, %16\n" "v_max_i32 %14, %14, %16\n" "v_max_i32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_LSHL_ADD) {
                asm volatile("v_lshl_add_u32 %0, %0, 1, %16\n" "v_lshl_add_u32 %1, %1, 1, %16\n" "v_lshl_add_u32 %2, %2, 1, %16\n" "v_lshl_add_u32 %3, %3, 1, %16\n" "v_lshl_add_u32 %4, %4, 1, %16\n" "v_lshl_add_u32 %5, %5, 1, %16\n" "v_lshl_add_u32 %6, %6, 1, %16\n" "v_lshl_add_u32 %7, %7, 1, %16\n" "v_lshl_add_u32 %8, %8, 1, %16\n" "v_lshl_add_u32 %9, %9, 1, %16\n" "v_lshl_add_u32 %10, %10, 1, %16\n" "v_lshl_add_u32 %11, %11, 1, %16\n" "v_lshl_add_u32 %12, %12, 1, %16\n" "v_lshl_add_u32 %13, %13, 1, %16\n" "v_lshl_add_u32 %14, %14, 1, %16\n" "v_lshl_add_u32 %15, %15, 1, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_AND_OR) {
                asm volatile("v_and_or_b32 %0, %0, %16, %17\n" "v_and_or_b32 %1, %1, %16, %17\n" "v_and_or_b32 %2, %2, %16, %17\n" "v_and_or_b32 %3, %3, %16, %17\n" "v_and_or_b32 %4, %4, %16, %17\n" "v_and_or_b32 %5, %5, %16, %17\n" "v_and_or_b32 %6, %6, %16, %17\n" "v_and_or_b32 %7, %7, %16, %17\n" "v_and_or_b32 %8, %8, %16, %17\n" "v_and_or_b32 %9, %9, %16, %17\n" "v_and_or_b32 %10, %10, %16, %17\n" "v_and_or_b32 %11, %11, %16, %17\n" "v_and_or_b32 %12, %12, %16, %17\n" "v_and_or_b32 %13, %13, %16, %17\n" "v_and_or_b32 %14, %14, %16, %17\n" "v_and_or_b32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_BFI) {
                asm volatile("v_bfi_b32 %0, %16, %0, %17\n" "v_bfi_b32 %1, %16, %1, %17\n" "v_bfi_b32 %2, %16, %2, %17\n" "v_bfi_b32 %3, %16, %3, %17\n" "v_bfi_b32 %4, %16, %4, %17\n" "v_bfi_b32 %5, %16, %5, %17\n" "v_bfi_b32 %6, %16, %6, %17\n" "v_bfi_b32 %7, %16, %7, %17\n" "v_bfi_b32 %8, %16, %8, %17\n" "v_bfi_b32 %9, %16, %9, %17\n" "v_bfi_b32 %10, %16, %10, %17\n" "v_bfi_b32 %11, %16, %11, %17\n" "v_bfi_b32 %12, %16, %12, %17\n" "v_bfi_b32 %13, %16, %13, %17\n" "v_bfi_b32 %14, %16, %14, %17\n" "v_bfi_b32 %15, %16, %15, %17\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_BITOP3) {
                asm volatile("v_bitop3_b32 %0, %0, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %1, %1, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %2, %2, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %3, %3, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %4, %4, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %5, %5, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %6, %6, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %7, %7, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %8, %8, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %9, %9, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %10, %10, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %11, %11, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %12, %12, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %13, %13, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %14, %14, %16, %17 bitop3:0x6c\n" "v_bitop3_b32 %15, %15, %16, %17 bitop3:0x6c\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_ASHR) {
                asm volatile("v_ashrrev_i32 %0, 1, %0\n" "v_ashrrev_i32 %1, 1, %1\n" "v_ashrrev_i32 %2, 1, %2\n" "v_ashrrev_i32 %3, 1, %3\n" "v_ashrrev_i32 %4, 1, %4\n" "v_ashrrev_i32 %5, 1, %5\n" "v_ashrrev_i32 %6, 1, %6\n" "v_ashrrev_i32 %7, 1, %7\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %9, 1, %9\n" "v_ashrrev_i32 %10, 1, %10\n" "v_ashrrev_i32 %11, 1, %11\n" "v_ashrrev_i32 %12, 1, %12\n" "v_ashrrev_i32 %13, 1, %13\n" "v_ashrrev_i32 %14, 1, %14\n" "v_ashrrev_i32 %15, 1, %15\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIX_BITOP3_MAX) {
                asm volatile("v_bitop3_b32 %0, %0, %16, %17 bitop3:0x6c\n" "v_max_f32 %1, %1, %16\n" "v_bitop3_b32 %2, %2, %16, %17 bitop3:0x6c\n" "v_max_f32 %3, %3, %16\n" "v_bitop3_b32 %4, %4, %16, %17 bitop3:0x6c\n" "v_max_f32 %5, %5, %16\n" "v_bitop3_b32 %6, %6, %16, %17 bitop3:0x6c\n" "v_max_f32 %7, %7, %16\n" "v_bitop3_b32 %8, %8, %16, %17 bitop3:0x6c\n" "v_max_f32 %9, %9, %16\n" "v_bitop3_b32 %10, %10, %16, %17 bitop3:0x6c\n" "v_max_f32 %11, %11, %16\n" "v_bitop3_b32 %12, %12, %16, %17 bitop3:0x6c\n" "v_max_f32 %13, %13, %16\n" "v_bitop3_b32 %14, %14, %16, %17 bitop3:0x6c\n" "v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIX_AND_MAX) {
                asm volatile("v_and_b32 %0, %0, %16\n" "v_max_f32 %1, %1, %16\n" "v_and_b32 %2, %2, %16\n" "v_max_f32 %3, %3, %16\n" "v_and_b32 %4, %4, %16\n" "v_max_f32 %5, %5, %16\n" "v_and_b32 %6, %6, %16\n" "v_max_f32 %7, %7, %16\n" "v_and_b32 %8, %8, %16\n" "v_max_f32 %9, %9, %16\n" "v_and_b32 %10, %10, %16\n" "v_max_f32 %11, %11, %16\n" "v_and_b32 %12, %12, %16\n" "v_max_f32 %13, %13, %16\n" "v_and_b32 %14, %14, %16\n" "v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIX_ADDU_MAX) {
                asm volatile("v_add_u32 %0, %0, %16\n" "v_max_f32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_max_f32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_max_f32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_max_f32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_max_f32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_max_f32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_max_f32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIX_MULF_MAX) {
                asm volatile("v_mul_f32 %0, %0, %16\n" "v_max_f32 %1, %1, %16\n" "v_mul_f32 %2, %2, %16\n" "v_max_f32 %3, %3, %16\n" "v_mul_f32 %4, %4, %16\n" "v_max_f32 %5, %5, %16\n" "v_mul_f32 %6, %6, %16\n" "v_max_f32 %7, %7, %16\n" "v_mul_f32 %8, %8, %16\n" "v_max_f32 %9, %9, %16\n" "v_mul_f32 %10, %10, %16\n" "v_max_f32 %11, %11, %16\n" "v_mul_f32 %12, %12, %16\n" "v_max_f32 %13, %13, %16\n" "v_mul_f32 %14, %14, %16\n" "v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIX_XOR_CND) {
                asm volatile("v_xor_b32 %0, %0, %16\n" "v_cndmask_b32_e64 %1, %1, %16, s[20:21]\n" "v_xor_b32 %2, %2, %16\n" "v_cndmask_b32_e64 %3, %3, %16, s[20:21]\n" "v_xor_b32 %4, %4, %16\n" "v_cndmask_b32_e64 %5, %5, %16, s[20:21]\n" "v_xor_b32 %6, %6, %16\n" "v_cndmask_b32_e64 %7, %7, %16, s[20:21]\n" "v_xor_b32 %8, %8, %16\n" "v_cndmask_b32_e64 %9, %9, %16, s[20:21]\n" "v_xor_b32 %10, %10, %16\n" "v_cndmask_b32_e64 %11, %11, %16, s[20:21]\n" "v_xor_b32 %12, %12, %16\n" "v_cndmask_b32_e64 %13, %13, %16, s[20:21]\n" "v_xor_b32 %14, %14, %16\n" "v_cndmask_b32_e64 %15, %15, %16, s[20:21]\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_BFE) {
                asm volatile("v_bfe_u32 %0, %0, 3, 7\n" "v_bfe_u32 %1, %1, 3, 7\n" "v_bfe_u32 %2, %2, 3, 7\n" "v_bfe_u32 %3, %3, 3, 7\n" "v_bfe_u32 %4, %4, 3, 7\n" "v_bfe_u32 %5, %5, 3, 7\n" "v_bfe_u32 %6, %6, 3, 7\n" "v_bfe_u32 %7, %7, 3, 7\n" "v_bfe_u32 %8, %8, 3, 7\n" "v_bfe_u32 %9, %9, 3, 7\n" "v_bfe_u32 %10, %10, 3, 7\n" "v_bfe_u32 %11, %11, 3, 7\n" "v_bfe_u32 %12, %12, 3, 7\n" "v_bfe_u32 %13, %13, 3, 7\n" "v_bfe_u32 %14, %14, 3, 7\n" "v_bfe_u32 %15, %15, 3, 7\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_PERM) {
                asm volatile("v_perm_b32 %0, %0, %16, %17\n" "v_perm_b32 %1, %1, %16, %17\n" "v_perm_b32 %2, %2, %16, %17\n" "v_perm_b32 %3, %3, %16, %17\n" "v_perm_b32 %4, %4, %16, %17\n" "v_perm_b32 %5, %5, %16, %17\n" "v_perm_b32 %6, %6, %16, %17\n" "v_perm_b32 %7, %7, %16, %17\n" "v_perm_b32 %8, %8, %16, %17\n" "v_perm_b32 %9, %9, %16, %17\n" "v_perm_b32 %10, %10, %16, %17\n" "v_perm_b32 %11, %11, %16, %17\n" "v_perm_b32 %12, %12, %16, %17\n" "v_perm_b32 %13, %13, %16, %17\n" "v_perm_b32 %14, %14, %16, %17\n" "v_perm_b32 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MUL24) {
                asm volatile("v_mul_u32_u24 %0, %0, %16\n" "v_mul_u32_u24 %1, %1, %16\n" "v_mul_u32_u24 %2, %2, %16\n" "v_mul_u32_u24 %3, %3, %16\n" "v_mul_u32_u24 %4, %4, %16\n" "v_mul_u32_u24 %5, %5, %16\n" "v_mul_u32_u24 %6, %6, %16\n" "v_mul_u32_u24 %7, %7, %16\n" "v_mul_u32_u24 %8, %8, %16\n" "v_mul_u32_u24 %9, %9, %16\n" "v_mul_u32_u24 %10, %10, %16\n" "v_mul_u32_u24 %11, %11, %16\n" "v_mul_u32_u24 %12, %12, %16\n" "v_mul_u32_u24 %13, %13, %16\n" "v_mul_u32_u24 %14, %14, %16\n" "v_mul_u32_u24 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MAD24) {
                asm volatile("v_mad_u32_u24 %0, %0, %16, %17\n" "v_mad_u32_u24 %1, %1, %16, %17\n" "v_mad_u32_u24 %2, %2, %16, %17\n" "v_mad_u32_u24 %3, %3, %16, %17\n" "v_mad_u32_u24 %4, %4, %16, %17\n" "v_mad_u32_u24 %5, %5, %16, %17\n" "v_mad_u32_u24 %6, %6, %16, %17\n" "v_mad_u32_u24 %7, %7, %16, %17\n" "v_mad_u32_u24 %8, %8, %16, %17\n" "v_mad_u32_u24 %9, %9, %16, %17\n" "v_mad_u32_u24 %10, %10, %16, %17\n" "v_mad_u32_u24 %11, %11, %16, %17\n" "v_mad_u32_u24 %12, %12, %16, %17\n" "v_mad_u32_u24 %13, %13, %16, %17\n" "v_mad_u32_u24 %14, %14, %16, %17\n" "v_mad_u32_u24 %15, %15, %16, %17\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MULLO) {
                asm volatile("v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %1, %1, %16\n" "v_mul_lo_u32 %2, %2, %16\n" "v_mul_lo_u32 %3, %3, %16\n" "v_mul_lo_u32 %4, %4, %16\n" "v_mul_lo_u32 %5, %5, %16\n" "v_mul_lo_u32 %6, %6, %16\n" "v_mul_lo_u32 %7, %7, %16\n" "v_mul_lo_u32 %8, %8, %16\n" "v_mul_lo_u32 %9, %9, %16\n" "v_mul_lo_u32 %10, %10, %16\n" "v_mul_lo_u32 %11, %11, %16\n" "v_mul_lo_u32 %12, %12, %16\n" "v_mul_lo_u32 %13, %13, %16\n" "v_mul_lo_u32 %14, %14, %16\n" "v_mul_lo_u32 %15, %15, %16\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_CVT) {
                asm volatile("v_cvt_f32_u32 %0, %0\n" "v_cvt_f32_u32 %1, %1\n" "v_cvt_f32_u32 %2, %2\n" "v_cvt_f32_u32 %3, %3\n" "v_cvt_f32_u32 %4, %4\n" "v_cvt_f32_u32 %5, %5\n" "v_cvt_f32_u32 %6, %6\n" "v_cvt_f32_u32 %7, %7\n" "v_cvt_f32_u32 %8, %8\n" "v_cvt_f32_u32 %9, %9\n" "v_cvt_f32_u32 %10, %10\n" "v_cvt_f32_u32 %11, %11\n" "v_cvt_f32_u32 %12, %12\n" "v_cvt_f32_u32 %13, %13\n" "v_cvt_f32_u32 %14, %14\n" "v_cvt_f32_u32 %15, %15\n" : ACC_OPERANDS : "v"(ib), "v"(ic) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MOV) {
                asm volatile("v_mov_b32 %0, %16\n" "v_mov_b32 %1, %16\n" "v_mov_b32 %2, %16\n" "v_mov_b32 %3, %16\n" "v_mov_b32 %4, %16\n" "v_mov_b32 %5, %16\n" "v_mov_b32 %6, %16\n" "v_mov_b32 %7, %16\n" "v_mov_b32 %8, %16\n" "v_mov_b32 %9, %16\n" "v_mov_b32 %10, %16\n" "v_mov_b32 %11, %16\n" "v_mov_b32 %12, %16\n" "v_mov_b32 %13, %16\n" "v_mov_b32 %14, %16\n" "v_mov_b32 %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MOV_DPP) {
                asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_READLANE) {
                asm volatile("v_readfirstlane_b32 s20, %0\n" "v_readfirstlane_b32 s20, %1\n" "v_readfirstlane_b32 s20, %2\n" "v_readfirstlane_b32 s20, %3\n" "v_readfirstlane_b32 s20, %4\n" "v_readfirstlane_b32 s20, %5\n" "v_readfirstlane_b32 s20, %6\n" "v_readfirstlane_b32 s20, %7\n" "v_readfirstlane_b32 s20, %8\n" "v_readfirstlane_b32 s20, %9\n" "v_readfirstlane_b32 s20, %10\n" "v_readfirstlane_b32 s20, %11\n" "v_readfirstlane_b32 s20, %12\n" "v_readfirstlane_b32 s20, %13\n" "v_readfirstlane_b32 s20, %14\n" "v_readfirstlane_b32 s20, %15\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_PKFMA) {
                asm volatile("v_pk_fma_f32 %0, %0, %16, %17\n" "v_pk_fma_f32 %1, %1, %16, %17\n" "v_pk_fma_f32 %2, %2, %16, %17\n" "v_pk_fma_f32 %3, %3, %16, %17\n" "v_pk_fma_f32 %4, %4, %16, %17\n" "v_pk_fma_f32 %5, %5, %16, %17\n" "v_pk_fma_f32 %6, %6, %16, %17\n" "v_pk_fma_f32 %7, %7, %16, %17\n" "v_pk_fma_f32 %8, %8, %16, %17\n" "v_pk_fma_f32 %9, %9, %16, %17\n" "v_pk_fma_f32 %10, %10, %16, %17\n" "v_pk_fma_f32 %11, %11, %16, %17\n" "v_pk_fma_f32 %12, %12, %16, %17\n" "v_pk_fma_f32 %13, %13, %16, %17\n" "v_pk_fma_f32 %14, %14, %16, %17\n" "v_pk_fma_f32 %15, %15, %16, %17\n" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]) : "v"(pb), "v"(pc));
            } else if constexpr (OP == OP_PKMUL) {
                asm volatile("v_pk_mul_f32 %0, %0, %16\n" "v_pk_mul_f32 %1, %1, %16\n" "v_pk_mul_f32 %2, %2, %16\n" "v_pk_mul_f32 %3, %3, %16\n" "v_pk_mul_f32 %4, %4, %16\n" "v_pk_mul_f32 %5, %5, %16\n" "v_pk_mul_f32 %6, %6, %16\n" "v_pk_mul_f32 %7, %7, %16\n" "v_pk_mul_f32 %8, %8, %16\n" "v_pk_mul_f32 %9, %9, %16\n" "v_pk_mul_f32 %10, %10, %16\n" "v_pk_mul_f32 %11, %11, %16\n" "v_pk_mul_f32 %12, %12, %16\n" "v_pk_mul_f32 %13, %13, %16\n" "v_pk_mul_f32 %14, %14, %16\n" "v_pk_mul_f32 %15, %15, %16\n" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]) : "v"(pb), "v"(pc));
            } else if constexpr (OP == OP_RCP) {
                asm volatile("v_rcp_f32 %0, %0\n" "v_rcp_f32 %1, %1\n" "v_rcp_f32 %2, %2\n" "v_rcp_f32 %3, %3\n" "v_rcp_f32 %4, %4\n" "v_rcp_f32 %5, %5\n" "v_rcp_f32 %6, %6\n" "v_rcp_f32 %7, %7\n" "v_rcp_f32 %8, %8\n" "v_rcp_f32 %9, %9\n" "v_rcp_f32 %10, %10\n" "v_rcp_f32 %11, %11\n" "v_rcp_f32 %12, %12\n" "v_rcp_f32 %13, %13\n" "v_rcp_f32 %14, %14\n" "v_rcp_f32 %15, %15\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc", "s20", "s21");
            } else if constexpr (OP == OP_MIX_FMA_MAX) {
                asm volatile("v_fma_f32 %0, %0, %16, %17\n v_max_f32 %8, %8, %16\n" "v_fma_f32 %1, %1, %16, %17\n v_max_f32 %9, %9, %16\n" "v_fma_f32 %2, %2, %16, %17\n v_max_f32 %10, %10, %16\n" "v_fma_f32 %3, %3, %16, %17\n v_max_f32 %11, %11, %16\n" "v_fma_f32 %4, %4, %16, %17\n v_max_f32 %12, %12, %16\n" "v_fma_f32 %5, %5, %16, %17\n v_max_f32 %13, %13, %16\n" "v_fma_f32 %6, %6, %16, %17\n v_max_f32 %14, %14, %16\n" "v_fma_f32 %7, %7, %16, %17\n v_max_f32 %15, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_MIX_FMA_CMP) {
                asm volatile("v_fma_f32 %0, %0, %16, %17\n v_cmp_lt_f32 vcc, %8, %16\n" "v_fma_f32 %1, %1, %16, %17\n v_cmp_lt_f32 vcc, %9, %16\n" "v_fma_f32 %2, %2, %16, %17\n v_cmp_lt_f32 vcc, %10, %16\n" "v_fma_f32 %3, %3, %16, %17\n v_cmp_lt_f32 vcc, %11, %16\n" "v_fma_f32 %4, %4, %16, %17\n v_cmp_lt_f32 vcc, %12, %16\n" "v_fma_f32 %5, %5, %16, %17\n v_cmp_lt_f32 vcc, %13, %16\n" "v_fma_f32 %6, %6, %16, %17\n v_cmp_lt_f32 vcc, %14, %16\n" "v_fma_f32 %7, %7, %16, %17\n v_cmp_lt_f32 vcc, %15, %16\n" : ACC_OPERANDS : "v"(b), "v"(c) : "vcc");
            } else if constexpr (OP == OP_DSREAD) {
                asm volatile("ds_read_b32 %0, %16 offset:0\n" "ds_read_b32 %1, %16 offset:1024\n" "ds_read_b32 %2, %16 offset:2048\n" "ds_read_b32 %3, %16 offset:3072\n" "ds_read_b32 %4, %16 offset:4096\n" "ds_read_b32 %5, %16 offset:5120\n" "ds_read_b32 %6, %16 offset:6144\n" "ds_read_b32 %7, %16 offset:7168\n" "ds_read_b32 %8, %16 offset:8192\n" "ds_read_b32 %9, %16 offset:9216\n" "ds_read_b32 %10, %16 offset:10240\n" "ds_read_b32 %11, %16 offset:11264\n" "ds_read_b32 %12, %16 offset:12288\n" "ds_read_b32 %13, %16 offset:13312\n" "ds_read_b32 %14, %16 offset:14336\n" "ds_read_b32 %15, %16 offset:15360\n" "s_waitcnt lgkmcnt(0)\n" : ACC_OPERANDS : "v"(ldsAddr), "v"(c) : "memory");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63u) == 0) {
        WaveRecord w; w.ticks = t1 - t0; w.realticks = r1 - r0;
        w.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);          // HW_REG_HW_ID, 32 bits
        w.xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);          // HW_REG_XCC_ID
        rec[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = w;
    }
}

typedef void (*Kernel)(WaveRecord*, float*, int, float, float);
static Kernel kKernels[OP_COUNT] = { issue_loop<OP_FMA>, issue_loop<OP_FMAC>, issue_loop<OP_ADD>, issue_loop<OP_SUB>, issue_loop<OP_MUL>, issue_loop<OP_MAX>, issue_loop<OP_MIN3>, issue_loop<OP_MED3>, issue_loop<OP_CNDMASK>, issue_loop<OP_CNDMASK_SWAP>, issue_loop<OP_CNDMASK_SGPR>, issue_loop<OP_CMP>, issue_loop<OP_CMP_SGPR>, issue_loop<OP_CMP_CNDMASK>, issue_loop<OP_CMP_4CND>, issue_loop<OP_CMPS_4CND>, issue_loop<OP_CMP_4CND_DEP>, issue_loop<OP_CMP_3FMA_CND>, issue_loop<OP_CND_FMA_ALT>, issue_loop<OP_CMP_CND_FMA_CND>, issue_loop<OP_CMP_CND_3FMA_CND>, issue_loop<OP_CND_E64_VCC>, issue_loop<OP_CND_NOP>, issue_loop<OP_CND_E32_E64_ALT>, issue_loop<OP_CMP_4CND_E64VCC>, issue_loop<OP_CMP_CND_NOP4>, issue_loop<OP_DIVFMAS>, issue_loop<OP_ADDC>, issue_loop<OP_CND2_FMA2>, issue_loop<OP_CND3_FMA>, issue_loop<OP_CND2_MAX2>, issue_loop<OP_CMP_CND2_FMA>, issue_loop<OP_AND>, issue_loop<OP_OR>, issue_loop<OP_XOR>, issue_loop<OP_LSHL>, issue_loop<OP_ADDU>, issue_loop<OP_SUBU>, issue_loop<OP_MINU>, issue_loop<OP_MAXI>, issue_loop<OP_LSHL_ADD>, issue_loop<OP_AND_OR>, issue_loop<OP_BFI>, issue_loop<OP_BFE>, issue_loop<OP_PERM>, issue_loop<OP_MUL24>, issue_loop<OP_MAD24>, issue_loop<OP_MULLO>, issue_loop<OP_CVT>, issue_loop<OP_MOV>, issue_loop<OP_MOV_DPP>, issue_loop<OP_READLANE>, issue_loop<OP_PKFMA>, issue_loop<OP_PKMUL>, issue_loop<OP_RCP>, issue_loop<OP_MIX_FMA_MAX>, issue_loop<OP_MIX_FMA_CMP>, issue_loop<OP_DSREAD>, issue_loop<OP_BITOP3>, issue_loop<OP_ASHR>, issue_loop<OP_MIX_BITOP3_MAX>, issue_loop<OP_MIX_AND_MAX>, issue_loop<OP_MIX_ADDU_MAX>, issue_loop<OP_MIX_MULF_MAX>, issue_loop<OP_MIX_XOR_CND> };

int main(int argc, char** argv)
{
    const double targetMs = argc > 1 ? atof(argv[1]) : 40.0;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s (%s), %d CUs, nominal clock %d kHz; target %.0f ms per case; 128 instructions per loop trip\n", prop.name, prop.gcnArchName, cus, prop.clockRate, targetMs);
    const int maxBlocks = cus * 8;
    WaveRecord* dRec; float* dSink;
    CHECK(hipMalloc(&dRec, sizeof(WaveRecord) * (size_t)maxBlocks * 4)); CHECK(hipMalloc(&dSink, sizeof(float) * (size_t)maxBlocks * 256));
    std::vector<WaveRecord> rec((size_t)maxBlocks * 4);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // warm the clocks: ~0.5 s of the plain FMA loop on the whole chip
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(kKernels[OP_FMA], dim3(cus * 4), dim3(256), 0, 0, dRec, dSink, 60000, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    printf("%-36s %5s %9s %9s | %9s %9s %9s | %10s %10s | %s\n", "instruction", "k", "ms", "clock MHz", "tick/inst", "(max wave)", "SIMDcyc/i", "wall cyc/i", "SIMDcyc/i", "waves per SIMD census (min..max, SIMDs used)");
    for (int op = 0; op < OP_COUNT; ++op) {
        if (argc > 2) { bool want = false; for (int q = 2; q < argc; ++q) want = want || atoi(argv[q]) == op; if (!want) continue; }
        for (int k : { 1, 2, 4, 6 }) {
            const int blocks = cus * k;
            int iters = 2000;
            float ms = 0.0f;
            for (int pass = 0; pass < 3; ++pass) {            // pass 0: calibration; pass 1: warm-up at full length; pass 2: the measurement
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kKernels[op], dim3(blocks), dim3(256), 0, 0, dRec, dSink, iters, 1.0001f, 0.5f);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (pass == 0) iters = std::max(2000, (int)(iters * targetMs / std::max(0.01f, ms)));
            }
            CHECK(hipMemcpy(rec.data(), dRec, sizeof(WaveRecord) * (size_t)blocks * 4, hipMemcpyDeviceToHost));
            const size_t nw = (size_t)blocks * 4;
            std::vector<double> ticks(nw), clk(nw);
            std::map<std::tuple<unsigned, unsigned>, int> census;      // (xcc, hw_id fields that name a SIMD) -> waves
            for (size_t i = 0; i < nw; ++i) {
                ticks[i] = (double)rec[i].ticks; clk[i] = (double)rec[i].ticks / (double)rec[i].realticks * 100.0;      // MHz
                // HW_ID (gfx9 family): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...: everything but the wave slot
                census[std::make_tuple(rec[i].xcc & 0xFu, rec[i].hwid & 0xFFF0u & ~0x00C0u)]++;
            }
            std::sort(ticks.begin(), ticks.end()); std::sort(clk.begin(), clk.end());
            const double inst = (double)iters * ((op == OP_CMP_4CND || op == OP_CMPS_4CND || op == OP_CMP_4CND_DEP || op == OP_CMP_4CND_E64VCC || op == OP_CMP_CND_NOP4) ? 120.0 : (op == OP_CND_NOP ? 64.0 : 128.0));
            const double medTicks = ticks[nw / 2], maxTicks = ticks[nw - 1], medClk = clk[nw / 2];
            int cmin = 1 << 30, cmax = 0; for (auto& kv : census) { cmin = std::min(cmin, kv.second); cmax = std::max(cmax, kv.second); }
            // per wave: ticks per instruction; per SIMD: divide by the waves that shared it (k when the census says so)
            const double wallCycles = ms * 1e-3 * medClk * 1e6;
            printf("%-36s %5d %9.2f %9.0f | %9.3f %9.3f %9.3f | %10.3f %10.3f | %d..%d on %zu SIMDs%s\n", kOpName[op], k, ms, medClk, medTicks / inst, maxTicks / inst, medTicks / inst / k,
                   wallCycles / inst, wallCycles / inst / k, cmin, cmax, census.size(), (cmin == k && cmax == k && census.size() == (size_t)cus * 4) ? "" : "  <-- UNEVEN placement");
            fflush(stdout);
        }
    }
    return 0;
}
