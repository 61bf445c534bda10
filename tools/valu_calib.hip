// valu_calib.hip -- what one gfx950 SIMD really issues, measured: long dependence-free streams of the instructions this library
// lives on, at 1 / 2 / 4 / 8 waves per SIMD, timed per wave with s_memtime and over the launch with hipEvents.
// Why: DESIGN.md priced the pipeline's VALU wave-instructions against "one VALU per 4 cycles per SIMD" (rounds 2 and 3 used 3
// and 4), /opt/skills/guides/MI355X_MICROARCH.md says a wave64 VALU occupies a SIMD-32 for 2 cycles.  The two differ by a
// factor of two in "how full is the chip", so the number is measured here (the HBM side was calibrated the same way,
// tools/pmc_calib.hip).
//   every kernel: 256-thread workgroups (one wave per SIMD of a CU), grid = CUs x waves-per-SIMD; every wave runs ITERS
//   iterations of 64 independent instructions of one kind (8 register chains x 8) and stores its s_memtime ticks and its
//   HW_ID / XCC_ID, so that the host can check where the waves really sat.
// Output: one JSON object on stdout (tools/gpu_valu_calib.sh puts it into gpurun_out/valu_calib.json).
//   per kind and waves/SIMD: cyc_per_instr_wave (median ticks a wave needs per instruction), instr_per_cyc_simd
//   (= resident waves on a SIMD x instructions / ticks, median over SIMDs), chip_ginstr_s (all wave-instructions / wall time).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

struct WaveRec {
    unsigned long long ticks;
    uint32_t hw_id, xcc_id;
};

__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ void stamp(WaveRec *rec, unsigned long long t0, unsigned long long t1)
{
    if ((threadIdx.x & 63) == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        WaveRec r;
        r.ticks = t1 - t0;
        r.hw_id = hw;
        r.xcc_id = xcc;
        rec[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
}

// ---- the instruction kinds.  BODY(k) = one instruction on chain k; eight of them make a round, eight rounds an iteration.
#define KERNEL_V(NAME, DECL, BODY, SINK)                                                            \
    __global__ __launch_bounds__(256) void NAME(WaveRec *rec, int iters, uint32_t *sink, uint32_t seed) \
    {                                                                                               \
        DECL;                                                                                       \
        const unsigned long long t0 = memtime();                                                    \
        for (int i = 0; i < iters; i++) {                                                           \
            R8(BODY) R8(BODY) R8(BODY) R8(BODY) R8(BODY) R8(BODY) R8(BODY) R8(BODY)                 \
        }                                                                                           \
        const unsigned long long t1 = memtime();                                                    \
        stamp(rec, t0, t1);                                                                         \
        SINK;                                                                                       \
    }

#define DECL_U32                                                                                    \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, \
             a7 = a0 * 19u, b = (blockIdx.x + seed) | 1u, c = seed * 2654435761u + 77u
#define SINK_U32                                                                                    \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) *sink = a0

#define B_ADD(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_add_u32, DECL_U32, B_ADD, SINK_U32)
#define B_DOT2C(k) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_v_dot2c_i32_i16, DECL_U32, B_DOT2C, SINK_U32)
#define B_MUL24(k) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_mul_i32_i24, DECL_U32, B_MUL24, SINK_U32)
#define B_MULLIT(k) asm volatile("v_mul_i32_i24 %0, 0x6b1, %0" : "+v"(a##k));  // (a 32-bit literal: 8-byte encoding)
KERNEL_V(k_v_mul_i32_i24_literal, DECL_U32, B_MULLIT, SINK_U32)
#define B_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_mul_lo_u32, DECL_U32, B_MULLO, SINK_U32)
#define B_PKSUB(k) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_pk_sub_u16, DECL_U32, B_PKSUB, SINK_U32)
#define B_ALIGN(k) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_alignbit_b32, DECL_U32, B_ALIGN, SINK_U32)
#define B_DPP(k) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_mov_b32_dpp, DECL_U32, B_DPP, SINK_U32)
#define B_ADDDPP(k) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_add_u32_dpp, DECL_U32, B_ADDDPP, SINK_U32)
#define B_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : );
#define DECL_VCC                                                                                    \
    DECL_U32;                                                                                       \
    asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc")
KERNEL_V(k_v_cndmask_b32, DECL_VCC, B_CNDMASK, SINK_U32)
// selects: the compare + select idiom of every "keep the smaller" in the walkers, and what could stand in for it
#define B_CNDMASK64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "s"(msk));
#define DECL_MSK                                                                                    \
    DECL_U32;                                                                                       \
    unsigned long long msk = __builtin_amdgcn_read_exec() ^ (0x5555ull * (seed | 1u))
KERNEL_V(k_v_cndmask_b32_sgprmask, DECL_MSK, B_CNDMASK64, SINK_U32)
#define B_CMPSEL(k)                                                                                 \
    asm volatile("v_cmp_lt_u32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : "vcc");
KERNEL_V(k_pair_cmp_cndmask_vcc, DECL_U32, B_CMPSEL, SINK_U32)  // 2 VALU per BODY
#define DECL_E                                                                                      \
    DECL_U32;                                                                                       \
    uint32_t e0 = a0 + 1;                                                                           \
    uint32_t e1 = a1 + 1;                                                                           \
    uint32_t e2 = a2 + 1;                                                                           \
    uint32_t e3 = a3 + 1;                                                                           \
    uint32_t e4 = a4 + 1;                                                                           \
    uint32_t e5 = a5 + 1;                                                                           \
    uint32_t e6 = a6 + 1;                                                                           \
    uint32_t e7 = a7 + 1;                                                                           \
    unsigned long long msk = 0
#define SINK_E                                                                                      \
    if ((e0 ^ e1 ^ e2 ^ e3 ^ e4 ^ e5 ^ e6 ^ e7) == 0x12345u) *sink = e0;                            \
    if (msk == 0x12345ull) *sink = 2;                                                               \
    SINK_U32
#define B_CMPSEL2(k)                                                                                \
    asm volatile("v_cmp_lt_u32_e32 vcc, %2, %0\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(a##k), "+v"(e##k) : "v"(b) : "vcc");
KERNEL_V(k_cmp_2cndmask_vcc, DECL_E, B_CMPSEL2, SINK_E)  // 3 VALU per BODY
#define B_CMPSEL4(k)                                                                                \
    asm volatile("v_cmp_lt_u32_e32 vcc, %2, %0\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %1, %1, %0, vcc" : "+v"(a##k), "+v"(e##k) : "v"(b) : "vcc");
KERNEL_V(k_cmp_4cndmask_vcc, DECL_E, B_CMPSEL4, SINK_E)  // 5 VALU per BODY
#define B_CMPSEL64(k)                                                                               \
    asm volatile("v_cmp_lt_u32_e64 %2, %3, %0\n v_cndmask_b32_e64 %0, %0, %3, %2\n v_cndmask_b32_e64 %1, %1, %3, %2" : "+v"(a##k), "+v"(e##k), "+s"(msk) : "v"(b));
KERNEL_V(k_cmp_2cndmask_sgpr, DECL_E, B_CMPSEL64, SINK_E)  // 3 VALU per BODY
#define B_MINU(k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_min_u32, DECL_U32, B_MINU, SINK_U32)
#define B_BFI(k) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_v_bfi_b32, DECL_U32, B_BFI, SINK_U32)
#define B_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_v_and_or_b32, DECL_U32, B_ANDOR, SINK_U32)
#define B_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_xor_b32, DECL_U32, B_XOR, SINK_U32)
#define B_LSHL(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##k));
KERNEL_V(k_v_lshlrev_b32, DECL_U32, B_LSHL, SINK_U32)
#define B_SUBU(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_sub_u32, DECL_U32, B_SUBU, SINK_U32)
#define B_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_v_add3_u32, DECL_U32, B_ADD3, SINK_U32)
#define B_ADDCO(k) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a##k) : "v"(b) : "vcc");
KERNEL_V(k_v_add_co_u32, DECL_U32, B_ADDCO, SINK_U32)
#define B_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a##k) : "v"(b));
KERNEL_V(k_v_mov_b32, DECL_U32, B_MOV, SINK_U32)
#define B_PKADD(k) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_pk_add_u16, DECL_U32, B_PKADD, SINK_U32)
#define B_ADDF32(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_add_f32_on_ints, DECL_U32, B_ADDF32, SINK_U32)
#define B_BFE(k) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_bfe_u32, DECL_U32, B_BFE, SINK_U32)
#define B_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_lshl_add_u32, DECL_U32, B_LSHLADD, SINK_U32)

// v_cmp -> SGPR pair (the threshold kernel's mask words), and v_writelane (its ballot parking)
#define DECL_CMP                                                                                     \
    DECL_U32;                                                                                       \
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0
#define SINK_CMP                                                                                    \
    if ((m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7) == 0x12345ull) *sink = (uint32_t)m0;                 \
    SINK_U32
#define B_CMP(k) asm volatile("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(m##k) : "v"(a##k), "v"(b));
KERNEL_V(k_v_cmp_to_sgpr, DECL_CMP, B_CMP, SINK_CMP)
#define B_CMPVCC(k) asm volatile("v_cmp_ge_i32_e32 vcc, %0, %1" : : "v"(a##k), "v"(b) : "vcc");
KERNEL_V(k_v_cmp_to_vcc, DECL_U32, B_CMPVCC, SINK_U32)
#define DECL_WL                                                                                     \
    DECL_U32;                                                                                       \
    uint32_t s0 = __builtin_amdgcn_readfirstlane(seed), s1 = __builtin_amdgcn_readfirstlane(seed * 3u)
#define B_WL(k) asm volatile("v_writelane_b32 %0, %1, " #k : "+v"(a##k) : "s"(s0));
KERNEL_V(k_v_writelane_b32, DECL_WL, B_WL, SINK_U32)
#define B_RL(k) asm volatile("v_readlane_b32 %0, %1, " #k : "=s"(s0) : "v"(a##k));
KERNEL_V(k_v_readlane_b32, DECL_WL, B_RL, if (s0 == 0x12345u) *sink = s0; SINK_U32)
// the threshold kernel's inner pattern: v_mul literal, v_cmp -> SGPR, two v_writelane from that pair
#define B_THR(k)                                                                                    \
    asm volatile("v_mul_i32_i24 %0, 0x6b1, %1" : "=v"(a##k) : "v"(b));                              \
    asm volatile("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(m##k) : "v"(c), "v"(a##k));                   \
    asm volatile("v_writelane_b32 %0, %1, " #k : "+v"(a0) : "s"((uint32_t)m##k));                    \
    asm volatile("v_writelane_b32 %0, %1, " #k : "+v"(a1) : "s"((uint32_t)(m##k >> 32)));
KERNEL_V(k_mix_mul_cmp_writelane2, DECL_CMP, B_THR, SINK_CMP)  // 4 VALU per BODY: 256 per iteration

#define B_v_and_b32(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_and_b32, DECL_U32, B_v_and_b32, SINK_U32)
#define B_v_or_b32(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_or_b32, DECL_U32, B_v_or_b32, SINK_U32)
#define B_v_lshrrev_b32(k) asm volatile("v_lshrrev_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_lshrrev_b32, DECL_U32, B_v_lshrrev_b32, SINK_U32)
#define B_v_ashrrev_i32(k) asm volatile("v_ashrrev_i32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_ashrrev_i32, DECL_U32, B_v_ashrrev_i32, SINK_U32)
#define B_v_max_u32(k) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_max_u32, DECL_U32, B_v_max_u32, SINK_U32)
#define B_v_max_i32(k) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_max_i32, DECL_U32, B_v_max_i32, SINK_U32)
#define B_v_min_i32(k) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_min_i32, DECL_U32, B_v_min_i32, SINK_U32)
#define B_v_mul_u32_u24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_mul_u32_u24, DECL_U32, B_v_mul_u32_u24, SINK_U32)
#define B_v_mul_hi_u32(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_mul_hi_u32, DECL_U32, B_v_mul_hi_u32, SINK_U32)
#define B_v_subrev_u32(k) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_subrev_u32, DECL_U32, B_v_subrev_u32, SINK_U32)
#define B_v_mul_f32(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_mul_f32, DECL_U32, B_v_mul_f32, SINK_U32)
#define B_v_sub_f32(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_sub_f32, DECL_U32, B_v_sub_f32, SINK_U32)
#define B_v_max_f32(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_max_f32, DECL_U32, B_v_max_f32, SINK_U32)
#define B_v_min_f32(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_min_f32, DECL_U32, B_v_min_f32, SINK_U32)
#define B_v_mad_u32_u24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_mad_u32_u24, DECL_U32, B_v_mad_u32_u24, SINK_U32)
#define B_v_mad_i32_i24(k) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_mad_i32_i24, DECL_U32, B_v_mad_i32_i24, SINK_U32)
#define B_v_lshl_or_b32(k) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_lshl_or_b32, DECL_U32, B_v_lshl_or_b32, SINK_U32)
#define B_v_or3_b32(k) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_or3_b32, DECL_U32, B_v_or3_b32, SINK_U32)
#define B_v_xad_u32(k) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_xad_u32, DECL_U32, B_v_xad_u32, SINK_U32)
#define B_v_perm_b32(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_perm_b32, DECL_U32, B_v_perm_b32, SINK_U32)
#define B_v_med3_i32(k) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_med3_i32, DECL_U32, B_v_med3_i32, SINK_U32)
#define B_v_min3_u32(k) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_min3_u32, DECL_U32, B_v_min3_u32, SINK_U32)
#define B_v_sad_u32(k) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_x_v_sad_u32, DECL_U32, B_v_sad_u32, SINK_U32)
#define B_v_not_b32(k) asm volatile("v_not_b32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_not_b32, DECL_U32, B_v_not_b32, SINK_U32)
#define B_v_bfrev_b32(k) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_bfrev_b32, DECL_U32, B_v_bfrev_b32, SINK_U32)
#define B_v_ffbh_u32(k) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_ffbh_u32, DECL_U32, B_v_ffbh_u32, SINK_U32)
#define B_v_ffbl_b32(k) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_ffbl_b32, DECL_U32, B_v_ffbl_b32, SINK_U32)
#define B_v_cvt_f32_u32(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_cvt_f32_u32, DECL_U32, B_v_cvt_f32_u32, SINK_U32)
#define B_v_cvt_f32_i32(k) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_cvt_f32_i32, DECL_U32, B_v_cvt_f32_i32, SINK_U32)
#define B_v_cvt_u32_f32(k) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_cvt_u32_f32, DECL_U32, B_v_cvt_u32_f32, SINK_U32)
#define B_v_cvt_f32_ubyte0(k) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a##k));
KERNEL_V(k_x_v_cvt_f32_ubyte0, DECL_U32, B_v_cvt_f32_ubyte0, SINK_U32)
#define B_bcnt(k) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_bcnt_u32_b32, DECL_U32, B_bcnt, SINK_U32)
#define B_mbcnt(k) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a##k) : "v"(b));
KERNEL_V(k_x_v_mbcnt_lo, DECL_U32, B_mbcnt, SINK_U32)
#define DECL_U64                                                                                    \
    unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u,      \
                       a6 = a0 * 17u, a7 = a0 * 19u;                                                \
    uint32_t b = (blockIdx.x + seed) | 1u, c = seed * 2654435761u + 77u
#define SINK_U64                                                                                    \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345ull) *sink = (uint32_t)a0
#define B_lshl64(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a##k));
KERNEL_V(k_x_v_lshlrev_b64, DECL_U64, B_lshl64, SINK_U64)
#define B_mad64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##k) : "v"(b), "v"(c) : "vcc");
KERNEL_V(k_x_v_mad_u64_u32, DECL_U64, B_mad64, SINK_U64)
// the matrix pipe, for the record of "box sums as banded 0 / 1 products" (DESIGN.md): i8 MFMA, 16 accumulator registers a chain
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_x_mfma_i32_32x32x32_i8(WaveRec *rec, int iters, uint32_t *sink, uint32_t seed)
{
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    v4i a = {(int)(threadIdx.x + seed), 1, 2, 3}, b = {(int)seed, 5, 6, 7};
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = memtime();
    stamp(rec, t0, t1);
    if (c0[0] + c1[1] + c2[2] + c3[3] == 0x12345) *sink = 1;
}

// floats and doubles
#define DECL_F32                                                                                    \
    float a0 = threadIdx.x + seed, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 * 11.f, a5 = a0 * 13.f, a6 = a0 * 17.f, \
          a7 = a0 * 19.f, b = 1.0000001f, c = 1e-9f
#define SINK_F32                                                                                    \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.12345f) *sink = 1
#define B_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_v_fma_f32, DECL_F32, B_FMA32, SINK_F32)
#define DECL_F64                                                                                    \
    double a0 = threadIdx.x + seed, a1 = a0 * 3., a2 = a0 * 5., a3 = a0 * 7., a4 = a0 * 11., a5 = a0 * 13., a6 = a0 * 17.,     \
           a7 = a0 * 19., b = 1.0000001, c = 1e-9
#define SINK_F64                                                                                    \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.12345) *sink = 1
#define B_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
KERNEL_V(k_v_fma_f64, DECL_F64, B_FMA64, SINK_F64)
#define B_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_add_f64, DECL_F64, B_ADD64, SINK_F64)
#define B_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##k) : "v"(b));
KERNEL_V(k_v_mul_f64, DECL_F64, B_MUL64, SINK_F64)
#define B_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##k));
KERNEL_V(k_v_rcp_f64, DECL_F64, B_RCP64, SINK_F64)
#define B_SQRT64(k) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a##k));
KERNEL_V(k_v_sqrt_f64, DECL_F64, B_SQRT64, SINK_F64)
// an IEEE double division as the compiler expands it (v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup): "instruction"
// = one division
#define B_DIV64(k) a##k = b / a##k;
KERNEL_V(k_f64_division, DECL_F64, B_DIV64, SINK_F64)

// scalar side: SALU alone, and SALU beside VALU in the same wave
#define DECL_S                                                                                      \
    DECL_U32;                                                                                       \
    uint32_t s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 * 3u, s2 = s0 * 5u, s3 = s0 * 7u, s4 = s0 * 11u, s5 = s0 * 13u, \
             s6 = s0 * 17u, s7 = s0 * 19u
#define SINK_S                                                                                      \
    if ((s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) == 0x12345u) *sink = s0;                            \
    SINK_U32
#define B_SADD(k) asm volatile("s_add_u32 %0, %0, 0x6b1" : "+s"(s##k) : : "scc");
KERNEL_V(k_s_add_u32, DECL_S, B_SADD, SINK_S)
#define B_VS(k)                                                                                     \
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));                                     \
    asm volatile("s_add_u32 %0, %0, 0x6b1" : "+s"(s##k) : : "scc");
KERNEL_V(k_mix_valu_salu_1to1, DECL_S, B_VS, SINK_S)  // 1 VALU + 1 SALU per BODY

// LDS reads beside nothing: ds_read_b64 issue rate (the threshold kernel's other half)
__global__ __launch_bounds__(256) void k_ds_read_b64(WaveRec *rec, int iters, uint32_t *sink, uint32_t seed)
{
    __shared__ unsigned long long buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = i * seed;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(buf + threadIdx.x);
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const unsigned long long t0 = memtime();
    for (int i = 0; i < iters; i++) {
#define B_DS(k) asm volatile("ds_read_b64 %0, %1 offset:" #k "*2048" : "=v"(a##k) : "v"(addr));
        R8(B_DS) R8(B_DS) R8(B_DS) R8(B_DS) R8(B_DS) R8(B_DS) R8(B_DS) R8(B_DS)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = memtime();
    stamp(rec, t0, t1);
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345ull) *sink = 1;
}

typedef void (*kern_t)(WaveRec *, int, uint32_t *, uint32_t);
struct Kind {
    const char *name;
    kern_t fn;
    int instr_per_iter;  // "instructions" of the named kind per loop iteration and wave
    const char *counts;
};

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2048;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int CUS = prop.multiProcessorCount;
    const Kind kinds[] = {
        {"v_add_u32", k_v_add_u32, 64, "VALU"},
        {"v_dot2c_i32_i16", k_v_dot2c_i32_i16, 64, "VALU"},
        {"v_mul_i32_i24", k_v_mul_i32_i24, 64, "VALU"},
        {"v_mul_i32_i24 (literal)", k_v_mul_i32_i24_literal, 64, "VALU"},
        {"v_mul_lo_u32", k_v_mul_lo_u32, 64, "VALU"},
        {"v_pk_sub_u16", k_v_pk_sub_u16, 64, "VALU"},
        {"v_alignbit_b32", k_v_alignbit_b32, 64, "VALU"},
        {"v_mov_b32 dpp row_shr", k_v_mov_b32_dpp, 64, "VALU"},
        {"v_add_u32 dpp row_shr", k_v_add_u32_dpp, 64, "VALU"},
        {"v_cndmask_b32", k_v_cndmask_b32, 64, "VALU"},
        {"v_cndmask_b32 e64, mask in an sgpr pair", k_v_cndmask_b32_sgprmask, 64, "VALU"},
        {"pair: v_cmp_lt_u32 vcc + v_cndmask vcc", k_pair_cmp_cndmask_vcc, 128, "VALU"},
        {"triple: v_cmp vcc + 2 v_cndmask vcc", k_cmp_2cndmask_vcc, 192, "VALU"},
        {"five: v_cmp vcc + 4 v_cndmask vcc", k_cmp_4cndmask_vcc, 320, "VALU"},
        {"triple: v_cmp e64 sgpr + 2 v_cndmask e64 sgpr", k_cmp_2cndmask_sgpr, 192, "VALU"},
        {"v_min_u32", k_v_min_u32, 64, "VALU"},
        {"v_bfi_b32", k_v_bfi_b32, 64, "VALU"},
        {"v_and_or_b32", k_v_and_or_b32, 64, "VALU"},
        {"v_xor_b32", k_v_xor_b32, 64, "VALU"},
        {"v_lshlrev_b32", k_v_lshlrev_b32, 64, "VALU"},
        {"v_sub_u32", k_v_sub_u32, 64, "VALU"},
        {"v_add3_u32", k_v_add3_u32, 64, "VALU"},
        {"v_add_co_u32 (vcc out)", k_v_add_co_u32, 64, "VALU"},
        {"v_mov_b32", k_v_mov_b32, 64, "VALU"},
        {"v_pk_add_u16", k_v_pk_add_u16, 64, "VALU"},
        {"v_add_f32", k_v_add_f32_on_ints, 64, "VALU"},
        {"v_bfe_u32", k_v_bfe_u32, 64, "VALU"},
        {"v_lshl_add_u32", k_v_lshl_add_u32, 64, "VALU"},
        {"v_cmp -> sgpr pair", k_v_cmp_to_sgpr, 64, "VALU"},
        {"v_cmp -> vcc", k_v_cmp_to_vcc, 64, "VALU"},
        {"v_writelane_b32", k_v_writelane_b32, 64, "VALU"},
        {"v_readlane_b32", k_v_readlane_b32, 64, "VALU"},
        {"mix: v_mul literal + v_cmp->sgpr + 2 v_writelane", k_mix_mul_cmp_writelane2, 256, "VALU"},
        {"v_and_b32", k_x_v_and_b32, 64, "VALU"},
        {"v_or_b32", k_x_v_or_b32, 64, "VALU"},
        {"v_lshrrev_b32", k_x_v_lshrrev_b32, 64, "VALU"},
        {"v_ashrrev_i32", k_x_v_ashrrev_i32, 64, "VALU"},
        {"v_max_u32", k_x_v_max_u32, 64, "VALU"},
        {"v_max_i32", k_x_v_max_i32, 64, "VALU"},
        {"v_min_i32", k_x_v_min_i32, 64, "VALU"},
        {"v_mul_u32_u24", k_x_v_mul_u32_u24, 64, "VALU"},
        {"v_mul_hi_u32", k_x_v_mul_hi_u32, 64, "VALU"},
        {"v_subrev_u32", k_x_v_subrev_u32, 64, "VALU"},
        {"v_mul_f32", k_x_v_mul_f32, 64, "VALU"},
        {"v_sub_f32", k_x_v_sub_f32, 64, "VALU"},
        {"v_max_f32", k_x_v_max_f32, 64, "VALU"},
        {"v_min_f32", k_x_v_min_f32, 64, "VALU"},
        {"v_mad_u32_u24", k_x_v_mad_u32_u24, 64, "VALU"},
        {"v_mad_i32_i24", k_x_v_mad_i32_i24, 64, "VALU"},
        {"v_lshl_or_b32", k_x_v_lshl_or_b32, 64, "VALU"},
        {"v_or3_b32", k_x_v_or3_b32, 64, "VALU"},
        {"v_xad_u32", k_x_v_xad_u32, 64, "VALU"},
        {"v_perm_b32", k_x_v_perm_b32, 64, "VALU"},
        {"v_med3_i32", k_x_v_med3_i32, 64, "VALU"},
        {"v_min3_u32", k_x_v_min3_u32, 64, "VALU"},
        {"v_sad_u32", k_x_v_sad_u32, 64, "VALU"},
        {"v_not_b32", k_x_v_not_b32, 64, "VALU"},
        {"v_bfrev_b32", k_x_v_bfrev_b32, 64, "VALU"},
        {"v_ffbh_u32", k_x_v_ffbh_u32, 64, "VALU"},
        {"v_ffbl_b32", k_x_v_ffbl_b32, 64, "VALU"},
        {"v_cvt_f32_u32", k_x_v_cvt_f32_u32, 64, "VALU"},
        {"v_cvt_f32_i32", k_x_v_cvt_f32_i32, 64, "VALU"},
        {"v_cvt_u32_f32", k_x_v_cvt_u32_f32, 64, "VALU"},
        {"v_cvt_f32_ubyte0", k_x_v_cvt_f32_ubyte0, 64, "VALU"},
        {"v_bcnt_u32_b32", k_x_v_bcnt_u32_b32, 64, "VALU"},
        {"v_mbcnt_lo_u32_b32", k_x_v_mbcnt_lo, 64, "VALU"},
        {"v_lshlrev_b64", k_x_v_lshlrev_b64, 64, "VALU"},
        {"v_mad_u64_u32", k_x_v_mad_u64_u32, 64, "VALU"},
        {"v_mfma_i32_32x32x32_i8", k_x_mfma_i32_32x32x32_i8, 64, "MFMA"},
        {"v_fma_f32", k_v_fma_f32, 64, "VALU"},
        {"v_fma_f64", k_v_fma_f64, 64, "VALU"},
        {"v_add_f64", k_v_add_f64, 64, "VALU"},
        {"v_mul_f64", k_v_mul_f64, 64, "VALU"},
        {"v_rcp_f64", k_v_rcp_f64, 64, "VALU"},
        {"v_sqrt_f64", k_v_sqrt_f64, 64, "VALU"},
        {"f64 division (compiler expansion)", k_f64_division, 64, "divisions"},
        {"s_add_u32", k_s_add_u32, 64, "SALU"},
        {"mix: 1 v_add_u32 + 1 s_add_u32", k_mix_valu_salu_1to1, 128, "VALU+SALU"},
        {"ds_read_b64", k_ds_read_b64, 64, "LDS"},
    };
    const int wps_list[] = {1, 2, 4, 8};
    WaveRec *d_rec = nullptr;
    uint32_t *d_sink = nullptr;
    const size_t max_waves = (size_t)CUS * 8 * 4;
    if (hipMalloc((void **)&d_rec, max_waves * sizeof(WaveRec)) != hipSuccess || hipMalloc((void **)&d_sink, 64) != hipSuccess) return 1;
    std::vector<WaveRec> rec(max_waves);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("{\n \"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"iters\": %d,\n", prop.name, prop.gcnArchName, CUS,
           prop.clockRate / 1000, iters);
    printf(" \"note\": \"workgroups of 4 waves (one per SIMD), grid = CUs x waves-per-SIMD; ticks = s_memtime; instr_per_cyc_simd = (waves resident on "
           "a SIMD x instructions per wave) / that SIMD's median ticks, median over the SIMDs that held exactly the intended number of waves; "
           "chip_ginstr_s = all wave-instructions of the launch / hipEvent wall time; ticks_per_us = median ticks / wall time (the s_memtime clock)\",\n");
    printf(" \"kinds\": {\n");
    bool first_kind = true;
    for (const Kind &K : kinds) {
        if (argc > 2 && !strstr(K.name, argv[2])) continue;
        printf("%s  \"%s\": {\"counts\": \"%s\", \"by_waves_per_simd\": {", first_kind ? "" : ",\n", K.name, K.counts);
        first_kind = false;
        bool first_w = true;
        for (int wps : wps_list) {
            const int blocks = CUS * wps;
            const size_t nw = (size_t)blocks * 4;
            // warm-up, then the timed launch
            hipLaunchKernelGGL(K.fn, dim3(blocks), dim3(256), 0, 0, d_rec, 64, d_sink, 1u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(K.fn, dim3(blocks), dim3(256), 0, 0, d_rec, iters, d_sink, 3u);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) return 2;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(rec.data(), d_rec, nw * sizeof(WaveRec), hipMemcpyDeviceToHost);
            // where the waves sat: key = (xcc, se, sh, cu, simd) from HW_ID (gfx9: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]) + XCC_ID[3:0]
            std::map<uint32_t, std::vector<unsigned long long>> by_simd;
            std::vector<unsigned long long> all;
            for (size_t i = 0; i < nw; i++) {
                const uint32_t hw = rec[i].hw_id;
                const uint32_t key = ((rec[i].xcc_id & 0xfu) << 16) | (((hw >> 13) & 7u) << 13) | (((hw >> 12) & 1u) << 12) | (((hw >> 8) & 0xfu) << 8) | ((hw >> 4) & 3u);
                by_simd[key].push_back(rec[i].ticks);
                all.push_back(rec[i].ticks);
            }
            std::sort(all.begin(), all.end());
            const double med_ticks = (double)all[all.size() / 2];
            const double n_instr = (double)K.instr_per_iter * iters;
            std::vector<double> ipc;
            size_t simds_exact = 0;
            for (auto &kv : by_simd) {
                if ((int)kv.second.size() != wps) continue;
                simds_exact++;
                std::sort(kv.second.begin(), kv.second.end());
                ipc.push_back(wps * n_instr / (double)kv.second[kv.second.size() / 2]);
            }
            std::sort(ipc.begin(), ipc.end());
            const double ipc_med = ipc.empty() ? 0. : ipc[ipc.size() / 2];
            printf("%s\"%d\": {\"cyc_per_instr_wave_min_p90_max\": [%.3f, %.3f, %.3f], \"cyc_per_instr_wave\": %.3f, \"instr_per_cyc_simd\": %.4f, \"chip_ginstr_s\": %.1f, \"wall_ms\": %.4f, \"simds_used\": %zu, "
                   "\"simds_with_exactly_n_waves\": %zu, \"ticks_per_us\": %.1f}",
                   first_w ? "" : ", ", wps, (double)all.front() / n_instr, (double)all[all.size() * 9 / 10] / n_instr, (double)all.back() / n_instr, med_ticks / n_instr, ipc_med, n_instr * nw / (ms * 1e-3) / 1e9, ms, by_simd.size(), simds_exact,
                   med_ticks / (ms * 1e3));
            first_w = false;
        }
        printf("}}");
    }
    printf("\n }\n}\n");
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
