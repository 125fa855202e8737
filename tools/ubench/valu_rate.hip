// VALU issue interval per op class on gfx950 (wave64), separated from clock and dependency effects.
//
//   clock      : every workgroup reads s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop;
//                shader clock = sum(d memtime) / sum(d realtime) * 100 MHz, measured while the op under test runs.
//   dependency : CH = 8 independent chains per wave, and 1 / 2 / 4 / 8 waves per SIMD (256-thread workgroups, one
//                wave per SIMD each, 256 * wps workgroups).  The figure to read is the one that stops improving.
//   op class   : one instruction per asm statement, written out (the compiler cannot fuse, reorder or strength-reduce).
//
// Reported per (op, waves per SIMD):
//   cyc/SIMD  = kernel time (hipEvent) * measured clock * 1024 SIMDs / wave-instructions issued   <- the issue interval
//   cyc/wave  = s_memtime of a wave's loop / its instructions (= cyc/SIMD * waves when the SIMD is the limit)
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_bench valu_rate.hip ; run: ./valu_rate_bench [> table]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <vector>

#define ITERS 1024
#define CH 8
#define UNROLL 8            // body = CH * UNROLL = 64 instructions of the class per trip
typedef float f2 __attribute__((ext_vector_type(2)));

enum Op {
    ADD_F32, MUL_F32, FMA_F32, FMAC_F32, MAX_F32, MED3_F32,
    PK_ADD_F32, PK_MUL_F32, PK_FMA_F32,
    ADD_F64, MUL_F64, FMA_F64,
    CVT_F64_F32, CVT_F32_F64, CVT_F32_I32, CVT_I32_F32,
    EXP_F32, RCP_F32, SQRT_F32, RCP_F64,
    ADD_U32, SUB_U32, AND_B32, LSHL_B32, BFE_U32, ADD3_U32, LSHL_ADD_U32, AND_OR_B32, PERM_B32, ALIGNBIT,
    MIN_U32, MED3_U32, MAD_U32_U24, MUL_LO_U32, MUL_HI_U32, ADD_CO_U32, LSHL_B64,
    SAD_U8, SAD_HI_U8, MSAD_U8,
    MOV_B32, CNDMASK, CMP_F32_VCC, CMP_F32_SGPR, CMP_U32_VCC, CMPX_F32,
    MOV_DPP_ROW, MOV_DPP_WAVE, ADD_F32_DPP, MBCNT, READLANE, READFIRSTLANE, PK_ADD_U16,
    CNDMASK_SGPR, OR_B32, XOR_B32, LSHR_B32, ASHR_I32, MUL_U32_U24, MUL_F32_SGPR, ADD_F32_LIT, SUB_F32, MIN_F32, FMA_F32_3V, MAD_U64_U32,
    BCNT, FFBL, FFBH, CVT_U32_F32, FLOOR_F32, FRACT_F32, TRUNC_F32, RNDNE_F32, OR3_B32, XAD_U32, ADD_LSHL_U32, SQRT_F64, LDEXP_F32, MUL_LEGACY, CMP_CLASS, MAX_U32, MIN_I32, SUBREV_U32, ADDC_U32, MOV_SGPR, NOT_B32, CVT_F16_F32, BFI_B32, MAD_I32_I24, FMA_MIX, S_ADD,
    N_OPS
};
static const char *names[N_OPS] = {
    "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_max_f32", "v_med3_f32",
    "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
    "v_add_f64", "v_mul_f64", "v_fma_f64",
    "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f32_i32", "v_cvt_i32_f32",
    "v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_rcp_f64",
    "v_add_u32", "v_sub_u32", "v_and_b32", "v_lshlrev_b32", "v_bfe_u32", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_perm_b32", "v_alignbit_b32",
    "v_min_u32", "v_med3_u32", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co_u32", "v_lshlrev_b64",
    "v_sad_u8", "v_sad_hi_u8", "v_msad_u8",
    "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr", "v_cmp_lt_u32 vcc", "v_cmpx_lt_f32",
    "v_mov_b32 dpp row_shr", "v_mov_b32 dpp wave_shr", "v_add_f32 dpp row_shr", "v_mbcnt_lo", "v_readlane_b32", "v_readfirstlane_b32", "v_pk_add_u16",
    "v_cndmask_b32 e64 sgpr", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_u32_u24", "v_mul_f32 (sgpr src)", "v_add_f32 (literal)", "v_sub_f32", "v_min_f32", "v_fma_f32 (3 vgpr)", "v_mad_u64_u32",
    "v_bcnt_u32_b32", "v_ffbl_b32", "v_ffbh_u32", "v_cvt_u32_f32", "v_floor_f32", "v_fract_f32", "v_trunc_f32", "v_rndne_f32", "v_or3_b32", "v_xad_u32", "v_add_lshl_u32", "v_sqrt_f64", "v_ldexp_f32", "v_mul_legacy_f32", "v_cmp_class_f32", "v_max_u32", "v_min_i32", "v_subrev_u32", "v_addc_co_u32", "v_mov_b32 (sgpr)", "v_not_b32", "v_cvt_f16_f32", "v_bfi_b32", "v_mad_i32_i24", "v_fma_mix_f32", "s_add_u32 (salu)",
};

struct State {
    float a[CH]; f2 p[CH]; double d[CH]; unsigned u[CH]; unsigned long long w[CH];
    float c0, c1, s0; f2 pc0, pc1; double dc0, dc1; unsigned uc0, uc1, uc1s, sacc; unsigned long long smask, sink;
};
template <int OP> __device__ __forceinline__ void emit(State &S, const int i) {
                if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.c1));
                else if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.c0));
                else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(S.a[i]) : "v"(S.c0), "v"(S.c1));
                else if constexpr (OP == FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(S.a[i]) : "v"(S.c0), "v"(S.c1));
                else if constexpr (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.c1));
                else if constexpr (OP == MED3_F32) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(S.a[i]) : "v"(S.c0), "v"(S.c1));
                else if constexpr (OP == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(S.p[i]) : "v"(S.pc1));
                else if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(S.p[i]) : "v"(S.pc0));
                else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(S.p[i]) : "v"(S.pc0), "v"(S.pc1));
                else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(S.d[i]) : "v"(S.dc1));
                else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(S.d[i]) : "v"(S.dc0));
                else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(S.d[i]) : "v"(S.dc0), "v"(S.dc1));
                else if constexpr (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(S.d[i]) : "v"(S.a[i]));
                else if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(S.a[i]) : "v"(S.d[i]));
                else if constexpr (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(S.d[i]));
                else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == LSHL_B32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(S.u[i]));
                else if constexpr (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(S.u[i]));
                else if constexpr (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == MED3_U32) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == ADD_CO_U32) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(S.u[i]) : "v"(S.uc1) : "vcc");
                else if constexpr (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(S.w[i]));
                else if constexpr (OP == SAD_U8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == SAD_HI_U8) asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == MSAD_U8) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == CMP_F32_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(S.a[i]), "v"(S.c1) : "vcc");
                else if constexpr (OP == CMP_F32_SGPR) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(S.w[i]) : "v"(S.a[i]), "v"(S.c1));
                else if constexpr (OP == CMP_U32_VCC) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(S.u[i]), "v"(S.uc1) : "vcc");
                else if constexpr (OP == CMPX_F32) asm volatile("v_cmpx_ge_f32 %0, %0" :: "v"(S.c1) : "exec");
                else if constexpr (OP == MOV_DPP_ROW) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(S.u[i]));
                else if constexpr (OP == MOV_DPP_WAVE) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(S.u[i]));
                else if constexpr (OP == ADD_F32_DPP) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(S.a[i]) : "v"(S.c1));
                else if constexpr (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == READLANE) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(S.u[i])); S.sink += s; }
                else if constexpr (OP == READFIRSTLANE) { unsigned s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(S.u[i])); S.sink += s; }
                else if constexpr (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc1), "s"(S.smask));
                else if constexpr (OP == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == LSHR_B32) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(S.u[i]));
                else if constexpr (OP == ASHR_I32) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(S.u[i]));
                else if constexpr (OP == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == MUL_F32_SGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(S.a[i]) : "s"(S.s0));
                else if constexpr (OP == ADD_F32_LIT) asm volatile("v_add_f32 %0, 0x3f000001, %0" : "+v"(S.a[i]));
                else if constexpr (OP == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.c1));
                else if constexpr (OP == MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.c1));
                else if constexpr (OP == FMA_F32_3V) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(S.a[i]) : "v"(S.a[(i + 1) % CH]), "v"(S.c0), "v"(S.c1));
                else if constexpr (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(S.w[i]) : "v"(S.uc0), "v"(S.uc1) : "vcc");
                else if constexpr (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(S.u[i]));
                else if constexpr (OP == FFBH) asm volatile("v_ffbh_u32 %0, %0" : "+v"(S.u[i]));
                else if constexpr (OP == CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == FLOOR_F32) asm volatile("v_floor_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == FRACT_F32) asm volatile("v_fract_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == TRUNC_F32) asm volatile("v_trunc_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == RNDNE_F32) asm volatile("v_rndne_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == OR3_B32) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == XAD_U32) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == ADD_LSHL_U32) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == SQRT_F64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(S.d[i]));
                else if constexpr (OP == LDEXP_F32) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.uc1));
                else if constexpr (OP == MUL_LEGACY) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(S.a[i]) : "v"(S.c0));
                else if constexpr (OP == CMP_CLASS) asm volatile("v_cmp_class_f32 vcc, %0, %1" :: "v"(S.a[i]), "v"(S.uc1) : "vcc");
                else if constexpr (OP == MAX_U32) asm volatile("v_max_u32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(S.u[i]) : "v"(S.uc0));
                else if constexpr (OP == SUBREV_U32) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(S.u[i]) : "v"(S.uc1));
                else if constexpr (OP == ADDC_U32) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(S.u[i]) : "v"(S.uc1) : "vcc");
                else if constexpr (OP == MOV_SGPR) asm volatile("v_mov_b32 %0, %1" : "=v"(S.u[i]) : "s"(S.uc1s));
                else if constexpr (OP == NOT_B32) asm volatile("v_not_b32 %0, %0" : "+v"(S.u[i]));
                else if constexpr (OP == CVT_F16_F32) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(S.a[i]));
                else if constexpr (OP == BFI_B32) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == MAD_I32_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(S.u[i]) : "v"(S.uc0), "v"(S.uc1));
                else if constexpr (OP == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(S.a[i]) : "v"(S.c0), "v"(S.c1));
                else if constexpr (OP == S_ADD) asm volatile("s_add_u32 %0, %0, 1" : "+s"(S.sacc));
}

template <int OP, int OPB> __global__ __launch_bounds__(256) void k(float *out, float s0, float s1, unsigned long long *clk) {
    State S;
    S.c0 = s0 + threadIdx.x * 1e-9f; S.c1 = s1; S.s0 = s0;
    S.pc0 = (f2){S.c0, S.c0}; S.pc1 = (f2){S.c1, S.c1};
    S.dc0 = S.c0; S.dc1 = S.c1;
    S.uc0 = threadIdx.x | 1u; S.uc1 = (unsigned)(s1 * 7.0f) + 3u;
#pragma unroll
    for (int i = 0; i < CH; i++) {
        S.a[i] = threadIdx.x * 0.001f + i; S.p[i] = (f2){S.a[i], S.a[i] + 0.5f}; S.d[i] = S.a[i]; S.u[i] = threadIdx.x * 77u + i; S.w[i] = S.u[i];
    }
    S.sink = 0;
    S.smask = __ballot(threadIdx.x & 1);
    S.uc1s = (unsigned)(uintptr_t)out;
    S.sacc = (unsigned)(uintptr_t)clk;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(S.c0), "v"(S.a[0]) : "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < UNROLL; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (OPB < 0 || !(i & 1)) emit<OP>(S, i); else emit<OPB>(S, i);      // a pair alternates A B A B (chains 0 2 4 6 / 1 3 5 7)
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float r = (float)S.sink + (float)S.sacc;
#pragma unroll
    for (int i = 0; i < CH; i++) r += S.a[i] + S.p[i].x + S.p[i].y + (float)S.d[i] + (float)S.u[i] + (float)S.w[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
typedef void (*kern_t)(float *, float, float, unsigned long long *);
static const int PAIR_A[] = {ADD_F32, LSHL_B32, PK_MUL_F32, CMP_F32_VCC};
#define N_PAIR_A 4
template <int OP> struct Fill {
    static void go(kern_t *t, kern_t (*pt)[N_OPS]) {
        t[OP] = k<OP, -1>; pt[0][OP] = k<ADD_F32, OP>; pt[1][OP] = k<LSHL_B32, OP>; pt[2][OP] = k<PK_MUL_F32, OP>; pt[3][OP] = k<CMP_F32_VCC, OP>;
        Fill<OP + 1>::go(t, pt);
    }
};
template <> struct Fill<N_OPS> { static void go(kern_t *, kern_t (*)[N_OPS]) {} };

static hipEvent_t e0, e1;
static float *dout; static unsigned long long *clk; static std::vector<unsigned long long> h;
static int rate_khz, SIMDS, CUS;
static void measure(kern_t kern, int wps, double &cyc_simd, double &cyc_wave, double &mhz) {
    const int blocks = CUS * wps;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, 1.0001f, 0.5f, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;        // rep 0 warms the clock up
    }
    (void)hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double st = 0, sr = 0;
    for (int b = 0; b < blocks; b++) { st += (double)h[2 * b]; sr += (double)h[2 * b + 1]; }
    mhz = st / sr * (rate_khz * 1e-3);
    const double per_wave = (double)ITERS * UNROLL * CH;
    const double instr = per_wave * blocks * 4;
    cyc_simd = (double)best * 1e-3 * mhz * 1e6 * SIMDS / instr;
    cyc_wave = st / blocks / per_wave;
}

int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : nullptr;
    static kern_t table[N_OPS], ptable[N_PAIR_A][N_OPS];
    Fill<0>::go(table, ptable);
    const int max_blocks = 256 * 8;
    (void)hipMalloc(&dout, (size_t)max_blocks * 256 * 4);
    (void)hipMalloc(&clk, (size_t)max_blocks * 16);
    h.resize(max_blocks * 2);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    CUS = prop.multiProcessorCount; SIMDS = CUS * 4;
    printf("# device %s, %d CUs, clockRate attribute %d kHz, wall clock %d kHz\n", prop.gcnArchName, CUS, prop.clockRate, rate_khz);
    printf("# %d chains x %d unroll x %d trips per wave; 256-thread workgroups (one wave per SIMD), 256 * wps workgroups\n", CH, UNROLL, ITERS);
    printf("# TABLE 1: one op class alone.  cyc/SIMD = issue interval seen by a SIMD; cyc/wave = per wave; MHz = shader clock during the run\n");
    printf("%-26s", "op");
    for (int wps : {1, 2, 4, 8}) printf(" | wps=%d cyc/SIMD cyc/wave  MHz", wps);
    printf("\n");
    double alone[N_OPS];
    for (int op = 0; op < N_OPS; op++) {
        if (only && !strstr(names[op], only)) continue;
        printf("%-26s", names[op]);
        for (int wps : {1, 2, 4, 8}) {
            double cs, cw, mhz; measure(table[op], wps, cs, cw, mhz);
            printf(" | %13.2f %8.2f %5.0f", cs, cw, mhz);
            alone[op] = cs;
        }
        printf("\n"); fflush(stdout);
    }
    printf("# TABLE 2: pairs A B A B ... (8 waves per SIMD): cycles per PAIR seen by a SIMD, beside the sum and the max of the two alone.\n");
    printf("#          pair ~ max: the two classes issue to different units and overlap; pair ~ sum: same unit.\n");
    printf("%-26s", "B \\ A");
    for (int a = 0; a < N_PAIR_A; a++) printf(" | %-18s pair  sum  max", names[PAIR_A[a]]);
    printf("\n");
    for (int op = 0; op < N_OPS; op++) {
        if (only && !strstr(names[op], only)) continue;
        printf("%-26s", names[op]);
        for (int a = 0; a < N_PAIR_A; a++) {
            double cs, cw, mhz; measure(ptable[a][op], 8, cs, cw, mhz);
            const double A = alone[PAIR_A[a]], B = alone[op];
            printf(" | %18s %5.2f %5.2f %5.2f", "", 2 * cs, A + B, A > B ? A : B);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
