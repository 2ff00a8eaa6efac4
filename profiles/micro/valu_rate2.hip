// VALU issue-rate microbenchmark, second pass (round 3): which ENCODINGS are cheap on gfx950.
// valu_rate.hip (round 1) found 2.4 cycles for v_add_f32 / v_and_b32 (VOP2) and 4.2-4.9 for every
// three-operand VOP3 integer / select form; it did not separate "integer" from "VOP3", nor measure
// the VOP2 select (vcc), the VOP2 dot product (v_dot4c_i32_i8), v_swap_b32 or the SDWA forms.
// 8 waves per SIMD, four independent destination registers per kind, 64 instructions per asm block.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
// I(d) expands to one instruction on destination register operand %d (0..3); %4, %5 are sources
#define BLOCK4(I) REP16(I("0") I("1") I("2") I("3"))

enum {
    K_ADD_F32, K_MUL_F32, K_FMA_F32, K_FMAC_F32, K_FMA_2SRC, K_ADD_U32, K_SUB_U32, K_LSHL, K_LSHR, K_AND, K_XOR,
    K_MOV, K_MAX_F32, K_MED3_F32, K_MIN3_F32, K_CND_VCC, K_CND_E64, K_CND_E64_SAME, K_CMP_VCC, K_CMP_SGPR, K_CMPX,
    K_MUL_U24, K_MAD_U24, K_MUL_I24, K_DOT4_U8, K_DOT4_I8, K_DOT4C, K_DOT2C, K_ADD3, K_LSHL_ADD, K_AND_OR, K_BFE,
    K_PERM, K_ALIGNBYTE, K_CVT_F32_U32, K_CVT_UBYTE0, K_CVT_UBYTE2, K_CVT_SDWA, K_ADD_U32_SDWA, K_SWAP, K_MOV_B64,
    K_PK_MOV, K_PK_ADD_F32, K_PK_MUL_F32, K_PK_ADD_U16, K_PK_MAD_U16, K_SAD_U8, K_LERP, K_RCP, K_MIX_DOT_ADD,
    K_MIX_CND_ADD, K_MIX_DOT4C_FMAC, K_SUBREV_F32, K_MUL_LEGACY, K_CVT_U32_F32, K_ADD_CO, K_MBCNT, K_READLANE, K_COUNT
};

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 + 7, u3 = u0 ^ 5;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, p3 = {a0, a2};
    const float c = 1.0001f, d = 0.5f;
    const unsigned uc = u1 | 1u, ud = u2;
    const unsigned long long sel = __builtin_amdgcn_readfirstlane(iters) > 5 ? 0xF0F0F0F0F0F0F0F0ull : 0x0ull;
#define F_(ins) asm volatile(BLOCK4(ins) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d))
#define FV_(ins) asm volatile("v_cmp_lt_f32 vcc, %4, %5\n" BLOCK4(ins) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d) : "vcc")
#define FS_(ins) asm volatile(BLOCK4(ins) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d), "s"(sel))
#define U_(ins) asm volatile(BLOCK4(ins) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(uc), "v"(ud))
#define P_(ins) asm volatile(BLOCK4(ins) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0), "v"(p1))
    for (int i = 0; i < iters; i++) {
#define I(dst) "v_add_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_ADD_F32) F_(I);
#undef I
#define I(dst) "v_mul_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_MUL_F32) F_(I);
#undef I
#define I(dst) "v_fma_f32 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_FMA_F32) F_(I);
#undef I
#define I(dst) "v_fmac_f32 %" dst ", %4, %5\n"
        if (KIND == K_FMAC_F32) F_(I);
#undef I
#define I(dst) "v_fma_f32 %" dst ", %" dst ", %4, %" dst "\n"
        if (KIND == K_FMA_2SRC) F_(I);
#undef I
#define I(dst) "v_add_u32 %" dst ", %" dst ", %4\n"
        if (KIND == K_ADD_U32) U_(I);
#undef I
#define I(dst) "v_sub_u32 %" dst ", %" dst ", %4\n"
        if (KIND == K_SUB_U32) U_(I);
#undef I
#define I(dst) "v_lshlrev_b32 %" dst ", 1, %" dst "\n"
        if (KIND == K_LSHL) U_(I);
#undef I
#define I(dst) "v_lshrrev_b32 %" dst ", 1, %" dst "\n"
        if (KIND == K_LSHR) U_(I);
#undef I
#define I(dst) "v_and_b32 %" dst ", %" dst ", %4\n"
        if (KIND == K_AND) U_(I);
#undef I
#define I(dst) "v_xor_b32 %" dst ", %" dst ", %4\n"
        if (KIND == K_XOR) U_(I);
#undef I
#define I(dst) "v_mov_b32 %" dst ", %4\n"
        if (KIND == K_MOV) U_(I);
#undef I
#define I(dst) "v_max_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_MAX_F32) F_(I);
#undef I
#define I(dst) "v_med3_f32 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_MED3_F32) F_(I);
#undef I
#define I(dst) "v_min3_f32 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_MIN3_F32) F_(I);
#undef I
#define I(dst) "v_cndmask_b32 %" dst ", %" dst ", %4, vcc\n"
        if (KIND == K_CND_VCC) FV_(I);
#undef I
#define I(dst) "v_cndmask_b32_e64 %" dst ", %" dst ", %4, %6\n"
        if (KIND == K_CND_E64) FS_(I);
#undef I
#define I(dst) "v_cndmask_b32_e64 %" dst ", %" dst ", %" dst ", %6\n"
        if (KIND == K_CND_E64_SAME) FS_(I);
#undef I
#define I(dst) "v_cmp_lt_f32 vcc, %" dst ", %4\n"
        if (KIND == K_CMP_VCC) asm volatile(BLOCK4(I) : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(c) : "vcc");
#undef I
#define I(dst) "v_cmp_lt_f32_e64 s[20:21], %" dst ", %4\n"
        if (KIND == K_CMP_SGPR) asm volatile(BLOCK4(I) : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(c) : "s20", "s21");
#undef I
        if (KIND == K_CMPX) {
            // v_cmpx writes exec: compare values that keep every lane on (a >= -1 for all)
            asm volatile("s_mov_b64 s[20:21], exec\n" REP16("v_cmpx_ge_f32 vcc, %0, %4\n v_cmpx_ge_f32 vcc, %1, %4\n v_cmpx_ge_f32 vcc, %2, %4\n v_cmpx_ge_f32 vcc, %3, %4\n")
                         "s_mov_b64 exec, s[20:21]\n" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(-1.0f) : "vcc", "s20", "s21");
        }
#define I(dst) "v_mul_u32_u24 %" dst ", %" dst ", %4\n"
        if (KIND == K_MUL_U24) U_(I);
#undef I
#define I(dst) "v_mad_u32_u24 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_MAD_U24) U_(I);
#undef I
#define I(dst) "v_mul_i32_i24 %" dst ", %" dst ", %4\n"
        if (KIND == K_MUL_I24) U_(I);
#undef I
#define I(dst) "v_dot4_u32_u8 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_DOT4_U8) U_(I);
#undef I
#define I(dst) "v_dot4_i32_i8 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_DOT4_I8) U_(I);
#undef I
#define I(dst) "v_dot4c_i32_i8 %" dst ", %4, %5\n"
        if (KIND == K_DOT4C) U_(I);
#undef I
#define I(dst) "v_dot2c_i32_i16 %" dst ", %4, %5\n"
        if (KIND == K_DOT2C) U_(I);
#undef I
#define I(dst) "v_add3_u32 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_ADD3) U_(I);
#undef I
#define I(dst) "v_lshl_add_u32 %" dst ", %" dst ", 2, %5\n"
        if (KIND == K_LSHL_ADD) U_(I);
#undef I
#define I(dst) "v_and_or_b32 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_AND_OR) U_(I);
#undef I
#define I(dst) "v_bfe_u32 %" dst ", %" dst ", 1, 30\n"
        if (KIND == K_BFE) U_(I);
#undef I
#define I(dst) "v_perm_b32 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_PERM) U_(I);
#undef I
#define I(dst) "v_alignbyte_b32 %" dst ", %" dst ", %4, 1\n"
        if (KIND == K_ALIGNBYTE) U_(I);
#undef I
#define I(dst) "v_cvt_f32_u32 %" dst ", %" dst "\n"
        if (KIND == K_CVT_F32_U32) U_(I);
#undef I
#define I(dst) "v_cvt_f32_ubyte0 %" dst ", %" dst "\n"
        if (KIND == K_CVT_UBYTE0) U_(I);
#undef I
#define I(dst) "v_cvt_f32_ubyte2 %" dst ", %" dst "\n"
        if (KIND == K_CVT_UBYTE2) U_(I);
#undef I
#define I(dst) "v_cvt_f32_u32_sdwa %" dst ", %" dst " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
        if (KIND == K_CVT_SDWA) U_(I);
#undef I
#define I(dst) "v_add_u32_sdwa %" dst ", %" dst ", %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
        if (KIND == K_ADD_U32_SDWA) U_(I);
#undef I
        if (KIND == K_SWAP) asm volatile(REP16("v_swap_b32 %0, %1\n v_swap_b32 %2, %3\n v_swap_b32 %0, %2\n v_swap_b32 %1, %3\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
#define I(dst) "v_mov_b64 %" dst ", %4\n"
        if (KIND == K_MOV_B64) P_(I);
#undef I
#define I(dst) "v_pk_mov_b32 %" dst ", %4, %5\n"
        if (KIND == K_PK_MOV) P_(I);
#undef I
#define I(dst) "v_pk_add_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_PK_ADD_F32) P_(I);
#undef I
#define I(dst) "v_pk_mul_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_PK_MUL_F32) P_(I);
#undef I
#define I(dst) "v_pk_add_u16 %" dst ", %" dst ", %4\n"
        if (KIND == K_PK_ADD_U16) U_(I);
#undef I
#define I(dst) "v_pk_mad_u16 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_PK_MAD_U16) U_(I);
#undef I
#define I(dst) "v_sad_u8 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_SAD_U8) U_(I);
#undef I
#define I(dst) "v_lerp_u8 %" dst ", %" dst ", %4, %5\n"
        if (KIND == K_LERP) U_(I);
#undef I
#define I(dst) "v_rcp_f32 %" dst ", %" dst "\n"
        if (KIND == K_RCP) F_(I);
#undef I
        // mixes: do the costs of a slow and a fast kind add (one issue port) or overlap?
        if (KIND == K_MIX_DOT_ADD)
            asm volatile(REP16("v_dot4_u32_u8 %0, %0, %4, %5\n v_and_b32 %1, %1, %4\n v_dot4_u32_u8 %2, %2, %4, %5\n v_and_b32 %3, %3, %4\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(uc), "v"(ud));
        if (KIND == K_MIX_CND_ADD)
            asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %4, %6\n v_add_f32 %1, %1, %4\n v_cndmask_b32_e64 %2, %2, %4, %6\n v_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d), "s"(sel));
        if (KIND == K_MIX_DOT4C_FMAC)
            asm volatile(REP16("v_dot4c_i32_i8 %0, %4, %5\n v_dot4c_i32_i8 %1, %4, %5\n v_dot4c_i32_i8 %2, %4, %5\n v_mul_u32_u24 %3, %3, %4\n") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(uc), "v"(ud));
#define I(dst) "v_subrev_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_SUBREV_F32) F_(I);
#undef I
#define I(dst) "v_mul_legacy_f32 %" dst ", %" dst ", %4\n"
        if (KIND == K_MUL_LEGACY) F_(I);
#undef I
#define I(dst) "v_cvt_u32_f32 %" dst ", %" dst "\n"
        if (KIND == K_CVT_U32_F32) U_(I);
#undef I
#define I(dst) "v_add_co_u32 %" dst ", vcc, %" dst ", %4\n"
        if (KIND == K_ADD_CO) asm volatile(BLOCK4(I) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(uc), "v"(ud) : "vcc");
#undef I
#define I(dst) "v_mbcnt_lo_u32_b32 %" dst ", %4, %" dst "\n"
        if (KIND == K_MBCNT) U_(I);
#undef I
        if (KIND == K_READLANE)
            asm volatile(REP16("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n") : : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "s20", "s21", "s22", "s23");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + (float)(u0 + u1 + u2 + u3);
}

template <int KIND>
void run(const char *name, float *out)
{
    const int iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 200);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = (double)iters * 64 * 8;   // 64 instructions per iteration per wave, 8 waves per SIMD
    printf("%-34s %8.3f ms  %.2f ns per wave-instruction per SIMD  (%.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    fflush(stdout);
}

#define RUN(K) run<K>(#K, out)
int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<K_ADD_F32>("(clock warm-up) K_ADD_F32", out);
    run<K_ADD_F32>("(clock warm-up) K_ADD_F32", out);
    RUN(K_ADD_F32); RUN(K_MUL_F32); RUN(K_SUBREV_F32); RUN(K_MUL_LEGACY); RUN(K_FMA_F32); RUN(K_FMAC_F32); RUN(K_FMA_2SRC);
    RUN(K_ADD_U32); RUN(K_SUB_U32); RUN(K_ADD_CO); RUN(K_LSHL); RUN(K_LSHR); RUN(K_AND); RUN(K_XOR); RUN(K_MOV);
    RUN(K_MAX_F32); RUN(K_MED3_F32); RUN(K_MIN3_F32);
    RUN(K_CND_VCC); RUN(K_CND_E64); RUN(K_CND_E64_SAME); RUN(K_CMP_VCC); RUN(K_CMP_SGPR); RUN(K_CMPX);
    RUN(K_MUL_U24); RUN(K_MAD_U24); RUN(K_MUL_I24); RUN(K_DOT4_U8); RUN(K_DOT4_I8); RUN(K_DOT4C); RUN(K_DOT2C);
    RUN(K_ADD3); RUN(K_LSHL_ADD); RUN(K_AND_OR); RUN(K_BFE); RUN(K_PERM); RUN(K_ALIGNBYTE);
    RUN(K_CVT_F32_U32); RUN(K_CVT_U32_F32); RUN(K_CVT_UBYTE0); RUN(K_CVT_UBYTE2); RUN(K_CVT_SDWA); RUN(K_ADD_U32_SDWA);
    RUN(K_SWAP); RUN(K_MOV_B64); RUN(K_PK_MOV); RUN(K_PK_ADD_F32); RUN(K_PK_MUL_F32); RUN(K_PK_ADD_U16); RUN(K_PK_MAD_U16);
    RUN(K_SAD_U8); RUN(K_LERP); RUN(K_RCP); RUN(K_MBCNT); RUN(K_READLANE);
    RUN(K_MIX_DOT_ADD); RUN(K_MIX_CND_ADD); RUN(K_MIX_DOT4C_FMAC);
    return 0;
}
