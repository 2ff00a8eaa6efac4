// VALU issue-rate microbenchmark on gfx950: wave64 instructions per SIMD-cycle for the
// instruction kinds the MOG2 kernel is made of (it is issue-bound in its 8-frame form).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    unsigned u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 + 7, u3 = u0 ^ 5;
    const float c = 1.0001f, d = 0.5f;
    const unsigned long long sel = __builtin_amdgcn_readfirstlane(iters) > 5 ? 0xF0F0F0F0F0F0F0F0ull : 0x0ull;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0), "v"(p1));) }
        if (KIND == 2) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");) }
        if (KIND == 10) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "s"(sel));) }
        if (KIND == 11) { REP16(asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1), "v"(u2));) }
        if (KIND == 12) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 13) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(c) : "vcc");) }
        if (KIND == 14) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1));) }
        if (KIND == 3) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1), "v"(u2));) }
        if (KIND == 4) { REP16(asm volatile("v_dot4_u32_u8 %0, %0, %4, %5\n v_dot4_u32_u8 %1, %1, %4, %5\n v_dot4_u32_u8 %2, %2, %4, %5\n v_dot4_u32_u8 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1), "v"(u2));) }
        if (KIND == 5) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1), "v"(u2));) }
        if (KIND == 6) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 7) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, %5\n v_alignbyte_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1), "v"(u2));) }
        if (KIND == 8) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1));) }
        if (KIND == 9) { REP16(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + (float)(u0 + u1 + u2 + u3);
}
template <int KIND> void run(const char* name, float* out)
{
    const int iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts_per_simd = (double)iters * 64 * 8;            // 64 instr per iteration per wave, 8 waves per SIMD
    printf("%-16s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_cndmask_b32", out); run<3>("v_perm_b32", out);
    run<4>("v_dot4_u32_u8", out); run<5>("v_mad_u32_u24", out); run<6>("v_rcp_f32", out); run<7>("v_alignbyte_b32", out);
    run<8>("v_mul_lo_u32", out); run<9>("v_mov_dpp wshr", out);
    run<10>("v_cndmask e64", out); run<11>("v_bfi_b32", out); run<12>("v_add_f32", out); run<13>("v_cmp_lt_f32", out);
    run<14>("v_and_b32", out);
    return 0;
}
