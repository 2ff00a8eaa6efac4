// Which lane holds which element?  v_mfma_i32_16x16x64_i8 on gfx950, checked with exact integer data
// against a plain matrix product: the assumed map is, for lane l,
//   A: row l & 15, k = 16 (l >> 4) + j  (byte j of the lane's 16-byte fragment)
//   B: k = 16 (l >> 4) + j, column l & 15
//   D: column l & 15, row 4 (l >> 4) + i  (register i of 4)
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_i8_probe mfma_i8_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const int8_t *A, const int8_t *B, int *D)   // A[16][64], B[64][16] row-major, D[16][16]
{
    const int l = threadIdx.x;
    i32x4 a, b, c = {0, 0, 0, 0};
    int8_t fa[16], fb[16];
    for (int j = 0; j < 16; j++) {
        fa[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
        fb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
    }
    memcpy(&a, fa, 16);
    memcpy(&b, fb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++)
        D[(4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
}

int main()
{
    int8_t hA[16 * 64], hB[64 * 16];
    int hD[256], ref[256];
    unsigned s = 12345;
    for (int i = 0; i < 16 * 64; i++) { s = s * 1664525u + 1013904223u; hA[i] = (int8_t)((s >> 24) & 255); }
    for (int i = 0; i < 64 * 16; i++) { s = s * 1664525u + 1013904223u; hB[i] = (int8_t)((s >> 24) % 120); }   // asymmetric
    for (int m = 0; m < 16; m++)
        for (int n = 0; n < 16; n++) {
            int acc = 0;
            for (int kk = 0; kk < 64; kk++) acc += (int)hA[m * 64 + kk] * (int)hB[kk * 16 + n];
            ref[m * 16 + n] = acc;
        }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("v_mfma_i32_16x16x64_i8 with the assumed lane map: %d of 256 elements differ from the plain product%s\n", bad,
           bad ? "" : "  -> map confirmed");
    if (bad) for (int i = 0; i < 8; i++) printf("  D[%d] = %d, ref %d\n", i, hD[i], ref[i]);
    return bad != 0;
}
