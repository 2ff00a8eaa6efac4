// Microbenchmark: what does the MOG2 access pattern (read F planes, write F
// planes, one dword per lane) sustain on MI355X as a function of the layout?
//   SoA   : plane f at base + f*np           (F far-apart streams per wave)
//   AoSoA : group g of 64 px holds F x 256 B back to back (one contiguous chunk)
// build: hipcc --offload-arch=gfx950 -O3 -o stream_layout stream_layout.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)

template <int F, bool AOSOA>
__global__ __launch_bounds__(256) void rw_kernel(float* __restrict__ st, size_t np, size_t npix)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    float v[F];
    if (AOSOA) {
        float* p = st + (i >> 6) * (size_t)(F * 64) + (i & 63);
#pragma unroll
        for (int f = 0; f < F; f++) v[f] = p[f * 64];
#pragma unroll
        for (int f = 0; f < F; f++) p[f * 64] = v[f] * 1.0001f + 1.f;
    } else {
        float* p = st + i;
#pragma unroll
        for (int f = 0; f < F; f++) v[f] = p[(size_t)f * np];
#pragma unroll
        for (int f = 0; f < F; f++) p[(size_t)f * np] = v[f] * 1.0001f + 1.f;
    }
}

template <int F, bool A>
float run(float* st, size_t np, size_t npix, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 g((unsigned)((npix + 255) / 256));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((rw_kernel<F, A>), g, dim3(256), 0, 0, st, np, npix);
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((rw_kernel<F, A>), g, dim3(256), 0, 0, st, np, npix);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main()
{
    const size_t npix = (size_t)8 * 1080 * 1920;      // 8 streams of 1080p
    const size_t np = npix;
    const int FMAX = 25;
    float* st; CK(hipMalloc(&st, np * FMAX * sizeof(float)));
    CK(hipMemset(st, 0, np * FMAX * sizeof(float)));
#define RUN(F) { float s = run<F,false>(st,np,npix,20), a = run<F,true>(st,np,npix,20); \
    double bytes = 2.0 * F * 4 * npix; \
    printf("F=%2d  SoA %8.1f us %7.1f GB/s   AoSoA %8.1f us %7.1f GB/s\n", F, s*1e3, bytes/s/1e6, a*1e3, bytes/a/1e6); }
    RUN(1) RUN(2) RUN(5) RUN(6) RUN(10) RUN(15) RUN(25)
    return 0;
}
