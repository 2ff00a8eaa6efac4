// Microbenchmark: read-modify-write of a MOG2-like model, K modes of {weight, var, mu0, mu1, mu2}
// per pixel, 8 x 1080p pixels, one pixel per lane, as
//   SoA5   : five dword planes per mode (the r01 layout: 5 x global_load_dword per mode)
//   W+V4   : a weight dword plane + one float4 plane {var, mu0, mu1, mu2} per mode
//            (1 x global_load_dword + 1 x global_load_dwordx4 per mode, 1 KiB per wave-instruction)
//   V4only : the float4 plane alone (what a settled pixel touches when its weight plane is skipped)
//   plus the same with non-temporal stores.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_layout4 stream_layout4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)

template <int K, int MODE, bool NT>   // MODE 0 SoA5, 1 W+V4, 2 V4 only
__global__ __launch_bounds__(256) void rw_kernel(float* __restrict__ st, size_t np, size_t npix)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    if (MODE == 0) {
        float v[K * 5];
#pragma unroll
        for (int f = 0; f < K * 5; f++) v[f] = st[(size_t)f * np + i];
#pragma unroll
        for (int f = 0; f < K * 5; f++) {
            float o = v[f] * 1.0001f + 1.f;
            if (NT) __builtin_nontemporal_store(o, st + (size_t)f * np + i); else st[(size_t)f * np + i] = o;
        }
    } else {
        float w[K]; float4 q[K];
#pragma unroll
        for (int m = 0; m < K; m++) {
            float* base = st + (size_t)m * 5 * np;
            if (MODE == 1) w[m] = base[i];
            q[m] = ((const float4*)(base + np))[i];
        }
#pragma unroll
        for (int m = 0; m < K; m++) {
            float* base = st + (size_t)m * 5 * np;
            float4 o = make_float4(q[m].x * 1.0001f + 1.f, q[m].y * 1.0001f + 1.f, q[m].z * 1.0001f + 1.f, q[m].w * 1.0001f + 1.f);
            if (MODE == 1) { float ow = w[m] * 1.0001f + 1.f; if (NT) __builtin_nontemporal_store(ow, base + i); else base[i] = ow; }
            float4* d = (float4*)(base + np) + i;
            if (NT) { __builtin_nontemporal_store(o.x, &d->x); __builtin_nontemporal_store(o.y, &d->y); __builtin_nontemporal_store(o.z, &d->z); __builtin_nontemporal_store(o.w, &d->w); }
            else *d = o;
        }
    }
}

template <int K, int MODE, bool NT>
float run(float* st, size_t np, size_t npix, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 g((unsigned)((npix + 255) / 256));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((rw_kernel<K, MODE, NT>), g, dim3(256), 0, 0, st, np, npix);
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((rw_kernel<K, MODE, NT>), g, dim3(256), 0, 0, st, np, npix);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main()
{
    const size_t npix = (size_t)8 * 1080 * 1920;
    const size_t np = npix;
    float* st; CK(hipMalloc(&st, np * 25 * sizeof(float)));
    CK(hipMemset(st, 0, np * 25 * sizeof(float)));
#define ROW(K) { \
    float a = run<K,0,false>(st,np,npix,20), b = run<K,1,false>(st,np,npix,20), c = run<K,2,false>(st,np,npix,20); \
    float an = run<K,0,true>(st,np,npix,20), bn = run<K,1,true>(st,np,npix,20), cn = run<K,2,true>(st,np,npix,20); \
    double b5 = 2.0 * K * 20 * npix, b4 = 2.0 * K * 16 * npix; \
    printf("K=%d  SoA5 %7.1f us %6.0f GB/s | W+V4 %7.1f us %6.0f GB/s | V4 %7.1f us %6.0f GB/s || nt: SoA5 %6.0f  W+V4 %6.0f  V4 %6.0f GB/s\n", \
           K, a*1e3, b5/a/1e6, b*1e3, b5/b/1e6, c*1e3, b4/c/1e6, b5/an/1e6, b5/bn/1e6, b4/cn/1e6); }
    ROW(1) ROW(2) ROW(3) ROW(5)
    return 0;
}
