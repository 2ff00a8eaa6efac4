// gfx950 (MI355X, CDNA4) kernels of the foreground path.  Written for wave64,
// 256-thread workgroups, SoA model planes read and written as fully coalesced
// 256-B wave segments.  All of these are HBM-bound stencils / per-pixel state
// machines: no MFMA (north_star), the work is coalescing, LDS staging and
// keeping the per-pixel model in registers.
//
// Compiled with -ffp-contract=off: the MOG2 update must round after every
// multiply and add exactly like the scalar C++ it restates; the
// float divisions are IEEE correctly rounded (hipcc default).
#include "gof_kernels.h"

#include <float.h>

namespace gofk {

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    // cv::borderInterpolate(BORDER_REFLECT_101) in closed form: period 2*len-2
    if (len == 1)
        return 0;
    const int period = 2 * len - 2;
    p %= period;
    if (p < 0)
        p += period;
    return p < len ? p : period - p;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------
// MOG2 (+ optional fused 3x3 fixed-point blur)
// ---------------------------------------------------------------------------
// Reference call sites: cv.GaussianBlur at pipeline.py:463-467 and
// cv.createBackgroundSubtractorMOG2().apply at pipeline.py:232/470; the
// arithmetic is cv2's, restated in oracle/gof_oracle.c.
//
// Tile = 64 px x 4 rows, one wave per row, one pixel per lane: every model
// plane access of a wave is one 256-B contiguous segment.

constexpr int TW = 64;
constexpr int TH = 4;

template <int C>
struct PixelModel {
    float w[kMaxModes];
    float var[kMaxModes];
    float mu[kMaxModes][C];
};

template <int C>
__device__ __forceinline__ void swap_modes(PixelModel<C> &M, int i, bool doit)
{
    // i is a compile-time constant after unrolling: registers, not scratch
    float a = M.w[i], b = M.w[i - 1];
    M.w[i] = doit ? b : a;
    M.w[i - 1] = doit ? a : b;
    a = M.var[i];
    b = M.var[i - 1];
    M.var[i] = doit ? b : a;
    M.var[i - 1] = doit ? a : b;
#pragma unroll
    for (int c = 0; c < C; c++) {
        a = M.mu[i][c];
        b = M.mu[i - 1][c];
        M.mu[i][c] = doit ? b : a;
        M.mu[i - 1][c] = doit ? a : b;
    }
}

// One pixel of MOG2Invoker::operator() on a register-resident model.
// Statement order follows oracle/gof_oracle.c mog2_pixel(); the only
// re-statement is that the fitted mode's new weight is stored in its slot
// BEFORE it is bubbled up, so that it travels with the swaps (OpenCV swaps
// first and then writes gmm[mode - swap_count].weight: same end state).
template <int C>
__device__ __forceinline__ uint8_t mog2_update(PixelModel<C> &M, int &nmodes_io, const float (&x)[C],
                                               const Mog2Consts &k)
{
    int nmodes = nmodes_io;
    bool background = false, fits = false;
    float totalWeight = 0.f;
    const float negprune = -k.prune;

#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m < nmodes) {  // live bound: nmodes shrinks when a mode is pruned
            float weight = k.alpha1 * M.w[m] + k.prune;
            bool fit_here = false;
            if (!fits) {
                float var = M.var[m];
                float d[C];
                float dist2;
                if (C == 3) {
                    d[0] = M.mu[m][0] - x[0];
                    d[1] = M.mu[m][1] - x[1];
                    d[2] = M.mu[m][2] - x[2];
                    dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                } else {
                    dist2 = 0.f;
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        d[c] = M.mu[m][c] - x[c];
                        dist2 += d[c] * d[c];
                    }
                }
                if (totalWeight < k.TB && dist2 < k.Tb * var)
                    background = true;
                if (dist2 < k.Tg * var) {
                    fits = true;
                    fit_here = true;
                    weight += k.alphaT;
                    float kk = k.alphaT / weight;
#pragma unroll
                    for (int c = 0; c < C; c++)
                        M.mu[m][c] -= kk * d[c];
                    float varnew = var + kk * (dist2 - var);
                    varnew = varnew > k.var_min ? varnew : k.var_min;
                    varnew = varnew > k.var_max ? k.var_max : varnew;
                    M.var[m] = varnew;
                }
            }
            float wstore = weight;
            if (weight < negprune) {
                wstore = 0.f;
                nmodes--;
            }
            M.w[m] = wstore;
            totalWeight += wstore;
            if (fit_here) {
                bool go = true;
#pragma unroll
                for (int i = m; i > 0; i--) {
                    go = go && !(weight < M.w[i - 1]);
                    swap_modes<C>(M, i, go);
                }
            }
        }
    }

    float inv = 0.f;
    if (fabsf(totalWeight) > FLT_EPSILON)
        inv = 1.f / totalWeight;
#pragma unroll
    for (int m = 0; m < kMaxModes; m++)
        if (m < nmodes)
            M.w[m] *= inv;

    if (!fits && k.alphaT > 0.f) {
        int slot = nmodes == k.nmix ? k.nmix - 1 : nmodes++;
        bool first = nmodes == 1;
#pragma unroll
        for (int m = 0; m < kMaxModes; m++) {
            if (m == slot) {
                M.w[m] = first ? 1.f : k.alphaT;
                M.var[m] = k.var_init;
#pragma unroll
                for (int c = 0; c < C; c++)
                    M.mu[m][c] = x[c];
            } else if (!first && m < nmodes - 1) {
                M.w[m] *= k.alpha1;
            }
        }
        bool go = true;
#pragma unroll
        for (int i = kMaxModes - 1; i > 0; i--) {
            bool in = i <= nmodes - 1;
            bool sw = in && go && !(k.alphaT < M.w[i - 1]);
            if (in)
                go = sw;
            swap_modes<C>(M, i, sw);
        }
    }
    nmodes_io = nmodes;

    if (background)
        return 0;
    bool shadow = false;
    if (k.detect_shadows) {
        float tW = 0.f;
        bool done = false;
#pragma unroll
        for (int m = 0; m < kMaxModes; m++) {
            if (!done && m < nmodes) {
                float num = 0.f, den = 0.f;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    num += x[c] * M.mu[m][c];
                    den += M.mu[m][c] * M.mu[m][c];
                }
                if (den == 0.f) {
                    done = true;
                } else {
                    if (num <= den && num >= k.tau * den) {
                        float a = num / den;
                        float dist2a = 0.f;
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            float dD = a * M.mu[m][c] - x[c];
                            dist2a += dD * dD;
                        }
                        if (dist2a < k.Tb * M.var[m] * a * a) {
                            shadow = true;
                            done = true;
                        }
                    }
                    if (!done) {
                        tW += M.w[m];
                        if (tW > k.TB)
                            done = true;
                    }
                }
            }
        }
    }
    return shadow ? (uint8_t)k.shadow_value : (uint8_t)255;
}

template <int C, bool BLUR3>
__global__ __launch_bounds__(256) void mog2_kernel(const uint8_t *__restrict__ frame,
                                                   uint8_t *__restrict__ mask,
                                                   float *__restrict__ planes,
                                                   uint8_t *__restrict__ modes, size_t np, int H,
                                                   int W, Mog2Consts k, uint32_t kx012,
                                                   uint32_t ky012)
{
    constexpr int F = 2 + C;  // floats per mode
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int img = blockIdx.z;
    const int x = x0 + lane, y = y0 + wave;
    const bool valid = x < W && y < H;
    const size_t N = (size_t)H * W;
    const uint8_t *fimg = frame + (size_t)img * N * C;

    // ---- stage the (TH+2) x (TW+2) px input tile, raw bytes, dword loads ----
    constexpr int ROWD = ((TW + 2) * C + 3 + 3) / 4;  // dwords per staged row incl. misalignment
    __shared__ uint32_t tile[BLUR3 ? (TH + 2) * ROWD : 1];
    const int sx0 = x0 > 0 ? x0 - 1 : 0;
    if (BLUR3) {
        const int sx1 = (x0 + TW + 1) < W ? (x0 + TW + 1) : W;
        const int nb = (sx1 - sx0) * C;
        for (int idx = threadIdx.x; idx < (TH + 2) * ROWD; idx += 256) {
            int r = idx / ROWD, d = idx - r * ROWD;
            int gy = clampi(reflect101(y0 - 1 + r, H), 0, H - 1);
            const uint8_t *a = fimg + ((size_t)gy * W + sx0) * C;
            int mis = (int)((uintptr_t)a & 3);
            int ndw = (mis + nb + 3) >> 2;
            if (d < ndw)
                tile[idx] = ((const uint32_t *)(a - mis))[d];
        }
    }

    // ---- issue the model loads early: mode count + mode 0 speculatively ----
    const size_t gi = (size_t)img * N + (size_t)(valid ? y : 0) * W + (valid ? x : 0);
    PixelModel<C> M;
    int nmodes = modes[gi];
    {
        const float *p = planes + gi;
        M.w[0] = p[0];
        M.var[0] = p[np];
#pragma unroll
        for (int c = 0; c < C; c++)
            M.mu[0][c] = p[(size_t)(2 + c) * np];
    }

    float xin[C];
    if (BLUR3) {
        __syncthreads();
        if (valid) {
            const int kx[3] = {(int)(kx012 & 1023), (int)((kx012 >> 10) & 1023), (int)(kx012 >> 20)};
            const int ky[3] = {(int)(ky012 & 1023), (int)((ky012 >> 10) & 1023), (int)(ky012 >> 20)};
            const int pl = (reflect101(x - 1, W) - sx0) * C, pc = (x - sx0) * C,
                      pr = (reflect101(x + 1, W) - sx0) * C;
            uint32_t acc[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                acc[c] = 32768u;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                int r = wave + j;
                int gy = clampi(reflect101(y0 - 1 + r, H), 0, H - 1);
                int mis = (int)((uintptr_t)(fimg + ((size_t)gy * W + sx0) * C) & 3);
                const uint8_t *row = (const uint8_t *)(tile + r * ROWD) + mis;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    uint32_t hsum = kx[0] * row[pl + c] + kx[1] * row[pc + c] + kx[2] * row[pr + c];
                    acc[c] += ky[j] * hsum;
                }
            }
#pragma unroll
            for (int c = 0; c < C; c++)
                xin[c] = (float)(acc[c] >> 16);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++)
            xin[c] = (float)fimg[((size_t)(valid ? y : 0) * W + (valid ? x : 0)) * C + c];
    }
    if (!valid)
        return;

    // ---- remaining modes, predicated on the pixel's own mode count ----
#pragma unroll
    for (int m = 1; m < kMaxModes; m++) {
        if (m < nmodes) {
            const float *p = planes + (size_t)m * F * np + gi;
            M.w[m] = p[0];
            M.var[m] = p[np];
#pragma unroll
            for (int c = 0; c < C; c++)
                M.mu[m][c] = p[(size_t)(2 + c) * np];
        } else {
            M.w[m] = 0.f;
            M.var[m] = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++)
                M.mu[m][c] = 0.f;
        }
    }

    uint8_t out = mog2_update<C>(M, nmodes, xin, k);

#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m < nmodes) {
            float *p = planes + (size_t)m * F * np + gi;
            p[0] = M.w[m];
            p[np] = M.var[m];
#pragma unroll
            for (int c = 0; c < C; c++)
                p[(size_t)(2 + c) * np] = M.mu[m][c];
        }
    }
    modes[gi] = (uint8_t)nmodes;
    mask[gi] = out;
}

// ---------------------------------------------------------------------------
// generic separable 8.8 fixed-point blur (any odd taps <= 31), REFLECT_101
// ---------------------------------------------------------------------------
// Replaces cv.GaussianBlur at pipeline.py:463-467 for every (ksize, sigma) the
// fused 3x3 path does not cover.  64 x 16 px output tile; the input tile is
// staged with the border already resolved, the horizontal pass lands in LDS as
// exact u16, the vertical pass rounds once.
constexpr int BTW = 64, BTH = 16;

template <int C>
__global__ __launch_bounds__(256) void blur_kernel(const uint8_t *__restrict__ src,
                                                   uint8_t *__restrict__ dst, int H, int W, Taps t)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int rx = t.nkx / 2, ry = t.nky / 2;
    const int inW = BTW + 2 * rx, inH = BTH + 2 * ry;
    uint8_t *tin = smem;                                          // [inH][inW][C]
    uint16_t *th = (uint16_t *)(smem + (((size_t)inH * inW * C + 15) & ~(size_t)15));  // [inH][BTW][C]
    const int x0 = blockIdx.x * BTW, y0 = blockIdx.y * BTH;
    const size_t N = (size_t)H * W;
    const uint8_t *simg = src + (size_t)blockIdx.z * N * C;
    uint8_t *dimg = dst + (size_t)blockIdx.z * N * C;

    for (int idx = threadIdx.x; idx < inH * inW; idx += 256) {
        int r = idx / inW, col = idx - r * inW;
        int gy = reflect101(y0 - ry + r, H);
        int gx = reflect101(x0 - rx + col, W);
        const uint8_t *p = simg + ((size_t)gy * W + gx) * C;
#pragma unroll
        for (int c = 0; c < C; c++)
            tin[(size_t)idx * C + c] = p[c];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < inH * BTW; idx += 256) {
        int r = idx / BTW, col = idx - r * BTW;
        uint32_t a[C];
#pragma unroll
        for (int c = 0; c < C; c++)
            a[c] = 0;
        for (int i = 0; i < t.nkx; i++) {
            const uint8_t *p = tin + ((size_t)r * inW + col + i) * C;
#pragma unroll
            for (int c = 0; c < C; c++)
                a[c] += (uint32_t)t.kx[i] * p[c];
        }
#pragma unroll
        for (int c = 0; c < C; c++)
            th[(size_t)idx * C + c] = (uint16_t)a[c];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BTH * BTW; idx += 256) {
        int r = idx / BTW, col = idx - r * BTW;
        int x = x0 + col, y = y0 + r;
        if (x >= W || y >= H)
            continue;
        uint32_t a[C];
#pragma unroll
        for (int c = 0; c < C; c++)
            a[c] = 32768u;
        for (int j = 0; j < t.nky; j++) {
            const uint16_t *p = th + ((size_t)(r + j) * BTW + col) * C;
#pragma unroll
            for (int c = 0; c < C; c++)
                a[c] += (uint32_t)t.ky[j] * p[c];
        }
        uint8_t *o = dimg + ((size_t)y * W + x) * C;
#pragma unroll
        for (int c = 0; c < C; c++) {
            uint32_t v = a[c] >> 16;
            o[c] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

// ---------------------------------------------------------------------------
// frame differencing  [REF segmentation.py:40-70]
// ---------------------------------------------------------------------------
// |X - prev| > floor(thr) -> 255 per channel, then the reference's
// np.mean(F, axis=-1, dtype=np.uint8): the channel sum wraps in uint8 and is
// then true-divided by C and truncated (colour masks are {0, 85, 84}).
template <int C>
__global__ __launch_bounds__(256) void fd_kernel(const uint8_t *__restrict__ cur,
                                                 const uint8_t *__restrict__ prev, int has_prev,
                                                 int ithr, uint8_t *__restrict__ mask, size_t npix)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix)
        return;
    unsigned acc = 0;
    if (has_prev) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            int a = cur[i * C + c], b = prev[i * C + c];
            int d = a > b ? a - b : b - a;
            acc += d > ithr ? 255u : 0u;
        }
        acc &= 255u;
    }
    // trunc(acc / float(C)) == acc / C for 0 <= acc <= 255
    mask[i] = (uint8_t)(acc / (unsigned)C);
}

// ---------------------------------------------------------------------------
// erode / dilate, arbitrary structuring element  (cv.erode / cv.dilate,
// pipeline.py:475)
// ---------------------------------------------------------------------------
constexpr int MTW = 64, MTH = 16;

__global__ __launch_bounds__(256) void morph_kernel(int op, const uint8_t *__restrict__ src,
                                                    uint8_t *__restrict__ dst, int H, int W,
                                                    SeBits se)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ay = se.kh / 2, ax = se.kw / 2;
    const int inW = MTW + se.kw - 1, inH = MTH + se.kh - 1;
    const int x0 = blockIdx.x * MTW, y0 = blockIdx.y * MTH;
    const size_t N = (size_t)H * W;
    const uint8_t *simg = src + (size_t)blockIdx.z * N;
    uint8_t *dimg = dst + (size_t)blockIdx.z * N;
    const uint8_t ident = op ? 0 : 255;  // out-of-image samples never win

    for (int idx = threadIdx.x; idx < inH * inW; idx += 256) {
        int r = idx / inW, col = idx - r * inW;
        int gy = y0 - ay + r, gx = x0 - ax + col;
        uint8_t v = ident;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = simg[(size_t)gy * W + gx];
        smem[idx] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < MTH * MTW; idx += 256) {
        int r = idx / MTW, col = idx - r * MTW;
        int x = x0 + col, y = y0 + r;
        if (x >= W || y >= H)
            continue;
        int v = ident;
        for (int dy = 0; dy < se.kh; dy++) {
            uint32_t bits = se.rows[dy];
            const uint8_t *p = smem + (size_t)(r + dy) * inW + col;
            for (int dx = 0; dx < se.kw; dx++) {
                if ((bits >> dx) & 1u) {
                    int s = p[dx];
                    v = op ? (s > v ? s : v) : (s < v ? s : v);
                }
            }
        }
        dimg[(size_t)y * W + x] = (uint8_t)v;
    }
}

// ---------------------------------------------------------------------------
// fused chain of rectangular erode/dilate ops (the reference default
// [erode 5x5, dilate 5x5], pipeline.py:177-186): one HBM read and one write of
// the mask, all intermediates in LDS.  Separable row-min / column-min per op.
// Between ops, positions outside the image are forced to the NEXT op's identity
// when they are read, which is exactly what cv2's constant border does to the
// intermediate image.
// ---------------------------------------------------------------------------
constexpr int CTW = 64, CTH = 32;

__global__ __launch_bounds__(256) void morph_chain_kernel(const uint8_t *__restrict__ src,
                                                          uint8_t *__restrict__ dst, int H, int W,
                                                          RectChain ch, int RX, int RY)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int inW = CTW + 2 * RX, inH = CTH + 2 * RY;
    uint8_t *A = smem;
    uint8_t *B = smem + (((size_t)inW * inH + 15) & ~(size_t)15);
    const int x0 = blockIdx.x * CTW - RX, y0 = blockIdx.y * CTH - RY;  // tile origin in image coords
    const size_t N = (size_t)H * W;
    const uint8_t *simg = src + (size_t)blockIdx.z * N;
    uint8_t *dimg = dst + (size_t)blockIdx.z * N;

    for (int idx = threadIdx.x; idx < inH * inW; idx += 256) {
        int r = idx / inW, col = idx - r * inW;
        int gy = y0 + r, gx = x0 + col;
        uint8_t v = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = simg[(size_t)gy * W + gx];
        A[idx] = v;
    }
    __syncthreads();

    int mx = 0, my = 0;  // margin already consumed on each side
    for (int o = 0; o < ch.n; o++) {
        const int op = ch.op[o], rx = ch.rx[o], ry = ch.ry[o];
        const int ident = op ? 0 : 255;
        // row pass A -> B over rows [my, inH-my), cols [mx+rx, inW-mx-rx)
        const int rows = inH - 2 * my, cols = inW - 2 * (mx + rx);
        for (int idx = threadIdx.x; idx < rows * cols; idx += 256) {
            int r = my + idx / cols, col = mx + rx + idx % cols;
            int gy = y0 + r;
            int v = ident;
            if (gy >= 0 && gy < H) {
                const uint8_t *p = A + (size_t)r * inW;
                for (int d = -rx; d <= rx; d++) {
                    int gx = x0 + col + d;
                    int s = (gx >= 0 && gx < W) ? p[col + d] : ident;
                    v = op ? (s > v ? s : v) : (s < v ? s : v);
                }
            }
            B[(size_t)r * inW + col] = (uint8_t)v;
        }
        __syncthreads();
        // column pass B -> A over rows [my+ry, inH-my-ry), same cols
        const int rows2 = inH - 2 * (my + ry);
        for (int idx = threadIdx.x; idx < rows2 * cols; idx += 256) {
            int r = my + ry + idx / cols, col = mx + rx + idx % cols;
            int v = ident;
            for (int d = -ry; d <= ry; d++) {
                int s = B[(size_t)(r + d) * inW + col];  // out-of-image rows hold ident already
                v = op ? (s > v ? s : v) : (s < v ? s : v);
            }
            A[(size_t)r * inW + col] = (uint8_t)v;
        }
        __syncthreads();
        mx += rx;
        my += ry;
    }
    for (int idx = threadIdx.x; idx < CTH * CTW; idx += 256) {
        int r = idx / CTW, col = idx - r * CTW;
        int gx = x0 + RX + col, gy = y0 + RY + r;
        if (gx < W && gy < H)
            dimg[(size_t)gy * W + gx] = A[(size_t)(r + RY) * inW + col + RX];
    }
}

__global__ __launch_bounds__(256) void sum_modes_kernel(const uint8_t *__restrict__ modes, size_t n,
                                                        unsigned long long *out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        s += modes[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(out, s);
}

inline dim3 grid3(int w, int h, int n, int tw, int th)
{
    return dim3((unsigned)((w + tw - 1) / tw), (unsigned)((h + th - 1) / th), (unsigned)n);
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers (shape checks live here: a kernel is never launched on operands
// its indexing does not cover)
// ---------------------------------------------------------------------------

hipError_t blur_sep_q8(const uint8_t *src, uint8_t *dst, int n_img, int h, int w, int c,
                       const Taps &t, hipStream_t s)
{
    if (n_img < 1 || h < 1 || w < 1 || t.nkx < 1 || t.nky < 1 || t.nkx > kMaxKsize ||
        t.nky > kMaxKsize || !(t.nkx & 1) || !(t.nky & 1) || (h + 15) / 16 > 65535 || n_img > 65535)
        return hipErrorInvalidValue;
    int inW = BTW + t.nkx - 1, inH = BTH + t.nky - 1;
    size_t lds = (((size_t)inH * inW * c + 15) & ~(size_t)15) + (size_t)inH * BTW * c * 2;
    dim3 g = grid3(w, h, n_img, BTW, BTH);
    switch (c) {
    case 1: hipLaunchKernelGGL(blur_kernel<1>, g, dim3(256), lds, s, src, dst, h, w, t); break;
    case 3: hipLaunchKernelGGL(blur_kernel<3>, g, dim3(256), lds, s, src, dst, h, w, t); break;
    case 4: hipLaunchKernelGGL(blur_kernel<4>, g, dim3(256), lds, s, src, dst, h, w, t); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t mog2_apply(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h,
                      int w, int c, const Mog2Consts &k, bool fused_blur3, const Taps &t,
                      hipStream_t s)
{
    if (n_img < 1 || h < 1 || w < 1 || (h + TH - 1) / TH > 65535 || n_img > 65535 ||
        st.np < (size_t)n_img * h * w || k.nmix < 1 || k.nmix > kMaxModes)
        return hipErrorInvalidValue;
    if (fused_blur3 && (t.nkx != 3 || t.nky != 3))
        return hipErrorInvalidValue;
    uint32_t kx = 0, ky = 0;
    if (fused_blur3) {
        kx = t.kx[0] | (t.kx[1] << 10) | (t.kx[2] << 20);
        ky = t.ky[0] | (t.ky[1] << 10) | (t.ky[2] << 20);
    }
    dim3 g = grid3(w, h, n_img, TW, TH);
#define GOF_LAUNCH_MOG2(CC, BB)                                                              \
    hipLaunchKernelGGL((mog2_kernel<CC, BB>), g, dim3(256), 0, s, frame, mask, st.planes,    \
                       st.modes, st.np, h, w, k, kx, ky)
    if (fused_blur3) {
        switch (c) {
        case 1: GOF_LAUNCH_MOG2(1, true); break;
        case 3: GOF_LAUNCH_MOG2(3, true); break;
        case 4: GOF_LAUNCH_MOG2(4, true); break;
        default: return hipErrorInvalidValue;
        }
    } else {
        switch (c) {
        case 1: GOF_LAUNCH_MOG2(1, false); break;
        case 3: GOF_LAUNCH_MOG2(3, false); break;
        case 4: GOF_LAUNCH_MOG2(4, false); break;
        default: return hipErrorInvalidValue;
        }
    }
#undef GOF_LAUNCH_MOG2
    return hipGetLastError();
}

hipError_t fd_apply(const uint8_t *cur, const uint8_t *prev, bool has_prev, int ithr,
                    uint8_t *mask, int n_img, int h, int w, int c, hipStream_t s)
{
    if (n_img < 1 || h < 1 || w < 1)
        return hipErrorInvalidValue;
    size_t npix = (size_t)n_img * h * w;
    size_t blocks = (npix + 255) / 256;
    if (blocks > 0x7fffffffu)
        return hipErrorInvalidValue;
    dim3 g((unsigned)blocks);
    switch (c) {
    case 1: hipLaunchKernelGGL(fd_kernel<1>, g, dim3(256), 0, s, cur, prev, (int)has_prev, ithr, mask, npix); break;
    case 3: hipLaunchKernelGGL(fd_kernel<3>, g, dim3(256), 0, s, cur, prev, (int)has_prev, ithr, mask, npix); break;
    case 4: hipLaunchKernelGGL(fd_kernel<4>, g, dim3(256), 0, s, cur, prev, (int)has_prev, ithr, mask, npix); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t morph_generic(int op, const uint8_t *src, uint8_t *dst, int n_img, int h, int w,
                         const SeBits &se, hipStream_t s)
{
    if (n_img < 1 || h < 1 || w < 1 || se.kh < 1 || se.kw < 1 || se.kh > kMaxKsize ||
        se.kw > kMaxKsize || (h + MTH - 1) / MTH > 65535 || n_img > 65535 || (op != 0 && op != 1))
        return hipErrorInvalidValue;
    size_t lds = (size_t)(MTW + se.kw - 1) * (MTH + se.kh - 1);
    hipLaunchKernelGGL(morph_kernel, grid3(w, h, n_img, MTW, MTH), dim3(256), lds, s, op, src, dst,
                       h, w, se);
    return hipGetLastError();
}

bool morph_rect_chain_supported(const RectChain &ch)
{
    if (ch.n < 1 || ch.n > 8)
        return false;
    int RX = 0, RY = 0;
    for (int i = 0; i < ch.n; i++) {
        if (ch.rx[i] < 0 || ch.ry[i] < 0 || (ch.op[i] != 0 && ch.op[i] != 1))
            return false;
        RX += ch.rx[i];
        RY += ch.ry[i];
    }
    size_t lds = 2 * ((((size_t)(CTW + 2 * RX) * (CTH + 2 * RY)) + 15) & ~(size_t)15);
    return lds <= 64 * 1024;
}

hipError_t morph_rect_chain(const uint8_t *src, uint8_t *dst, int n_img, int h, int w,
                            const RectChain &ch, hipStream_t s)
{
    if (n_img < 1 || h < 1 || w < 1 || !morph_rect_chain_supported(ch) ||
        (h + CTH - 1) / CTH > 65535 || n_img > 65535)
        return hipErrorInvalidValue;
    int RX = 0, RY = 0;
    for (int i = 0; i < ch.n; i++) {
        RX += ch.rx[i];
        RY += ch.ry[i];
    }
    size_t lds = 2 * ((((size_t)(CTW + 2 * RX) * (CTH + 2 * RY)) + 15) & ~(size_t)15);
    hipLaunchKernelGGL(morph_chain_kernel, grid3(w, h, n_img, CTW, CTH), dim3(256), lds, s, src,
                       dst, h, w, ch, RX, RY);
    return hipGetLastError();
}

hipError_t sum_modes(const uint8_t *modes, size_t n, unsigned long long *out, hipStream_t s)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048)
        blocks = 2048;
    if (blocks < 1)
        blocks = 1;
    hipLaunchKernelGGL(sum_modes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, modes, n, out);
    return hipGetLastError();
}

}  // namespace gofk
