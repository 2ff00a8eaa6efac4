// K-nearest-neighbours background subtractor for gfx950: what GOFPID(frg_detect='KNN') runs in
// place of cv.createBackgroundSubtractorKNN().apply(X) [REF pygofpid/pipeline.py:233-234, :470].
//
// cv2's model and rule [OPENCV-RECALL bgfg_KNN.cpp], restated in oracle/gof_oracle.c (gofo_knn_*):
// per pixel three sets (short / mid / long term) of nN samples, a sample = the pixel's channel
// bytes + a flag byte; background = at least kNN flagged samples within dist2Threshold, shadow =
// that many flagged samples passing the chromaticity test, else foreground; the sets are refreshed
// at staggered times, each pixel drawing its slot of the next cycle.  cv2 draws with rand() inside
// a row-parallel loop -- its own output is not reproducible -- so the draw here is a counter-based
// hash of (pixel, frame, set), the same function in the oracle: kernel and oracle agree bit for bit
// (masks and every model byte); against cv2 itself only the statistics can.
//
// Layout in HBM: one lane owns one pixel; every sample of every pixel is ONE dword (channel bytes
// + the flag in byte 3; two dwords for 4-channel frames), sample s of all pixels is a plane, so a
// wave reads a sample as one coalesced 256-B segment.  The check scans the planes in cv2's order
// and stops as soon as every pixel of the wave is decided: a wave of settled background reads 2-3
// of the 21 planes (cv2's early `return 1`, wave-uniform).  HBM-bound: 4 B per sample examined.
#include "gof_mog2_dev.h"   // FrameTiles / TapPatterns: the staged-tile 3x3 blur front end the fused form borrows

namespace gofk {

namespace {

__device__ __forceinline__ uint32_t knn_draw(uint32_t pixel, uint32_t frame, uint32_t set)
{
    // oracle/gof_oracle.c gofo_knn_draw: 31 bits from (pixel, frame, set)
    uint32_t x = pixel * 0x9E3779B1u ^ (frame * 0x85EBCA77u + set * 0xC2B2AE3Du);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x >> 1;
}

struct KnnArgs {
    const uint8_t *frame;    // [n_img][h][w][C]
    uint8_t *mask;           // [n_img][h][w]
    uint32_t *samples;       // [3 nN][WS] planes of pitch np (WS = 1 dword per sample, 2 for C = 4)
    uint32_t *next, *index;  // planes: bytes (short, mid, long, 0) per pixel
    size_t np, n;            // plane pitch, pixels
    uint32_t hw;             // pixels per image (the draw is keyed on the pixel within its image)
    int nN, knn;
    float Tb, tau;
    int detect_shadows, shadow_value;
    int short_counter, mid_counter, long_counter;   // this frame's counters (before the increment)
    int n_short, n_mid, n_long;                     // this frame's cycle lengths
    uint32_t frame_no;                              // 1-based
    // fused 3x3 blur (knn_blur3_kernel): geometry, tap table, vertical multipliers and rounding constant
    int H, W;
    uint32_t vky0, vky1, vround;
    uint32_t taps[kTapTableDwords];
};

template <int C>
struct KnnSample {
    float v[C];
    bool flag;
};

template <int C>
__device__ __forceinline__ KnnSample<C> knn_unpack(uint32_t w0, uint32_t w1)
{
    KnnSample<C> r;
#pragma unroll
    for (int c = 0; c < C; c++)
        r.v[c] = (float)((w0 >> (8 * (c & 3))) & 255u);
    r.flag = C <= 3 ? (w0 >> 24) != 0 : (w1 & 255u) != 0;
    return r;
}

template <int C>
__device__ __forceinline__ KnnSample<C> knn_load(const uint32_t *samples, size_t np, int s, size_t pix)
{
    constexpr int WS = C <= 3 ? 1 : 2;
    const uint32_t w0 = samples[(size_t)s * WS * np + pix];
    const uint32_t w1 = C <= 3 ? 0u : samples[((size_t)s * WS + 1) * np + pix];
    return knn_unpack<C>(w0, w1);
}

// What a pixel's frame needs whatever the pixel looks like: its bookkeeping words and the first two sample
// planes (no pixel can have its kNN >= 2 flagged neighbours before both are examined).  Requested at the
// kernel's start, so that the round trip lies under the frame staging / blur instead of behind it.
template <int C>
struct KnnPrefetch {
    uint32_t nx, ix, w[2][C <= 3 ? 1 : 2];
    bool two;   // wave-uniform: planes 0 and 1 are here
};

template <int C>
__device__ __forceinline__ KnnPrefetch<C> knn_prefetch(const KnnArgs &a, size_t pix)
{
    constexpr int WS = C <= 3 ? 1 : 2;
    KnnPrefetch<C> p;
    p.nx = a.next[pix];
    p.ix = a.index[pix];
    p.two = a.knn >= 2 && a.nN >= 1;   // (3 nN >= 3 planes)
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int q = 0; q < WS; q++)
            p.w[s][q] = p.two ? a.samples[((size_t)s * WS + q) * a.np + pix] : 0u;
    return p;
}

#ifndef GOF_KNN_SCAN
#define GOF_KNN_SCAN 7
#endif
constexpr int kScan = GOF_KNN_SCAN;   // planes requested together once a wave scans beyond the first two

// One pixel of cv2's KNNInvoker: check (background / shadow / foreground), refresh, new slots.
// data / dbytes: the pixel's channels as floats / packed bytes; pix: its index in the planes;
// valid == false: a lane that shadows a pixel (all of its stores are masked).
// NS: the number of sample planes when it is known at compile time (21 = cv2's 3 x 7, and kNN >= 2: the launcher
// checks), else 0.  With NS known the planes a wave has read stay in registers (NS dwords per pixel), and the
// shadow loop -- which only ever runs in a wave that has read them all: a pixel that is not background has
// been through every plane -- reads nothing again.
template <int C, int NS = 0>
__device__ __forceinline__ void knn_pixel(const KnnArgs &a, size_t pix, bool valid, const float (&data)[C], uint32_t dbytes,
                                          const KnnPrefetch<C> &pre)
{
    constexpr int WS = C <= 3 ? 1 : 2;
    const int ns = NS ? NS : 3 * a.nN;
    uint32_t kept[NS ? NS : 1][WS];   // NS != 0: the planes read so far

    const uint32_t nx = pre.nx, ix = pre.ix;

    // ---- _cvCheckPixelBackgroundNP, first loop: neighbours within Tb, flagged ones among them
    int Pbf = 0, Pb = 0;
    bool bg = false;
    auto examine = [&](const KnnSample<C> &m) {
        float dist2;
        if (C == 3) {
            const float d0 = m.v[0] - data[0], d1 = m.v[1] - data[1], d2 = m.v[2] - data[2];
            dist2 = d0 * d0 + d1 * d1 + d2 * d2;
        } else {
            dist2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float d = m.v[c] - data[c];
                dist2 += d * d;
            }
        }
        if (!bg && dist2 < a.Tb) {
            Pbf++;
            if (m.flag) {
                Pb++;
                if (Pb >= a.knn)
                    bg = true;
            }
        }
    };
    int s = 0;
    if (NS || pre.two) {
        examine(knn_unpack<C>(pre.w[0][0], pre.w[0][WS - 1]));
        examine(knn_unpack<C>(pre.w[1][0], pre.w[1][WS - 1]));
        s = 2;
    }
    if constexpr (NS != 0) {
#pragma unroll
        for (int q = 0; q < WS; q++) {
            kept[0][q] = pre.w[0][q];
            kept[1][q] = pre.w[1][q];
        }
        // the same groups, with compile-time plane numbers: every plane lands in its own registers
        bool go = true;
#pragma unroll
        for (int g = 2; g < NS; g += kScan) {
            go = go && __builtin_amdgcn_ballot_w64(!bg) != 0;
            if (go) {
#pragma unroll
                for (int j = 0; j < kScan; j++)
                    if (g + j < NS) {
#pragma unroll
                        for (int q = 0; q < WS; q++)
                            kept[g + j][q] = a.samples[((size_t)(g + j) * WS + q) * a.np + pix];
                    }
#pragma unroll
                for (int j = 0; j < kScan; j++)
                    if (g + j < NS && (j == 0 || __builtin_amdgcn_ballot_w64(!bg) != 0))
                        examine(knn_unpack<C>(kept[g + j][0], kept[g + j][WS - 1]));
            }
        }
        s = ns;
    }
    // A wave that is still looking after those reads on in groups of kScan planes requested together
    // (examined in order, with the exit test between them: what a group fetched in vain is at most
    // kScan - 1 planes, and only waves that hold a pixel off the settled background get here)
    while (s < ns) {
        if (__builtin_amdgcn_ballot_w64(!bg) == 0)
            break;   // every pixel of the wave has its kNN flagged neighbours: cv2's `return 1`
        uint32_t w[kScan][WS];
#pragma unroll
        for (int j = 0; j < kScan; j++) {
            const int sj = s + j < ns ? s + j : ns - 1;
#pragma unroll
            for (int q = 0; q < WS; q++)
                w[j][q] = a.samples[((size_t)sj * WS + q) * a.np + pix];
        }
#pragma unroll
        for (int j = 0; j < kScan; j++) {
            if (s + j >= ns || (j > 0 && __builtin_amdgcn_ballot_w64(!bg) == 0))
                break;
            examine(knn_unpack<C>(w[j][0], w[j][WS - 1]));
        }
        s += kScan;
    }
    const bool include = bg || Pbf >= a.knn;

    // ---- second loop (not background): shadow = kNN flagged samples that explain the pixel as a
    //      darker version of themselves; a flagged all-zero sample ends the test (cv2's `return 0`)
    int result = bg ? 1 : 0;
    if (a.detect_shadows && __builtin_amdgcn_ballot_w64(!bg) != 0) {
        bool done = bg;
        int Ps = 0;
        auto shadow = [&](const KnnSample<C> &m) {
            if (!done && m.flag) {
                float num = 0.f, den = 0.f;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    num += data[c] * m.v[c];
                    den += m.v[c] * m.v[c];
                }
                if (den == 0.f) {
                    done = true;
                } else if (num <= den && num >= a.tau * den) {
                    const float al = num / den;
                    float dist2a = 0.f;
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float dD = al * m.v[c] - data[c];
                        dist2a += dD * dD;
                    }
                    if (dist2a < a.Tb * al * al) {
                        Ps++;
                        if (Ps >= a.knn) {
                            result = 2;
                            done = true;
                        }
                    }
                }
            }
        };
        if constexpr (NS != 0) {
            // (some pixel of the wave is not background: the wave has been through all NS planes above)
            bool go = true;
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) {
                go = go && __builtin_amdgcn_ballot_w64(!done) != 0;
                if (go)
                    shadow(knn_unpack<C>(kept[s2][0], kept[s2][WS - 1]));
            }
        }
        for (int s2 = NS ? ns : 0; s2 < ns; s2 += kScan) {
            if (__builtin_amdgcn_ballot_w64(!done) == 0)
                break;
            uint32_t w[kScan][WS];
#pragma unroll
            for (int j = 0; j < kScan; j++) {
                const int sj = s2 + j < ns ? s2 + j : ns - 1;
#pragma unroll
                for (int q = 0; q < WS; q++)
                    w[j][q] = a.samples[((size_t)sj * WS + q) * a.np + pix];
            }
#pragma unroll
            for (int j = 0; j < kScan; j++) {
                if (s2 + j >= ns || (j > 0 && __builtin_amdgcn_ballot_w64(!done) == 0))
                    break;
                shadow(knn_unpack<C>(w[j][0], w[j][WS - 1]));
            }
        }
    }

    // ---- _cvUpdatePixelBackgroundNP: long <- the mid set's oldest, mid <- the short set's oldest,
    //      short <- this pixel (offsets taken before any of the three); new slots at a cycle's end
    uint32_t nS = nx & 255u, nM = (nx >> 8) & 255u, nL = (nx >> 16) & 255u;
    uint32_t iS = ix & 255u, iM = (ix >> 8) & 255u, iL = (ix >> 16) & 255u;
    const uint32_t img_pix = (uint32_t)(pix % a.hw);
    const size_t oS = (size_t)iS * WS * a.np + pix, oM = (size_t)(iM + a.nN) * WS * a.np + pix,
                 oL = (size_t)(iL + 2 * a.nN) * WS * a.np + pix;
    const uint32_t last = (uint32_t)a.nN - 1u;
    if (valid && nL == (uint32_t)a.long_counter) {
#pragma unroll
        for (int q = 0; q < WS; q++)
            a.samples[oL + (size_t)q * a.np] = a.samples[oM + (size_t)q * a.np];
        iL = iL >= last ? 0u : iL + 1u;
    }
    if (a.long_counter == a.n_long - 1)
        nL = (knn_draw(img_pix, a.frame_no, 2u) % (uint32_t)a.n_long) & 255u;
    if (valid && nM == (uint32_t)a.mid_counter) {
#pragma unroll
        for (int q = 0; q < WS; q++)
            a.samples[oM + (size_t)q * a.np] = a.samples[oS + (size_t)q * a.np];
        iM = iM >= last ? 0u : iM + 1u;
    }
    if (a.mid_counter == a.n_mid - 1)
        nM = (knn_draw(img_pix, a.frame_no, 1u) % (uint32_t)a.n_mid) & 255u;
    if (valid && nS == (uint32_t)a.short_counter) {
        if (C <= 3) {
            a.samples[oS] = dbytes | (include ? 1u << 24 : 0u);
        } else {
            a.samples[oS] = dbytes;
            a.samples[oS + a.np] = include ? 1u : 0u;
        }
        iS = iS >= last ? 0u : iS + 1u;
    }
    if (a.short_counter == a.n_short - 1)
        nS = (knn_draw(img_pix, a.frame_no, 0u) % (uint32_t)a.n_short) & 255u;
    if (!valid)
        return;
    const uint32_t nx2 = nS | (nM << 8) | (nL << 16), ix2 = iS | (iM << 8) | (iL << 16);
    if (nx2 != nx)
        a.next[pix] = nx2;
    if (ix2 != ix)
        a.index[pix] = ix2;
    a.mask[pix] = result == 0 ? (uint8_t)255 : (result == 1 ? (uint8_t)0 : (uint8_t)a.shadow_value);
}

template <int C, int NS = 0>
__global__ __launch_bounds__(256) void knn_kernel(const KnnArgs a)
{
    const size_t pix0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = pix0 < a.n;
    const size_t pix = valid ? pix0 : a.n - 1;   // lanes past the end shadow the last pixel; stores masked
    const KnnPrefetch<C> pre = knn_prefetch<C>(a, pix);
    const uint8_t *px = a.frame + pix * C;
    uint32_t dbytes = 0;
    float data[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const uint32_t b = px[c];
        dbytes |= b << (8 * (c & 3));
        data[c] = (float)b;
    }
    knn_pixel<C, NS>(a, pix, valid, data, dbytes, pre);
}

// GaussianBlur / cv.blur 3x3 + KNN in one kernel [REF pipeline.py:463-470]: the staging and tap-pattern
// front end of the MOG2 kernels (64 x 4 px tiles, one wave per row, rows staged raw, the blur as dot
// products against per-lane tap patterns), then knn_pixel on the blurred pixel: the blurred frame never
// exists in memory.  Aligned frames only (fd_frames_supported); anything else runs blur_kernel + knn_kernel.
template <int C, int NS = 0>
__global__ __launch_bounds__(256) void knn_blur3_kernel(const KnnArgs a)
{
    typedef FrameTiles<C, 1> Tiles;
    constexpr int ROWD = Tiles::ROWD;
    __shared__ __attribute__((aligned(16))) uint32_t tiles[Tiles::LDS_DWORDS];
    const int H = a.H, W = a.W;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, img = blockIdx.z;
    const int x = x0 + lane, y = y0 + wave;
    Tiles ft;
    uint32_t raw[1][2][Tiles::NDW];
    ft.load(a.frame, 0, H, W, 1, img, x0, y0, wave, lane, raw);
    const bool valid = y < H && x < W;
    // a lane outside the image shadows the first pixel of its tile's row inside it (stores masked)
    const size_t pix = (size_t)img * H * W + (size_t)(y < H ? y : H - 1) * W + (x < W ? x : x0);
    const KnnPrefetch<C> pre = knn_prefetch<C>(a, pix);   // (in flight across the staging barrier and the blur)
    TapPatterns<C> taps;
    {
        typedef const KnnArgs __attribute__((address_space(4))) *ArgsPtr0;
        taps.load(valid ? x : x0, W, ft.sx0, ft.mis0, ((ArgsPtr0)__builtin_amdgcn_kernarg_segment_ptr())->taps);
    }
    Tiles::store(tiles, wave, lane, raw);
    __syncthreads();
    const uint32_t ky[3] = {a.vky0, a.vky1, a.vky0};
    uint32_t v[C];
    taps.template blur<ROWD>(tiles, (uint32_t)(wave * ROWD) * 4u + (uint32_t)taps.A, ky, a.vround, false, v);
    uint32_t dbytes = 0;
    float data[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        dbytes |= v[c] << (8 * (c & 3));
        data[c] = (float)v[c];
    }
    knn_pixel<C, NS>(a, pix, valid, data, dbytes, pre);
}

// the model as byte planes for export: samples[3 nN][C + 1][n], state[6][n]
template <int C>
__global__ __launch_bounds__(256) void knn_export_kernel(const uint32_t *samples, const uint32_t *next,
                                                         const uint32_t *index, size_t np, size_t n, int nN,
                                                         uint8_t *out_samples, uint8_t *out_state)
{
    constexpr int WS = C <= 3 ? 1 : 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    for (int s = 0; s < 3 * nN; s++) {
        const uint32_t w0 = samples[(size_t)s * WS * np + i];
        const uint32_t w1 = WS == 2 ? samples[((size_t)s * WS + 1) * np + i] : 0u;
        for (int c = 0; c < C; c++)
            out_samples[((size_t)s * (C + 1) + c) * n + i] = (uint8_t)(w0 >> (8 * (c & 3)));
        out_samples[((size_t)s * (C + 1) + C) * n + i] = C <= 3 ? (uint8_t)(w0 >> 24) : (uint8_t)w1;
    }
    const uint32_t nx = next[i], ix = index[i];
    for (int k = 0; k < 3; k++) {
        out_state[(size_t)k * n + i] = (uint8_t)(nx >> (8 * k));
        out_state[(size_t)(3 + k) * n + i] = (uint8_t)(ix >> (8 * k));
    }
}

}  // namespace

hipError_t knn_apply(const uint8_t *frame, uint8_t *mask, const KnnState &st, int n_img, int h, int w, int c,
                     const KnnFrame &f, hipStream_t s, const Taps *blur3)
{
    if (n_img < 1 || h < 1 || w < 1 || st.nN < 1 || st.nN > 16 || st.knn < 1 || !st.samples || !st.next ||
        !st.index || f.n_short < 1 || f.n_mid < 1 || f.n_long < 1)
        return hipErrorInvalidValue;
    const size_t n = (size_t)n_img * h * w;
    if (st.np < n || (n + 255) / 256 > 0x7fffffffull || (size_t)h * w > 0xffffffffull)
        return hipErrorInvalidValue;
    KnnArgs a;
    a.frame = frame;
    a.mask = mask;
    a.samples = st.samples;
    a.next = st.next;
    a.index = st.index;
    a.np = st.np;
    a.n = n;
    a.hw = (uint32_t)((size_t)h * w);
    a.nN = st.nN;
    a.knn = st.knn;
    a.Tb = st.Tb;
    a.tau = st.tau;
    a.detect_shadows = st.detect_shadows;
    a.shadow_value = st.shadow_value;
    a.short_counter = f.short_counter;
    a.mid_counter = f.mid_counter;
    a.long_counter = f.long_counter;
    a.n_short = f.n_short;
    a.n_mid = f.n_mid;
    a.n_long = f.n_long;
    a.frame_no = f.frame_no;
    a.H = h;
    a.W = w;
    a.vky0 = a.vky1 = a.vround = 0;
    // cv2's shape (3 x 7 samples, kNN >= 2): the planes a wave reads stay in registers (knn_pixel<C, 21>)
    const bool cv2_shape = st.nN == 7 && st.knn >= 2;
    if (blur3) {
        // the 3x3 blur inside the kernel (the caller has checked knn_blur3_supported)
        if (!knn_blur3_supported(frame, n_img, h, w, c, *blur3))
            return hipErrorInvalidValue;
        a.vky0 = blur3->vky[0];
        a.vky1 = blur3->vky[1];
        a.vround = blur3->vround;
        tap_table((uint32_t)blur3->kx[0] | ((uint32_t)blur3->kx[1] << 8) | ((uint32_t)blur3->kx[2] << 16), c, a.taps);
        const dim3 gb = grid3(w, h, n_img, TW, TH);
#define GOF_LAUNCH_KNN(KERNEL, CC, GRID)                                                     \
    do {                                                                                    \
        if (cv2_shape)                                                                      \
            hipLaunchKernelGGL((KERNEL<CC, 21>), GRID, dim3(256), 0, s, a);                  \
        else                                                                                \
            hipLaunchKernelGGL((KERNEL<CC, 0>), GRID, dim3(256), 0, s, a);                   \
    } while (0)
        switch (c) {
        case 1: GOF_LAUNCH_KNN(knn_blur3_kernel, 1, gb); break;
        case 3: GOF_LAUNCH_KNN(knn_blur3_kernel, 3, gb); break;
        case 4: GOF_LAUNCH_KNN(knn_blur3_kernel, 4, gb); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const dim3 g((unsigned)((n + 255) / 256));
    switch (c) {
    case 1: GOF_LAUNCH_KNN(knn_kernel, 1, g); break;
    case 3: GOF_LAUNCH_KNN(knn_kernel, 3, g); break;
    case 4: GOF_LAUNCH_KNN(knn_kernel, 4, g); break;
    default: return hipErrorInvalidValue;
    }
#undef GOF_LAUNCH_KNN
    return hipGetLastError();
}

bool knn_blur3_supported(const uint8_t *frame, int n_img, int h, int w, int c, const Taps &t)
{
    // as the fused frame-differencing kernel: byte taps, symmetric vertical multipliers, aligned rows / frames
    return fd_frames_supported(frame, n_img, h, w, c, t) && (h + TH - 1) / TH <= 65535 && n_img <= 65535;
}

hipError_t knn_export(const KnnState &st, int c, size_t n, uint8_t *d_samples, uint8_t *d_state, hipStream_t s)
{
    if (n < 1 || n > st.np || !d_samples || !d_state)
        return hipErrorInvalidValue;
    const dim3 g((unsigned)((n + 255) / 256));
    switch (c) {
    case 1: hipLaunchKernelGGL(knn_export_kernel<1>, g, dim3(256), 0, s, st.samples, st.next, st.index, st.np, n, st.nN, d_samples, d_state); break;
    case 3: hipLaunchKernelGGL(knn_export_kernel<3>, g, dim3(256), 0, s, st.samples, st.next, st.index, st.np, n, st.nN, d_samples, d_state); break;
    case 4: hipLaunchKernelGGL(knn_export_kernel<4>, g, dim3(256), 0, s, st.samples, st.next, st.index, st.np, n, st.nN, d_samples, d_state); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gofk
