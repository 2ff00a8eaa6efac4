// gfx950 (MI355X, CDNA4) kernels of the foreground path.  Written for wave64,
// 256-thread workgroups, SoA model planes read and written as fully coalesced
// 256-B wave segments.  All of these are byte / per-pixel state-machine stencils:
// no MFMA (north_star), the work is coalescing, LDS staging, keeping the per-pixel
// model in registers and, once that has taken the memory out of the way, the
// number of instructions a wave issues.
//
//   mog2_kernel          blur 3x3 + MOG2, one frame per launch (and every layout the
//                        fast paths do not cover)
//   mog2_frames_kernel   the same for up to 8 frames per launch (FrameTiles,
//                        TapPatterns, settled-wave loop)
//   fd_frames_kernel     blur 3x3 + frame differencing on the same front end
//   blur_kernel, fd_kernel, gray_kernel   the unfused steps
//   morph_stream_bits_kernel / morph_stream_kernel / morph_chain_kernel / morph_kernel
//                        erode / dilate chains (registers + DPP, LDS, any SE)
//
// Compiled with -ffp-contract=off: the MOG2 update must round after every
// multiply and add exactly like the scalar C++ it restates; the
// float divisions are IEEE correctly rounded (hipcc default).
#include "gof_mog2_dev.h"

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

// ---------------------------------------------------------------------------
// fd_frames_kernel: GaussianBlur 3x3 + FrameDifferencing.apply [REF pipeline.py:
// 463-470, segmentation.py:40-70] for T consecutive frames of every stream in one
// launch, on the staging / tap-pattern front end of mog2_frames_kernel.  The
// blurred frame never exists in memory: the lane keeps its blurred pixel of frame
// t - 1 in registers, and only the last one of the launch is written out (C bytes
// per pixel) for the next launch to start from.  Per pixel and frame: C bytes in,
// one mask byte out, 2 C / T for the hand-over -- against 6 C + 1 + 6 C for the
// separate blur, difference and copy-of-the-previous-frame kernels.
// ---------------------------------------------------------------------------
struct FdFramesArgs {
    const uint8_t *frame;
    uint8_t *mask;
    // the blurred last frame of the previous launch (if has_prev) / of this launch:
    // one dword per pixel, channel c in byte c (the kernels' own hand-over format)
    const uint32_t *prev_in;
    uint32_t *prev_out;
    uint8_t *rowflags;        // as Mog2FramesArgs::rowflags (or NULL)
    size_t frame_stride, mask_stride;
    int H, W, T, has_prev, ithr, flag_pitch;
    uint32_t kxw, vky0, vky1, vround;   // (as Mog2FramesArgs)
    int g0, g1, g2;                     // CS != 0: luma weights of source channels 0 / 1 / 2
};

// NT: frames staged per launch (1 for the per-frame call, else kMaxT).  CS: channels of the frames as
// they arrive when step 0 (cv.cvtColor to grey) is switched on -- the colour rows are staged and the
// grey tile is built in LDS (gray_tiles_from_staged), as in mog2_frames_kernel; C == 1 then.
template <int C, int NT, int CS = 0>
__global__ __launch_bounds__(256) void fd_frames_kernel(const FdFramesArgs a)
{
    static_assert(CS == 0 || C == 1, "the conversion produces one channel");
    typedef FrameTiles<(CS ? CS : C), NT> Tiles;
    constexpr int ROWD = CS ? kGrayRowD : Tiles::ROWD, TILE = (TH + 2) * ROWD;
    __shared__ uint32_t tiles[NT * TILE];
    __shared__ uint32_t ctiles[CS ? Tiles::LDS_DWORDS : 1];   // the raw colour tiles (CS != 0)

    const int H = a.H, W = a.W, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int img = blockIdx.z;
    const int x = x0 + lane, y = y0 + wave;
    const size_t N = (size_t)H * W;

    Tiles ft;
    uint32_t raw[NT][2][Tiles::NDW];
    ft.load(a.frame, a.frame_stride, H, W, T, img, x0, y0, wave, lane, raw);

    // a lane outside the image shadows a pixel inside it; only its stores are masked
    const bool valid = y < H && x < W;
    const unsigned xl = valid ? (unsigned)x : 0u;
    const size_t pix = (size_t)img * N + (size_t)(y < H ? y : 0) * W + xl;
    const uint32_t prev_px = a.has_prev ? a.prev_in[pix] : 0u;
    uint32_t prev[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        prev[c] = (prev_px >> (8 * c)) & 255u;

    Tiles::store(CS ? ctiles : tiles, wave, lane, raw);
    TapPatterns<C> taps;
    // (a grey tile row starts with the pixel itself: no offset within its first dword)
    taps.build(valid ? x : x0, W, ft.sx0, CS ? 0 : ft.mis0, a.kxw);
    uint32_t wa = (uint32_t)(wave * ROWD) * 4u + (uint32_t)taps.A;
    uint32_t half = a.vround;
    asm volatile("" : "+v"(half));
    const uint32_t ky[3] = {a.vky0, a.vky1, a.vky0};
    __syncthreads();
    if constexpr (CS != 0) {
        gray_tiles_from_staged<CS, NT>(ctiles, tiles, ft, x0, W, wave, lane, a.g0, a.g1, a.g2);
        __syncthreads();
    }

    uint8_t *mp = a.mask + pix;
    bool have = a.has_prev != 0;   // the first frame ever gives an all-zero mask
    uint32_t nz = 0;
    for (int ti = 0; ti < T; ti++) {
        uint32_t v[C];
        taps.template blur<ROWD>(tiles, wa, ky, half, true, v);
        wa += (uint32_t)TILE * 4u;
        // |X - prev| > floor(thr) -> 255 per channel; np.mean(..., dtype=uint8): the
        // channel sum wraps in uint8, then / C truncated (fd_kernel's statements)
        unsigned acc = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int d = (int)v[c] > (int)prev[c] ? (int)v[c] - (int)prev[c] : (int)prev[c] - (int)v[c];
            acc += d > a.ithr ? 255u : 0u;
            prev[c] = v[c];
        }
        acc &= 255u;
        const uint8_t out = have ? (uint8_t)(acc / (unsigned)C) : (uint8_t)0;
        have = true;
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
        if (valid)
            *mp = out;
        mp += a.mask_stride;
    }
    if (a.rowflags != nullptr && y < H && lane == 0)
        a.rowflags[((size_t)img * H + y) * a.flag_pitch + blockIdx.x] = (uint8_t)nz;
    if (valid) {
        uint32_t px = 0;
#pragma unroll
        for (int c = 0; c < C; c++)
            px |= prev[c] << (8 * c);
        a.prev_out[pix] = px;
    }
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
            uint32_t v;
            if (t.box_scale != 0.0) {
                // cv.blur: saturate_cast<uchar>(sum * scale) = round-half-even of the
                // double product [OPENCV-RECALL]; a[c] carries the +2^15 of the other form
                v = (uint32_t)__double2int_rn((double)(a[c] - 32768u) * t.box_scale);
            } else {
                v = a[c] >> 16;
            }
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
// the mask, all intermediates in LDS, 4 pixels per lane packed in a dword.
//
// Separable: per op a row pass (window 2*rx+1) A -> B and a column pass
// (window 2*ry+1) B -> A.  A shifted window operand is one v_alignbyte of two
// neighbouring LDS dwords; the byte-wise min/max runs as two v_pk_{min,max}_u16
// on the even and odd bytes.  Out-of-image bytes hold the CURRENT op's identity
// (255 for erode, 0 for dilate): written at staging time and re-written after
// every op, which is what cv2's constant border does to each intermediate image.
//
// Tile = 128 x 32 output px; the halo is Q = sum(ceil(rx/4)) dwords per side
// and RY = sum(ry) rows, so every op's compute range stays dword-granular.
// ---------------------------------------------------------------------------
constexpr int CTWD = 32;   // output tile width in dwords (128 px)
constexpr int CTH = 32;    // output tile height
constexpr int CMAXQ = 4;   // per-op horizontal reach in dwords: rx <= 15
constexpr int CMAXR = 15;  // per-op vertical radius

typedef unsigned short gof_us2 __attribute__((ext_vector_type(2)));

template <bool IS_MAX>
__device__ __forceinline__ uint32_t pk16(uint32_t a, uint32_t b)
{
    gof_us2 x = __builtin_bit_cast(gof_us2, a), y = __builtin_bit_cast(gof_us2, b);
    gof_us2 r = IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(uint32_t, r);
}

// accumulate one packed-u8 operand into the split (even bytes, odd bytes) accumulators
template <bool IS_MAX>
__device__ __forceinline__ void acc_u8x4(uint32_t &ae, uint32_t &ao, uint32_t v)
{
    ae = pk16<IS_MAX>(ae, v & 0x00FF00FFu);
    ao = pk16<IS_MAX>(ao, (v >> 8) & 0x00FF00FFu);
}

// bytes of the dword at (gy, gx0..gx0+3) that lie inside the image (gx0 % 4 == 0)
__device__ __forceinline__ uint32_t in_image_mask(int gy, int gx0, int H, int W)
{
    if (gy < 0 || gy >= H || gx0 < 0 || gx0 >= W)
        return 0u;
    int nvalid = W - gx0;
    return nvalid >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvalid)) - 1u);
}

// Strided walk of a rows x cols index space by 256 threads without a division
// per element: (r, c) advance by (256 / cols, 256 % cols) with one carry.
struct Walk2D {
    int r, c, dr, dc, cols;
    __device__ __forceinline__ Walk2D(int cols_) : cols(cols_)
    {
        r = (int)threadIdx.x / cols;
        c = (int)threadIdx.x - r * cols;
        dr = 256 / cols;
        dc = 256 - dr * cols;
    }
    __device__ __forceinline__ void next()
    {
        r += dr;
        c += dc;
        if (c >= cols) {
            c -= cols;
            r++;
        }
    }
};

template <bool IS_MAX>
__device__ __forceinline__ void chain_op(uint32_t *A, uint32_t *B, int inWd, int inH, int c_lo,
                                         int c_hi, int my, int rx, int ry, int X0, int Y0, int H,
                                         int W, uint32_t next_ident)
{
    const uint32_t ident = IS_MAX ? 0u : 0x00FF00FFu;   // per 16-bit lane
    const int cols = c_hi - c_lo;
    // ---- row pass A -> B: rows [my, inH-my), dwords [c_lo, c_hi) ----
    const int rows = inH - 2 * my;
    for (Walk2D wk(cols); wk.r < rows; wk.next()) {
        const int r = my + wk.r, cd = c_lo + wk.c;
        const uint32_t *p = A + r * inWd + cd;
        uint32_t ae = ident, ao = ident;
#pragma unroll
        for (int q = -CMAXQ; q <= CMAXQ; q++) {
            if (4 * q + 3 < -rx || 4 * q > rx)   // wave-uniform
                continue;
            const uint32_t lo = p[q];
            const uint32_t hi = (4 * q + 1 <= rx && q < CMAXQ) ? p[q + 1] : 0u;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int sft = 4 * q + b;
                if (sft < -rx || sft > rx)
                    continue;
                acc_u8x4<IS_MAX>(ae, ao, b == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, b));
            }
        }
        B[r * inWd + cd] = ae | (ao << 8);
    }
    __syncthreads();
    // ---- column pass B -> A: rows [my+ry, inH-my-ry) ----
    const int rows2 = inH - 2 * (my + ry);
    for (Walk2D wk(cols); wk.r < rows2; wk.next()) {
        const int r = my + ry + wk.r, cd = c_lo + wk.c;
        const uint32_t *p = B + r * inWd + cd;
        uint32_t ae = ident, ao = ident;
#pragma unroll
        for (int d = -CMAXR; d <= CMAXR; d++) {
            if (d < -ry || d > ry)
                continue;
            acc_u8x4<IS_MAX>(ae, ao, p[d * inWd]);
        }
        uint32_t v = ae | (ao << 8);
        const uint32_t m = in_image_mask(Y0 + r, X0 + 4 * cd, H, W);
        A[r * inWd + cd] = (v & m) | (next_ident & ~m);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void morph_chain_kernel(const uint8_t *__restrict__ src,
                                                          uint8_t *__restrict__ dst, int H, int W,
                                                          RectChain ch, int HQ, int RY, int fast)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int inWd = CTWD + 2 * HQ, inH = CTH + 2 * RY;
    uint32_t *A = (uint32_t *)smem;
    uint32_t *B = A + inWd * inH;
    const int X0 = blockIdx.x * (4 * CTWD) - 4 * HQ, Y0 = blockIdx.y * CTH - RY;
    const size_t N = (size_t)H * W;
    const uint8_t *simg = src + (size_t)blockIdx.z * N;
    uint8_t *dimg = dst + (size_t)blockIdx.z * N;

    const uint32_t ident0 = ch.op[0] ? 0u : 0xFFFFFFFFu;
    for (Walk2D wk(inWd); wk.r < inH; wk.next()) {
        const int r = wk.r, cd = wk.c, idx = r * inWd + cd;
        const int gy = Y0 + r, gx0 = X0 + 4 * cd;
        const uint32_t m = in_image_mask(gy, gx0, H, W);
        uint32_t v = 0;
        if (m) {
            const uint8_t *p = simg + (size_t)gy * W + gx0;
            if (fast) {
                v = *(const uint32_t *)p;
            } else {
                const int nvalid = W - gx0 < 4 ? W - gx0 : 4;
                for (int b = 0; b < nvalid; b++)
                    v |= (uint32_t)p[b] << (8 * b);
            }
        }
        A[idx] = (v & m) | (ident0 & ~m);
    }
    __syncthreads();

    int c_lo = 0, my = 0;
    for (int o = 0; o < ch.n; o++) {
        const int rx = ch.rx[o], ry = ch.ry[o];
        c_lo += (rx + 3) >> 2;
        const uint32_t next_ident = (o + 1 < ch.n && ch.op[o + 1] == 0) ? 0xFFFFFFFFu : 0u;
        if (ch.op[o])
            chain_op<true>(A, B, inWd, inH, c_lo, inWd - c_lo, my, rx, ry, X0, Y0, H, W, next_ident);
        else
            chain_op<false>(A, B, inWd, inH, c_lo, inWd - c_lo, my, rx, ry, X0, Y0, H, W, next_ident);
        my += ry;
    }

    for (int idx = threadIdx.x; idx < CTH * CTWD; idx += 256) {
        const int r = idx / CTWD, cd = idx - r * CTWD;
        const int gy = Y0 + RY + r, gx0 = X0 + 4 * (HQ + cd);
        if (gy >= H || gx0 >= W)
            continue;
        const uint32_t v = A[(r + RY) * inWd + HQ + cd];
        uint8_t *p = dimg + (size_t)gy * W + gx0;
        if (fast) {
            *(uint32_t *)p = v;
        } else {
            const int nvalid = W - gx0 < 4 ? W - gx0 : 4;
            for (int b = 0; b < nvalid; b++)
                p[b] = (uint8_t)(v >> (8 * b));
        }
    }
}

// ---------------------------------------------------------------------------
// register-streaming variant of the fused chain for the common small kernels
// (every op rectangular with rx <= 4, ry <= 4: 3x3 ... 9x9).
//
// A wave owns a strip of 64 dwords = 256 px; lane l keeps its 4-px column of NR
// consecutive rows in registers, split into even bytes E = (px0, px2) and odd
// bytes O = (px1, px3) as two u16 pairs, so that a byte-wise min/max is one
// v_pk_{min,max}_u16 per register with no unpacking inside the loops.
//   row pass    : the neighbours' registers come over DPP wave shifts (no LDS);
//                 a window shifted by s px is at most one v_alignbit per register.
//   column pass : out[j] = op(in[j-ry .. j+ry]) over the register array, written
//                 back at index j-ry (dead by then), so the whole chain runs in
//                 place; RY is a template parameter to keep every index static.
// Each op shrinks the valid region by one lane per side and ry rows per side;
// out-of-image bytes are forced to the next op's identity after every op.
// ---------------------------------------------------------------------------
template <bool IS_MAX>
__device__ __forceinline__ uint32_t pkmm(uint32_t a, uint32_t b)
{
    return pk16<IS_MAX>(a, b);
}

__device__ __forceinline__ uint32_t lane_prev(uint32_t v)   // value of lane-1 (lane 0: unchanged)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_next(uint32_t v)   // value of lane+1 (lane 63: unchanged)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /*wave_shl:1*/, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t al16(uint32_t hi, uint32_t lo)   // (lo.hi16, hi.lo16)
{
    return __builtin_amdgcn_alignbit(hi, lo, 16);
}

template <int NR, bool IS_MAX, int RY>
__device__ __forceinline__ void stream_op(uint32_t (&E)[NR], uint32_t (&O)[NR], int rx)
{
    // Straight-line over all NR register rows: rows past the still-valid range
    // hold don't-care values (never stored), which is cheaper than guarding each
    // row with a branch.
    // ---- row pass ----
#pragma unroll
    for (int j = 0; j < NR; j++) {
        {
            const uint32_t e = E[j], o = O[j];
            const uint32_t ep = lane_prev(e), en = lane_next(e), op = lane_prev(o), on = lane_next(o);
            const uint32_t a1 = al16(en, e), a2 = al16(on, o), a3 = al16(e, ep), a4 = al16(o, op);
            uint32_t re = e, ro = o;
            if (rx >= 1) {   // +1: (o, a1)   -1: (a4, e)
                re = pkmm<IS_MAX>(pkmm<IS_MAX>(re, o), a4);
                ro = pkmm<IS_MAX>(pkmm<IS_MAX>(ro, a1), e);
            }
            if (rx >= 2) {   // +2: (a1, a2)  -2: (a3, a4)
                re = pkmm<IS_MAX>(pkmm<IS_MAX>(re, a1), a3);
                ro = pkmm<IS_MAX>(pkmm<IS_MAX>(ro, a2), a4);
            }
            if (rx >= 3) {   // +3: (a2, en)  -3: (op, a3)
                re = pkmm<IS_MAX>(pkmm<IS_MAX>(re, a2), op);
                ro = pkmm<IS_MAX>(pkmm<IS_MAX>(ro, en), a3);
            }
            if (rx >= 4) {   // +4: (en, on)  -4: (ep, op)
                re = pkmm<IS_MAX>(pkmm<IS_MAX>(re, en), ep);
                ro = pkmm<IS_MAX>(pkmm<IS_MAX>(ro, on), op);
            }
            E[j] = re;
            O[j] = ro;
        }
    }
    // ---- column pass: result row j (centre) is stored at index j - RY ----
#pragma unroll
    for (int j = RY; j < NR - RY; j++) {
        {
            uint32_t re = E[j - RY], ro = O[j - RY];
#pragma unroll
            for (int d = -RY + 1; d <= RY; d++) {
                re = pkmm<IS_MAX>(re, E[j + d]);
                ro = pkmm<IS_MAX>(ro, O[j + d]);
            }
            E[j - RY] = re;
            O[j - RY] = ro;
        }
    }
}

// Bitwise form of the same op for masks whose values form a chain under bitwise
// inclusion, a <= b  <=>  (a & b) == a.  Then min == AND and max == OR on whole
// dwords: 4 px per instruction, no even/odd split.  Every mask a detector of
// this library produces is such a chain: MOG2 {0, shadow_value, 255} (0 and 255
// are below / above every byte), FD {0, 255}, {0, 84, 85} (BGR) or {0, 63} (4 ch).
// RXT >= 0: the horizontal radius is a compile-time constant (square kernels, the
// common case) so that only the taps that exist are emitted; RXT = -1 takes it at
// run time (hipcc then predicates all four tap levels instead of branching).
template <int NR, bool IS_MAX, int RY, int RXT>
__device__ __forceinline__ void stream_op_bits(uint32_t (&R)[NR], int rx_rt)
{
    const int rx = RXT >= 0 ? RXT : rx_rt;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const uint32_t r = R[j];
        const uint32_t p = lane_prev(r), n = lane_next(r);
        uint32_t acc = r;
#define GOF_BW(x) acc = IS_MAX ? (acc | (x)) : (acc & (x))
        if (rx >= 1) {
            GOF_BW(__builtin_amdgcn_alignbyte(n, r, 1));
            GOF_BW(__builtin_amdgcn_alignbyte(r, p, 3));
        }
        if (rx >= 2) {
            GOF_BW(__builtin_amdgcn_alignbyte(n, r, 2));
            GOF_BW(__builtin_amdgcn_alignbyte(r, p, 2));
        }
        if (rx >= 3) {
            GOF_BW(__builtin_amdgcn_alignbyte(n, r, 3));
            GOF_BW(__builtin_amdgcn_alignbyte(r, p, 1));
        }
        if (rx >= 4) {
            GOF_BW(n);
            GOF_BW(p);
        }
        R[j] = acc;
    }
#pragma unroll
    for (int j = RY; j < NR - RY; j++) {
        uint32_t acc = R[j - RY];
#pragma unroll
        for (int d = -RY + 1; d <= RY; d++)
            GOF_BW(R[j + d]);
        R[j - RY] = acc;
    }
#undef GOF_BW
}

template <int NR>
__global__ __launch_bounds__(64) void morph_stream_bits_kernel(const uint8_t *__restrict__ src,
                                                                uint8_t *__restrict__ dst, int H,
                                                                int W, RectChain ch, int HQ, int RY,
                                                                const uint8_t *__restrict__ rowflags,
                                                                int streams)
{
    // dword-aligned images only (the launcher checks); same geometry as morph_stream_kernel
    const int lane = threadIdx.x;
    const int rows_out = NR - 2 * RY;
    const int strip_px = (64 - 2 * HQ) * 4;
    const int sx0 = blockIdx.x * strip_px;
    const int y0 = blockIdx.y * rows_out;
    if (sx0 >= W || y0 >= H)
        return;
    const size_t N = (size_t)H * W;
    const uint8_t *simg = src + (size_t)blockIdx.z * N;
    uint8_t *dimg = dst + (size_t)blockIdx.z * N;
    const int gx0 = sx0 - 4 * HQ + 4 * lane;
    const int Y0 = y0 - RY;
    const uint32_t cm = (gx0 >= 0 && gx0 < W) ? 0xFFFFFFFFu : 0u;   // W % 4 == 0: all or nothing
    // interior strips never see the border: skip the identity re-imposition there
    const bool interior = __builtin_amdgcn_readfirstlane(
        (int)(sx0 - 4 * HQ >= 0 && sx0 - 4 * HQ + 256 <= W && Y0 >= 0 && Y0 + NR <= H));

    // The detector said which 64-px row segments of its masks are all zero.  If
    // every segment this strip reads (halo included) is, every output is zero too:
    // a min / max over zeros and the ops' out-of-image identities (255 only under a
    // min) is zero.  Then nothing is read -- most strips of a surveillance mask.
    if (rowflags != nullptr) {
        const int n = blockIdx.z, t = n / streams, st = n - t * streams;
        const int pitch = (W + 63) >> 6;
        const uint8_t *fp = rowflags + ((size_t)((t >> 3) * streams + st) * H) * pitch;
        const int xa = sx0 - 4 * HQ < 0 ? 0 : sx0 - 4 * HQ;
        const int xb = sx0 - 4 * HQ + 255 > W - 1 ? W - 1 : sx0 - 4 * HQ + 255;
        const int gy = Y0 + (lane < NR ? lane : 0);
        const int gyc = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
        uint32_t v = 0;
        for (int cx = xa >> 6; cx <= (xb >> 6); cx++)   // at most 5 tile columns
            v |= fp[(size_t)gyc * pitch + cx];
        if (__builtin_amdgcn_ballot_w64(((v >> (t & 7)) & 1u) != 0) == 0) {
            if (lane >= HQ && lane < 64 - HQ && gx0 < W) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const int oy = y0 + j;
                    if (j < rows_out && oy < H)
                        *(uint32_t *)(dimg + (size_t)oy * W + gx0) = 0u;
                }
            }
            return;
        }
    }

    uint32_t R[NR];
    const int gxc = gx0 < 0 ? 0 : (gx0 > W - 4 ? W - 4 : gx0);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int gy = Y0 + j;
        const int gyc = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
        R[j] = *(const uint32_t *)(simg + (size_t)gyc * W + gxc);
    }
    if (!interior) {
        const uint32_t id0 = ch.op[0] ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = Y0 + j;
            const uint32_t m = (gy >= 0 && gy < H) ? cm : 0u;
            R[j] = (R[j] & m) | (id0 & ~m);
        }
    }

    int ycum = 0;
    for (int o = 0; o < ch.n; o++) {
        const int rx = ch.rx[o], ry = ch.ry[o];
        const bool mx = ch.op[o] != 0;
#define GOF_STREAM_OP(RR, XX)                              \
    if (mx)                                                \
        stream_op_bits<NR, true, RR, XX>(R, rx);           \
    else                                                   \
        stream_op_bits<NR, false, RR, XX>(R, rx);
        if (rx == ry) {   // square kernels: both radii static
            switch (ry) {
            case 0: GOF_STREAM_OP(0, 0) break;
            case 1: GOF_STREAM_OP(1, 1) break;
            case 2: GOF_STREAM_OP(2, 2) break;
            case 3: GOF_STREAM_OP(3, 3) break;
            default: GOF_STREAM_OP(4, 4) break;
            }
        } else {
            switch (ry) {
            case 0: GOF_STREAM_OP(0, -1) break;
            case 1: GOF_STREAM_OP(1, -1) break;
            case 2: GOF_STREAM_OP(2, -1) break;
            case 3: GOF_STREAM_OP(3, -1) break;
            default: GOF_STREAM_OP(4, -1) break;
            }
        }
#undef GOF_STREAM_OP
        ycum += ry;
        if (o + 1 < ch.n && !interior) {
            const uint32_t idn = ch.op[o + 1] ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int gy = Y0 + ycum + j;
                const uint32_t m = (gy >= 0 && gy < H) ? cm : 0u;
                R[j] = (R[j] & m) | (idn & ~m);
            }
        }
    }

    if (lane >= HQ && lane < 64 - HQ && gx0 < W) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = y0 + j;
            if (j < rows_out && gy < H)
                *(uint32_t *)(dimg + (size_t)gy * W + gx0) = R[j];
        }
    }
}

template <int NR>
__global__ __launch_bounds__(64) void morph_stream_kernel(const uint8_t *__restrict__ src,
                                                           uint8_t *__restrict__ dst, int H, int W,
                                                           RectChain ch, int HQ, int RY, int fast)
{
    // one wave per workgroup: the waves are independent, and single-wave
    // workgroups let the dispatcher balance the (few thousand) strips over the SIMDs
    const int lane = threadIdx.x;
    const int rows_out = NR - 2 * RY;
    const int strip_px = (64 - 2 * HQ) * 4;
    const int sx0 = blockIdx.x * strip_px;   // first output px of this wave's strip
    const int y0 = blockIdx.y * rows_out;
    if (sx0 >= W || y0 >= H)
        return;   // whole wave; there is no barrier in this kernel
    const size_t N = (size_t)H * W;
    const uint8_t *simg = src + (size_t)blockIdx.z * N;
    uint8_t *dimg = dst + (size_t)blockIdx.z * N;
    const int gx0 = sx0 - 4 * HQ + 4 * lane;   // image x of this lane's first px (multiple of 4)
    const int Y0 = y0 - RY;

    // in-image byte mask of this lane's column, as seen by the packed halves
    uint32_t cm = 0;
    if (gx0 >= 0 && gx0 < W)
        cm = (W - gx0) >= 4 ? 0xFFFFFFFFu : ((1u << (8 * (W - gx0))) - 1u);
    const uint32_t cmE = cm & 0x00FF00FFu, cmO = (cm >> 8) & 0x00FF00FFu;

    uint32_t E[NR], O[NR];
    const uint32_t id0 = ch.op[0] ? 0u : 0x00FF00FFu;
    if (fast) {
        // branch-free: clamped (always valid) addresses, so that all NR row loads
        // are in flight together; out-of-image data is masked to the identity after
        const int gxc = gx0 < 0 ? 0 : (gx0 > W - 4 ? W - 4 : gx0);
        uint32_t raw[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = Y0 + j;
            const int gyc = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
            raw[j] = *(const uint32_t *)(simg + (size_t)gyc * W + gxc);
        }
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = Y0 + j;
            const bool rowin = gy >= 0 && gy < H;
            const uint32_t mE = rowin ? cmE : 0u, mO = rowin ? cmO : 0u;
            E[j] = ((raw[j] & 0x00FF00FFu) & mE) | (id0 & ~mE);
            O[j] = (((raw[j] >> 8) & 0x00FF00FFu) & mO) | (id0 & ~mO);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = Y0 + j;
            uint32_t v = 0;
            const bool rowin = gy >= 0 && gy < H;
            if (rowin && cm) {
                const uint8_t *p = simg + (size_t)gy * W + gx0;
                const int nvalid = W - gx0 < 4 ? W - gx0 : 4;
                for (int b = 0; b < nvalid; b++)
                    v |= (uint32_t)p[b] << (8 * b);
            }
            const uint32_t mE = rowin ? cmE : 0u, mO = rowin ? cmO : 0u;
            E[j] = ((v & 0x00FF00FFu) & mE) | (id0 & ~mE);
            O[j] = (((v >> 8) & 0x00FF00FFu) & mO) | (id0 & ~mO);
        }
    }

    int ycum = 0;
    for (int o = 0; o < ch.n; o++) {
        const int rx = ch.rx[o], ry = ch.ry[o];
        const bool mx = ch.op[o] != 0;
#define GOF_STREAM_OP(RR)                                         \
    if (mx)                                                       \
        stream_op<NR, true, RR>(E, O, rx);                        \
    else                                                          \
        stream_op<NR, false, RR>(E, O, rx);
        switch (ry) {
        case 0: GOF_STREAM_OP(0) break;
        case 1: GOF_STREAM_OP(1) break;
        case 2: GOF_STREAM_OP(2) break;
        case 3: GOF_STREAM_OP(3) break;
        default: GOF_STREAM_OP(4) break;
        }
#undef GOF_STREAM_OP
        ycum += ry;
        if (o + 1 < ch.n) {
            // what cv2's constant border does to the intermediate image
            const uint32_t idn = ch.op[o + 1] ? 0u : 0x00FF00FFu;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int gy = Y0 + ycum + j;
                const bool rowin = gy >= 0 && gy < H;
                const uint32_t mE = rowin ? cmE : 0u, mO = rowin ? cmO : 0u;
                E[j] = (E[j] & mE) | (idn & ~mE);
                O[j] = (O[j] & mO) | (idn & ~mO);
            }
        }
    }

    if (lane >= HQ && lane < 64 - HQ && gx0 < W) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = y0 + j;
            if (j < rows_out && gy < H) {
                const uint32_t v = E[j] | (O[j] << 8);
                uint8_t *p = dimg + (size_t)gy * W + gx0;
                if (fast) {
                    *(uint32_t *)p = v;
                } else {
                    const int nvalid = W - gx0 < 4 ? W - gx0 : 4;
                    for (int b = 0; b < nvalid; b++)
                        p[b] = (uint8_t)(v >> (8 * b));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// step 0 (optional): cv.cvtColor(X, COLOR_BGR2GRAY | COLOR_RGB2GRAY), pipeline.py:458-459
// 8-bit fixed point of the OpenCV 3.4 line the reference pins [OPENCV-RECALL]:
//   Y = (B*1868 + G*9617 + R*4899 + 2^13) >> 14
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t *__restrict__ src,
                                                   uint8_t *__restrict__ dst, size_t npix, int c,
                                                   int c0, int c1, int c2)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix)
        return;
    const uint8_t *p = src + i * c;
    dst[i] = (uint8_t)((p[0] * c0 + p[1] * c1 + p[2] * c2 + (1 << 13)) >> 14);
}

// four pixels per thread: CS aligned dwords in, one dword out (GrayQuad: the same integers)
template <int CS>
__global__ __launch_bounds__(256) void gray_quad_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                        size_t nquads, int g0, int g1, int g2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nquads)
        return;
    GrayQuad<CS> gq;
    gq.build(g0, g1, g2);
    uint32_t d[CS];
#pragma unroll
    for (int q = 0; q < CS; q++)
        d[q] = src[i * CS + q];
    dst[i] = gq.convert(d);
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

bool fd_frames_supported(const uint8_t *frame, int n_img, int h, int w, int c, const Taps &t, int src_c)
{
    // byte taps whose folded outer pair fits a byte, at least two columns, and every
    // staged row of every frame at the same offset mod 4 (see mog2_apply).  src_c != 0: the
    // frames arrive with src_c (3 | 4) channels and are converted to grey (c == 1) on the staged tiles
    if (src_c != 0 && (c != 1 || (src_c != 3 && src_c != 4)))
        return false;
    const int cc = src_c ? src_c : c;
    return t.fused3_ok && t.vky[0] == t.vky[2] && t.kx[0] + t.kx[2] <= 255 && w > 1 && ((size_t)w * cc) % 4 == 0 &&
           ((size_t)h * w * cc) % 4 == 0 && ((size_t)n_img * h * w * cc) % 4 == 0 &&
           (uintptr_t)frame % 4 == 0 && (c == 1 || c == 3 || c == 4);
}

hipError_t fd_frames_apply(const uint8_t *frame, uint8_t *mask, const uint32_t *prev_in,
                           uint32_t *prev_out, bool has_prev, int ithr, int n_img, int h, int w, int c,
                           int T, const Taps &t, hipStream_t s, uint8_t *rowflags, int src_c, bool src_rgb)
{
    if (n_img < 1 || h < 1 || (h + TH - 1) / TH > 65535 || n_img > 65535 || T < 1 || T > kMaxT ||
        !fd_frames_supported(frame, n_img, h, w, c, t, src_c) || !prev_out || (has_prev && !prev_in))
        return hipErrorInvalidValue;
    FdFramesArgs a;
    a.frame = frame;
    a.mask = mask;
    a.prev_in = prev_in;
    a.prev_out = prev_out;
    a.rowflags = rowflags;
    a.frame_stride = (size_t)n_img * h * w * (src_c ? src_c : c);
    a.mask_stride = (size_t)n_img * h * w;
    a.H = h;
    a.W = w;
    a.T = T;
    a.has_prev = has_prev ? 1 : 0;
    a.ithr = ithr;
    gray_weights(src_rgb, a.g0, a.g1, a.g2);
    a.flag_pitch = rowflag_pitch(w);
    a.kxw = (uint32_t)t.kx[0] | ((uint32_t)t.kx[1] << 8) | ((uint32_t)t.kx[2] << 16);
    a.vky0 = t.vky[0];
    a.vky1 = t.vky[1];
    a.vround = t.vround;
    dim3 g = grid3(w, h, n_img, TW, TH);
#define GOF_LAUNCH_FD(CC)                                                                   \
    do {                                                                                    \
        if (T == 1)                                                                         \
            hipLaunchKernelGGL((fd_frames_kernel<CC, 1>), g, dim3(256), 0, s, a);           \
        else                                                                                \
            hipLaunchKernelGGL((fd_frames_kernel<CC, kMaxT>), g, dim3(256), 0, s, a);       \
    } while (0)
    if (src_c) {
        if (T == 1 && src_c == 3)
            hipLaunchKernelGGL((fd_frames_kernel<1, 1, 3>), g, dim3(256), 0, s, a);
        else if (T == 1)
            hipLaunchKernelGGL((fd_frames_kernel<1, 1, 4>), g, dim3(256), 0, s, a);
        else if (src_c == 3)
            hipLaunchKernelGGL((fd_frames_kernel<1, kMaxT, 3>), g, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((fd_frames_kernel<1, kMaxT, 4>), g, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    switch (c) {
    case 1: GOF_LAUNCH_FD(1); break;
    case 3: GOF_LAUNCH_FD(3); break;
    case 4: GOF_LAUNCH_FD(4); break;
    default: return hipErrorInvalidValue;
    }
#undef GOF_LAUNCH_FD
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

namespace {
// halo of the packed chain kernel: dwords per side, rows per side, LDS bytes
bool chain_geometry(const RectChain &ch, int *HQ, int *RY, size_t *lds)
{
    if (ch.n < 1 || ch.n > 8)
        return false;
    int hq = 0, ry = 0;
    for (int i = 0; i < ch.n; i++) {
        if (ch.rx[i] < 0 || ch.ry[i] < 0 || ch.rx[i] > 4 * CMAXQ - 1 || ch.ry[i] > CMAXR ||
            (ch.op[i] != 0 && ch.op[i] != 1))
            return false;
        hq += (ch.rx[i] + 3) >> 2;
        ry += ch.ry[i];
    }
    *HQ = hq;
    *RY = ry;
    *lds = 2 * sizeof(uint32_t) * (size_t)(CTWD + 2 * hq) * (size_t)(CTH + 2 * ry);
    return *lds <= 64 * 1024;
}
}  // namespace

bool morph_rect_chain_supported(const RectChain &ch)
{
    int HQ, RY;
    size_t lds;
    return chain_geometry(ch, &HQ, &RY, &lds);
}

namespace {
// the register-streaming kernel takes chains of small rectangular ops
bool stream_geometry(const RectChain &ch, int *HQ, int *RY)
{
    if (ch.n < 1 || ch.n > 8)
        return false;
    int hq = 0, ry = 0;
    for (int i = 0; i < ch.n; i++) {
        if (ch.rx[i] < 0 || ch.rx[i] > 4 || ch.ry[i] < 0 || ch.ry[i] > 4 ||
            (ch.op[i] != 0 && ch.op[i] != 1))
            return false;
        hq += ch.rx[i] > 0 ? 1 : 0;
        ry += ch.ry[i];
    }
    *HQ = hq;
    *RY = ry;
    return ry <= 16 && hq <= 8;
}
}  // namespace

hipError_t morph_rect_chain(const uint8_t *src, uint8_t *dst, int n_img, int h, int w,
                            const RectChain &ch, bool chain_valued, hipStream_t s,
                            const uint8_t *rowflags, int streams)
{
    int HQ, RY;
    size_t lds;
    if (n_img < 1 || h < 1 || w < 1 || n_img > 65535)
        return hipErrorInvalidValue;
    // dword loads/stores need every row (and every image) to start 4-byte aligned
    const int fast = (w % 4 == 0) && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0);
    static const bool no_stream = getenv("GOF_MORPH_LDS") != nullptr;   // A/B knob
    if (!no_stream && stream_geometry(ch, &HQ, &RY)) {
        const int NR = RY <= 8 ? 32 : 48;   // register rows per strip (32 measured best for RY = 4)
        const int rows_out = NR - 2 * RY, strip_px = (64 - 2 * HQ) * 4;
        if ((h + rows_out - 1) / rows_out > 65535)
            return hipErrorInvalidValue;
        dim3 g((unsigned)((w + strip_px - 1) / strip_px), (unsigned)((h + rows_out - 1) / rows_out),
               (unsigned)n_img);
        static const bool no_bits = getenv("GOF_MORPH_NOBITS") != nullptr;   // A/B knob
        static const bool no_skip = getenv("GOF_MORPH_NOSKIP") != nullptr;   // A/B knob
        const uint8_t *rf = (no_skip || streams < 1 || n_img % (streams < 1 ? 1 : streams)) ? nullptr : rowflags;
        if (chain_valued && fast && !no_bits) {
            switch (NR) {
            case 16: hipLaunchKernelGGL(morph_stream_bits_kernel<16>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, rf, streams); break;
            case 24: hipLaunchKernelGGL(morph_stream_bits_kernel<24>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, rf, streams); break;
            case 32: hipLaunchKernelGGL(morph_stream_bits_kernel<32>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, rf, streams); break;
            case 48: hipLaunchKernelGGL(morph_stream_bits_kernel<48>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, rf, streams); break;
            default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        switch (NR) {
        case 16: hipLaunchKernelGGL(morph_stream_kernel<16>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, fast); break;
        case 24: hipLaunchKernelGGL(morph_stream_kernel<24>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, fast); break;
        case 32: hipLaunchKernelGGL(morph_stream_kernel<32>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, fast); break;
        case 48: hipLaunchKernelGGL(morph_stream_kernel<48>, g, dim3(64), 0, s, src, dst, h, w, ch, HQ, RY, fast); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (!chain_geometry(ch, &HQ, &RY, &lds) || (h + CTH - 1) / CTH > 65535)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(morph_chain_kernel, grid3(w, h, n_img, 4 * CTWD, CTH), dim3(256), lds, s,
                       src, dst, h, w, ch, HQ, RY, fast);
    return hipGetLastError();
}

hipError_t cvt_gray(const uint8_t *src, uint8_t *dst, int n_img, int h, int w, int c, bool rgb,
                    hipStream_t s)
{
    if (n_img < 1 || h < 1 || w < 1 || (c != 3 && c != 4))
        return hipErrorInvalidValue;
    size_t npix = (size_t)n_img * h * w;
    if ((npix + 255) / 256 > 0x7fffffffu)
        return hipErrorInvalidValue;
    int g0, g1, g2;
    gray_weights(rgb, g0, g1, g2);
    // aligned buffers: four pixels per thread on dwords; the last npix % 4 pixels (and everything else) bytewise
    size_t done = 0;
    if ((uintptr_t)src % 4 == 0 && (uintptr_t)dst % 4 == 0 && npix >= 4) {
        const size_t nquads = npix / 4;
        const dim3 gq((unsigned)((nquads + 255) / 256));
        if (c == 3)
            hipLaunchKernelGGL(gray_quad_kernel<3>, gq, dim3(256), 0, s, (const uint32_t *)src, (uint32_t *)dst, nquads, g0, g1, g2);
        else
            hipLaunchKernelGGL(gray_quad_kernel<4>, gq, dim3(256), 0, s, (const uint32_t *)src, (uint32_t *)dst, nquads, g0, g1, g2);
        done = nquads * 4;
    }
    if (done < npix)
        hipLaunchKernelGGL(gray_kernel, dim3((unsigned)((npix - done + 255) / 256)), dim3(256), 0, s, src + done * c,
                           dst + done, npix - done, c, g0, g1, g2);
    return hipGetLastError();
}


}  // namespace gofk
