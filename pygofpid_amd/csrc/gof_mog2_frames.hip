// mog2_frames_kernel: blur 3x3 + MOG2 for up to 8 consecutive frames of every stream per
// launch -- the kernel bench.py and ForegroundPipeline.apply_many run.  Its own translation
// unit (compiles in seconds); the update functions live in gof_mog2_dev.h.
// Compiled with -ffp-contract=off (see gof_mog2_dev.h / Makefile).
#include "gof_mog2_dev.h"

namespace gofk {

namespace {

// ---------------------------------------------------------------------------
// mog2_frames_kernel: the multi-frame launch on aligned frames (what bench.py and
// ForegroundPipeline.apply_many run).  Same tile, same staging idea and the same
// update functions as mog2_kernel<C, true, true, true>, re-written around what the
// counters say bounds that launch: the number of vector AND scalar instructions a
// wave issues (profiles/ DESIGN.md 5.2), not HBM.
//   * the frame rows of all T frames are staged from two running row pointers
//     (every row of every frame has the same offset mod 4, which the launcher
//     checks): a handful of scalar instructions per frame instead of a 64-bit
//     address computation per row and frame;
//   * the per-frame rates come as one 32-byte record from the kernel-argument
//     segment (one scalar load per frame);
//   * lanes outside the image shadow a pixel inside it (same model, same
//     window) and only their stores are masked, so the frame loop runs without
//     exec-mask bookkeeping;
//   * settled waves (every pixel one mode of weight exactly 1.0f) take the
//     host-evaluated weight arithmetic (FrameRates) instead of two division
//     chains per lane.
// ---------------------------------------------------------------------------
#ifdef GOF_PATH_STATS
__device__ unsigned long long g_path_stats[32];
#endif

struct Mog2FramesArgs {
    const uint8_t *frame;
    uint8_t *mask;
    float *planes;
    uint8_t *modes;
    size_t np, frame_stride, mask_stride;
    int H, W, T;
    uint32_t kxw;     // horizontal taps as bytes (k0 | k1 << 8 | k2 << 16)
    uint32_t vky0, vky1, vround;   // vertical multipliers (rows 0 / 2 share one) and rounding constant (Taps::vky)
    uint8_t *rowflags;   // [image][row][tile column] bytes, bit t = frame t's mask not all zero there (or NULL)
    int flag_pitch;
    Mog2Consts k;
    uint32_t taps[kTapTableDwords];   // tap_table(kxw, C): the lanes' blur patterns
    int g0, g1, g2;                   // CS != 0: luma weights of source channels 0 / 1 / 2 (cvt_gray's, 14-bit)
};

// (7 waves per SIMD for BGR: the issue-bound launch needs every one of them; the float4
// model loads alone would let the allocator drift to 74 registers = 6 waves, 8 % slower)
#ifndef GOF_FRAMES_WAVES
#define GOF_FRAMES_WAVES 7   // (A/B knob of profiles/: waves per SIMD the register allocation aims at)
#endif
// CS: channels of the frames as they arrive when step 0 (cv.cvtColor to grey, pipeline.py:458-459)
// is fused into the staging -- the colour rows are staged raw, converted to a grey tile in LDS
// (cvt_gray's integer, one pass of the workgroup), and the kernel proceeds as the one-channel
// kernel on that tile: no grey image ever exists in HBM.  CS = 0: frames arrive with C channels.
// NT: frames staged per launch (kMaxT, or 1 for the per-frame call of a converting pipeline).
constexpr int kGrayRowD = 20;   // dwords per grey tile row: 66 px + the 12-byte window's reach
template <int C, int CS = 0, int NT = kMaxT>
__global__ __launch_bounds__(256, (C <= 3 && CS != 4) ? GOF_FRAMES_WAVES : 1) void mog2_frames_kernel(const Mog2FramesArgs a)
{
    constexpr int F = 2 + C;
    static_assert(CS == 0 || C == 1, "the conversion produces one channel");
    typedef FrameTiles<(CS ? CS : C), NT> Tiles;
    constexpr int ROWD = CS ? kGrayRowD : Tiles::ROWD, TILE = (TH + 2) * ROWD;
    __shared__ uint32_t tiles[NT * TILE];
    __shared__ uint32_t ctiles[CS ? Tiles::LDS_DWORDS : 1];   // the raw colour tiles (CS != 0)

    const int H = a.H, W = a.W, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (An XCD-aware workgroup -> tile mapping -- runs of q tiles per XCD, q = 2 .. 480, or one
    // contiguous eighth of the tiles each -- was measured again in round 3 on this kernel: 0 to
    // +3 % time for every q although it removes the re-fetch of halo rows and shared lines between
    // tiles; profiles/r03_xcd_mapping.txt.  The plain grid stays.)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int x0 = bx * TW, y0 = by * TH;
    const int img = bz;
    const int x = x0 + lane, y = y0 + wave;
    const size_t N = (size_t)H * W;

    // ---- stage the tiles of all T frames (loads now, LDS writes after the model
    //      loads are in flight); borders are resolved at read time ----
    Tiles ft;
    uint32_t raw[NT][2][Tiles::NDW];
    ft.load(a.frame, a.frame_stride, H, W, T, img, x0, y0, wave, lane, raw);

    // ---- model: mode count + mode 0 speculatively, the rest only if present ----
    const bool valid = y < H && x < W;
    const int yy = y < H ? y : 0;
    const unsigned xl = valid ? (unsigned)x : 0u;
    const size_t rowoff = (size_t)img * N + (size_t)yy * W;   // wave-uniform
    float *const planes_ = a.planes;
    float *const prow = a.planes + rowoff;
    const size_t np = a.np;
    PixelModel<C> M;
    uint8_t *const mrow_modes = a.modes + rowoff;     // (wave-uniform row bases + the lane's 32-bit column)
    const int mbyte_in = (int)row_load<uint8_t>(mrow_modes, xl);         // mode count | kW0One (see mog2_kernel)
    int nmodes = mbyte_in & kModeCount;
    load_mode_vec_row<C>(planes_, np, rowoff, xl, M.var[0], M.mu[0]);
    M.w[0] = 1.f;
    if (__builtin_amdgcn_ballot_w64(!(mbyte_in & kW0One)) != 0)
        M.w[0] = row_load<float>(prow, xl * 4u);
    const uint32_t w0_in = __float_as_uint(M.w[0]);
    // lane-invariant tap patterns (three loads from the launcher's table, issued with the model
    // loads); a lane outside the image shadows the tile's first pixel
    TapPatterns<C> taps;
    {
        typedef const Mog2FramesArgs __attribute__((address_space(4))) *ArgsPtr0;
        // (a grey tile row starts with the pixel itself: no offset within its first dword)
        taps.load(valid ? x : x0, W, ft.sx0, CS ? 0 : ft.mis0, ((ArgsPtr0)__builtin_amdgcn_kernarg_segment_ptr())->taps);
    }
    // slots 1.. are loaded only if some pixel of the wave has them; otherwise slot 1
    // starts empty (the single-mode loops may fill it) and slots 2.. stay untouched
    // until the general loop is entered
    // the staged rows go to LDS before the further modes are asked for: the 16 staging
    // registers are free again by then (settled waves, four in five, load no further mode)
    Tiles::store(CS ? ctiles : tiles, wave, lane, raw);
    const bool multi_in = __builtin_amdgcn_ballot_w64(nmodes > 1) != 0;
    if (multi_in) {
        // slot m is fetched by the whole wave if any of its pixels has it (the lines travel anyway;
        // what a pixel without the slot gets is dead storage, see mog2_update): no per-lane test,
        // no zero fill
#pragma unroll
        for (int m = 1; m < kMaxModes; m++) {
            if (__builtin_amdgcn_ballot_w64(m < nmodes) != 0) {
                const float *pm = planes_ + (size_t)m * F * np;
                M.w[m] = row_load<float>(pm + rowoff, xl * 4u);
                load_mode_vec_row<C>(pm, np, rowoff, xl, M.var[m], M.mu[m]);
            } else {
                M.w[m] = 0.f;
                M.var[m] = 0.f;
#pragma unroll
                for (int c = 0; c < C; c++)
                    M.mu[m][c] = 0.f;
            }
        }
    } else {
        M.w[1] = 0.f;
        M.var[1] = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++)
            M.mu[1][c] = 0.f;
    }
    // The only barrier, as soon as this wave's tile rows are in LDS.  It orders LDS only (the
    // model loads above stay in flight across it: __syncthreads() would wait for them, and a
    // settled wave would then sit out the second round trip of a neighbour that fetches five modes).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (CS) {
        // step 0 on the staged tiles: every pixel of every staged row of every frame once --
        // (B * g0 + G * g1 + R * g2 + 2^13) >> 14, gray_kernel's statement -- into the grey tile
        const int npx = ((x0 + TW + 1) < W ? (x0 + TW + 1) : W) - ft.sx0;   // staged pixels per row (<= 66)
        const uint8_t *cb = (const uint8_t *)ctiles + ft.mis0;
        uint8_t *gb = (uint8_t *)tiles;
        // a wave takes every fourth staged row (all frames), a lane pixel `lane` of it and, for the
        // first lanes, pixel 64 + lane: no index arithmetic beyond two running offsets
        for (int rr = wave; rr < NT * (TH + 2); rr += 4) {
            const uint8_t *crow = cb + (size_t)rr * Tiles::ROWD * 4;
            uint8_t *grow = gb + (size_t)rr * ROWD * 4;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int px = lane + 64 * half;
                if (px < npx) {
                    const uint8_t *q = crow + px * CS;
                    grow[px] = (uint8_t)((q[0] * a.g0 + q[1] * a.g1 + q[2] * a.g2 + (1 << 13)) >> 14);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    uint32_t wa = (uint32_t)(wave * ROWD) * 4u + (uint32_t)taps.A;   // window address, current frame
    uint32_t half = a.vround;
    float var_max_v = a.k.var_max;   // (the clamp takes one scalar operand, var_min)
    asm volatile("" : "+v"(half), "+v"(var_max_v));   // kept in registers, not re-materialised per use
    const uint32_t ky[3] = {a.vky0, a.vky1, a.vky0};

    // blurred pixel of the current frame as floats (batch: the nine LDS reads issued
    // back to back -- not in the general loop, which has the whole mixture in registers)
    auto frame_input = [&](float (&xin)[C], bool batch) {
        uint32_t v[C];
        taps.template blur<ROWD>(tiles, wa, ky, half, batch, v);
        wa += (uint32_t)TILE * 4u;   // next frame
#pragma unroll
        for (int c = 0; c < C; c++)
            xin[c] = (float)v[c];
    };
    // The frame's scalars: one 32-byte scalar load from the argument segment,
    // requested one frame ahead (the record of frame ti + 1 travels while frame ti
    // is computed; a load placed in the iteration that uses it gets sunk to its
    // first use and its latency lands on the critical path).
    typedef const Mog2FramesArgs __attribute__((address_space(4))) *ArgsPtr;
    const ArgsPtr ap = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    auto load_rates = [&](int ti) -> u32x8 {
        const int tc = ti < kMaxT ? ti : kMaxT - 1;
        return *(const u32x8 __attribute__((address_space(4))) *)&ap->k.fr[tc];
    };
    auto frame_consts = [&](const u32x8 &r) -> Mog2Consts {
        Mog2Consts kt = a.k;
        kt.alphaT = __uint_as_float(r[0]);
        kt.alpha1 = __uint_as_float(r[1]);
        kt.prune = __uint_as_float(r[2]);
        kt.kk1 = __uint_as_float(r[3]);
        kt.wnfit = __uint_as_float(r[4]);
        kt.wnmiss = __uint_as_float(r[5]);
        kt.uni_ok = r[6];
        return kt;
    };
    // this wave's row in the mask of the current frame: a wave-uniform base (scalar registers,
    // advanced per frame on the scalar unit) + the lane's 32-bit offset
    uint8_t *mrow = a.mask + rowoff;

    // Three frame loops, entered in this order and never re-entered: settled waves,
    // then waves whose pixels all have at most one mode, then the general update.
    int ti = 0;
    uint32_t nz = 0;   // bit t: some mask byte of frame t is not zero in this wave (wave-uniform)
#ifdef GOF_PATH_STATS
    PathStats ps;
#pragma unroll
    for (int i = 0; i < 32; i++)
        ps.v[i] = 0;
    ps.v[kStWaves] = 1;
    ps.v[kStWavesMulti] = multi_in ? 1u : 0u;
#endif
    u32x8 rates = load_rates(0);
    // settled on entry?  Afterwards mog2_try_settled reports it frame by frame.
    bool settled = __builtin_amdgcn_ballot_w64(!(nmodes == 1 && M.w[0] == 1.f)) == 0;
    // (a.k.uni_mask: the frames whose rates allow the shortcut, known before the
    // frame's record arrives)
    while (settled && ti < T && ((a.k.uni_mask >> ti) & 1u) != 0) {
        const u32x8 next = load_rates(ti + 1);
        const Mog2Consts kt = frame_consts(rates);
        float xin[C];
        frame_input(xin, true);
        if (!mog2_try_settled<C>(M, xin, kt, var_max_v, settled)) {
            // some pixel missed its mode or is not background: redo the frame below
            wa -= (uint32_t)TILE * 4u;
            break;
        }
        if (valid)
            row_store<uint8_t>(mrow, xl, (uint8_t)0);
        mrow += a.mask_stride;
        rates = next;
        ti++;
        GOF_STAT(kStSettled, 1);
    }
    for (; ti < T; ti++) {
        if (__builtin_amdgcn_ballot_w64(nmodes > 1) != 0)
            break;
        const u32x8 next = load_rates(ti + 1);
        const Mog2Consts kt = frame_consts(rates);
        float xin[C];
        frame_input(xin, true);
        const uint8_t out = mog2_update_single<C, true>(M, nmodes, xin, kt);
        GOF_STAT(kStSingle, 1);
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
        if (valid)
            row_store<uint8_t>(mrow, xl, out);
        mrow += a.mask_stride;
        rates = next;
    }
    if (!multi_in && ti < T) {
#pragma unroll
        for (int m = 2; m < kMaxModes; m++) {
            M.w[m] = 0.f;
            M.var[m] = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++)
                M.mu[m][c] = 0.f;
        }
    }
    for (; ti < T; ti++) {
        const u32x8 next = load_rates(ti + 1);
        const Mog2Consts kt = frame_consts(rates);
        float xin[C];
        frame_input(xin, false);
        // (waves on an object's trail, most frames: every first mode fits)
        uint8_t out = 0;
        if (!mog2_try_first_fits<C, true>(M, nmodes, xin, kt GOF_STAT_PASS)) {
            GOF_STAT(kStGeneral, 1);
            out = mog2_update<C, kMaxModes, true>(M, nmodes, xin, kt GOF_STAT_PASS);
        } else {
            GOF_STAT(kStFirstFits, 1);
        }
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
        if (valid)
            row_store<uint8_t>(mrow, xl, out);
        mrow += a.mask_stride;
        rates = next;
    }
    // one flag byte per wave and launch for the morphology that follows (the settled
    // loop only ever writes zeros, so it contributes no bits)
    if (a.rowflags != nullptr && y < H && lane == 0)
        a.rowflags[((size_t)img * H + y) * a.flag_pitch + bx] = (uint8_t)nz;
#ifdef GOF_PATH_STATS
    if (lane == 0 && y < H) {
#pragma unroll
        for (int i = 0; i < kStCount; i++)
            if (ps.v[i])
                atomicAdd(&g_path_stats[i], (unsigned long long)ps.v[i]);
    }
#endif
    if (!valid)
        return;

    // planes no pixel of the wave changed are not written back (see mog2_kernel)
    const bool w0_dirty = __builtin_amdgcn_ballot_w64(nmodes > 0 && __float_as_uint(M.w[0]) != w0_in) != 0;
    const int mbyte = nmodes | ((nmodes > 0 && __float_as_uint(M.w[0]) == 0x3f800000u) ? kW0One : 0);
    const bool nm_dirty = __builtin_amdgcn_ballot_w64(mbyte != mbyte_in) != 0;
#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m < nmodes) {
            float *pm = planes_ + (size_t)m * F * np;
            if (m > 0 || w0_dirty)
                row_store<float>(pm + rowoff, xl * 4u, M.w[m]);
            store_mode_vec_row<C>(pm, np, rowoff, xl, M.var[m], M.mu[m]);
        }
    }
    if (nm_dirty)
        row_store<uint8_t>(mrow_modes, xl, (uint8_t)mbyte);
}

}  // namespace

// Launch; mog2_apply (gof_mog2.hip) has checked the shapes and that every staged row of every
// frame starts at the same offset mod 4, the taps are bytes whose folded outer pair fits a
// byte, the vertical taps are symmetric, 2 <= T <= kMaxT and w > 1.
hipError_t mog2_frames_launch(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h,
                              int w, int c, int T, const Mog2Consts &k, const Taps &t, hipStream_t s,
                              uint8_t *rowflags, int src_c, bool src_rgb)
{
    // src_c != 0: the frames arrive with src_c (3 | 4) channels and are converted to grey inside the
    // kernel (c == 1); T == 1 is taken too then (the per-frame kernel has no converting form)
    if ((src_c != 0 && (c != 1 || (src_c != 3 && src_c != 4))) || (src_c == 0 && T < 2))
        return hipErrorInvalidValue;
    if (n_img < 1 || h < 1 || w < 2 || T < 1 || T > kMaxT || (h + TH - 1) / TH > 65535 || n_img > 65535 ||
        st.np < (size_t)n_img * h * w || !t.fused3_ok || t.kx[0] + t.kx[2] > 255 ||
        t.vky[0] != t.vky[2] || !(k.var_min <= k.var_max))
        return hipErrorInvalidValue;
    Mog2FramesArgs a;
    a.frame = frame;
    a.mask = mask;
    a.planes = st.planes;
    a.modes = st.modes;
    a.np = st.np;
    a.frame_stride = (size_t)n_img * h * w * (src_c ? src_c : c);
    const int B2Y = 1868, G2Y = 9617, R2Y = 4899;   // (cvt_gray's weights, gof_kernels.hip)
    a.g0 = src_rgb ? R2Y : B2Y;
    a.g1 = G2Y;
    a.g2 = src_rgb ? B2Y : R2Y;
    a.mask_stride = (size_t)n_img * h * w;
    a.H = h;
    a.W = w;
    a.T = T;
    a.kxw = (uint32_t)t.kx[0] | ((uint32_t)t.kx[1] << 8) | ((uint32_t)t.kx[2] << 16);
    a.vky0 = t.vky[0];
    a.vky1 = t.vky[1];
    a.vround = t.vround;
    a.rowflags = rowflags;
    a.flag_pitch = rowflag_pitch(w);
    a.k = k;
    tap_table(a.kxw, c, a.taps);
    const dim3 g = grid3(w, h, n_img, TW, TH);
    if (src_c) {
        if (T == 1) {
            if (src_c == 3)
                hipLaunchKernelGGL((mog2_frames_kernel<1, 3, 1>), g, dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL((mog2_frames_kernel<1, 4, 1>), g, dim3(256), 0, s, a);
        } else if (src_c == 3) {
            hipLaunchKernelGGL((mog2_frames_kernel<1, 3, kMaxT>), g, dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((mog2_frames_kernel<1, 4, kMaxT>), g, dim3(256), 0, s, a);
        }
        return hipGetLastError();
    }
    switch (c) {
    case 1: hipLaunchKernelGGL(mog2_frames_kernel<1>, g, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(mog2_frames_kernel<3>, g, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(mog2_frames_kernel<4>, g, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#ifdef GOF_PATH_STATS
// (stats builds only) read and clear the counters: out[32]
extern "C" int gof_debug_path_stats(unsigned long long *out)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_path_stats), 32 * sizeof(unsigned long long));
    unsigned long long z[32] = {0};
    if (e == hipSuccess)
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_path_stats), z, sizeof z);
    return (int)e;
}
#endif

bool mog2_convert_supported(const uint8_t *frame, int n_img, int h, int w, int src_c, const Taps &t,
                            const Mog2Consts &k)
{
    const size_t fstride = (size_t)n_img * h * w * src_c;
    return (src_c == 3 || src_c == 4) && n_img >= 1 && h >= 1 && w > 1 && t.fused3_ok && t.kx[0] + t.kx[2] <= 255 &&
           t.vky[0] == t.vky[2] && k.var_min <= k.var_max && ((size_t)w * src_c) % 4 == 0 && fstride % 4 == 0 &&
           ((size_t)h * w * src_c) % 4 == 0 && (uintptr_t)frame % 4 == 0 && (h + TH - 1) / TH <= 65535 &&
           n_img <= 65535;
}

hipError_t mog2_convert_apply(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h, int w,
                              int src_c, bool src_rgb, int T, const Mog2Consts &k, const Taps &t, hipStream_t s,
                              uint8_t *rowflags)
{
    if (!mog2_convert_supported(frame, n_img, h, w, src_c, t, k))
        return hipErrorNotSupported;
    return mog2_frames_launch(frame, mask, st, n_img, h, w, 1, T, k, t, s, rowflags, src_c, src_rgb);
}

}  // namespace gofk
