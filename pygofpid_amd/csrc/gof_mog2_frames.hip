// mog2_frames_kernel: blur 3x3 + MOG2 for up to 8 consecutive frames of every stream per
// launch -- the kernel bench.py and ForegroundPipeline.apply_many run.  Its own translation
// unit (compiles in seconds); the update functions live in gof_mog2_dev.h.
// Compiled with -ffp-contract=off (see gof_mog2_dev.h / Makefile).
#include "gof_mog2_dev.h"

#include <map>
#include <mutex>
#include <string.h>
#include <utility>
#include <vector>

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
    const void *btab;                 // MB: the blur's B fragments (mfma_blur_table), 16 B per lane and fragment
    int cinit;                        // MB: 128 * (sum of all 2-D taps) + the rounding constant
    uint32_t bspecial[4];             // MB: per tile type, bit nb (low 12) = that block's fragments differ from an interior
                                      // block's; bits 16.. = an interior block of the type
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
// MB: the 3x3 blur on the matrix pipe.  The launch is bound by vector-instruction issue and a third
// of those instructions were the blur (27 v_dot4 + 6 multiply-adds per pixel and frame) while the
// matrix pipe sat idle -- the two issue side by side.  The blur of a whole tile row is the product
// of the staged bytes (A: 16 output rows x K = two staged rows x 32 bytes, v_mfma_i32_16x16x64_i8)
// with a banded matrix of the 2-D taps ky[j] kx[i] (B: host-built, mfma_blur_table: the taps split
// into two base-128 digits, border reflection folded in), so that per output byte the vector unit
// only combines the two digit sums, shifts, and stores a byte to a blurred tile in LDS; a pixel's
// lane later reads its three bytes.  Same integer: (sum of taps x pixels + rounding) >> 16.
template <int C, int CS = 0, int NT = kMaxT, bool MB = false>
__global__ __launch_bounds__(256, (C <= 3 && CS != 4) ? GOF_FRAMES_WAVES : 1) void mog2_frames_kernel(const Mog2FramesArgs a)
{
    constexpr int F = 2 + C;
    static_assert(CS == 0 || C == 1, "the conversion produces one channel");
    static_assert(!MB || (C == 3 && CS == 0 && NT % 4 == 0), "the matrix-pipe blur is built for BGR frames");
    typedef FrameTiles<(CS ? CS : C), NT> Tiles;
    constexpr int ROWD = CS ? kGrayRowD : Tiles::ROWD, TILE = (TH + 2) * ROWD;
    constexpr int BROW = TW * C;   // bytes of a blurred tile row (MB)
    __shared__ __attribute__((aligned(16))) uint32_t tiles[NT * TILE];
    __shared__ uint32_t ctiles[CS ? Tiles::LDS_DWORDS : 1];   // the raw colour tiles (CS != 0)
    __shared__ uint32_t btiles[MB ? TH * NT * BROW / 4 : 1];  // the blurred rows (MB): [wave = tile row][frame][byte]

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

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int ttype = (x0 == 0 ? 1 : 0) + (x0 + TW >= W ? 2 : 0);   // 0 interior, 1 first column inside, 2 last, 3 both

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
        M.w[0] = row_load<float>(prow, lane_col(xl) * 4u);
    // what came in, for the epilogue's "did anything change" tests: parked in LDS for the length of
    // the kernel (two registers the frame loops do not have to carry; the compiler's own answer
    // was a scratch spill, whose reload costs every wave a memory round trip at its end)
    __shared__ uint32_t keep[2 * 256];
    keep[threadIdx.x] = __float_as_uint(M.w[0]);
    keep[256 + threadIdx.x] = (uint32_t)mbyte_in;
    // lane-invariant tap patterns (three loads from the launcher's table, issued with the model
    // loads); a lane outside the image shadows the tile's first pixel
    TapPatterns<C> taps;
    if (!MB) {
        typedef const Mog2FramesArgs __attribute__((address_space(4))) *ArgsPtr0;
        // (a grey tile row starts with the pixel itself: no offset within its first dword)
        taps.load(valid ? x : x0, W, ft.sx0, CS ? 0 : ft.mis0, ((ArgsPtr0)__builtin_amdgcn_kernarg_segment_ptr())->taps);
    } else {
        taps.A = 0;
    }
    // slots 1.. are loaded only if some pixel of the wave has them; otherwise slot 1
    // starts empty (the single-mode loops may fill it) and slots 2.. stay untouched
    // until the general loop is entered
    // the staged rows go to LDS before the further modes are asked for: the 16 staging
    // registers are free again by then (settled waves, four in five, load no further mode)
    if (MB) {
        // bytes as signed (x ^ 0x80: the matrix instruction multiplies int8), and the image's first
        // tile one dword to the right, so that in EVERY tile pixel x0 - 1 sits at byte 1 of a row
        // (rows and frames start dword-aligned, 3 * 64 bytes per tile step: the launcher checks)
        static_assert(!MB || Tiles::NDW == 1, "one dword per lane and staged row");
        const int sh = x0 == 0 ? 1 : 0;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = wave + 4 * rr;
            if (r < TH + 2 && lane + sh < ROWD) {
#pragma unroll
                for (int ti = 0; ti < NT; ti++)
                    tiles[ti * TILE + r * ROWD + lane + sh] = raw[ti][rr][0] ^ 0x80808080u;
            }
        }
    } else {
        Tiles::store(CS ? ctiles : tiles, wave, lane, raw);
    }
    bool multi_in = false;
    auto load_further_modes = [&]() {
        multi_in = __builtin_amdgcn_ballot_w64(nmodes > 1) != 0;
        if (multi_in) {
            // slot m is fetched by the whole wave if any of its pixels has it (the lines travel anyway;
            // what a pixel without the slot gets is dead storage, see mog2_update): no per-lane test,
            // no zero fill
    #pragma unroll
            for (int m = 1; m < kMaxModes; m++) {
                if (__builtin_amdgcn_ballot_w64(m < nmodes) != 0) {
                    const float *pm = planes_ + (size_t)m * F * np;
                    M.w[m] = row_load<float>(pm + rowoff, lane_col(xl) * 4u);
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
    };
    // (with the blur on the matrix pipe the further modes are asked for after that phase: their twenty
    // registers do not fit beside it; settled waves, which load none, are not affected)
    if (!MB)
        load_further_modes();
    // The only barrier, as soon as this wave's tile rows are in LDS.  It orders LDS only (the
    // model loads above stay in flight across it: __syncthreads() would wait for them, and a
    // settled wave would then sit out the second round trip of a neighbour that fetches five modes).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (CS != 0) {
        // step 0 on the staged tiles, into the grey tile (gof_mog2_dev.h)
        gray_tiles_from_staged<CS, NT>(ctiles, tiles, ft, x0, W, wave, lane, a.g0, a.g1, a.g2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    if (MB) {
        // This wave blurs ITS tile row for all frames: D[n][m] = sum_K taps[n][K] * bytes[K][m], n = 16
        // byte-channels of the row (block nb), m = frame (columns 8..15 repeat 0..7), K = the 32-byte
        // windows of staged rows wave, wave + 1 (first product) and wave + 2 (second).  A lane ends up
        // with 4 consecutive byte-channels of one frame: one dword of the blurred row.  No other wave
        // is involved, so no second barrier: the row is written and read back by this wave alone.
        const int m = lane & 15, kg = lane >> 4, fr = m & (NT - 1);
        const char *p0base = (const char *)tiles + (size_t)(fr * TILE + (wave + (lane >> 5)) * ROWD) * 4 + 16 * (kg & 1);
        const char *p1base = (const char *)tiles + (size_t)(fr * TILE + (wave + 2) * ROWD) * 4 + 16 * (kg & 1);
        uint32_t *brow = btiles + (size_t)(wave * NT + fr) * (BROW / 4) + kg;
        const uint32_t special = a.bspecial[ttype] & 0xfffu, bref = a.bspecial[ttype] >> 16;
        const i32x4 *tab = (const i32x4 *)a.btab + (size_t)ttype * 12 * 4 * 64 + lane;
        // the tap fragments of an interior block (block `bref` of this tile type): 4 x 16 bytes per lane
        i32x4 tapf[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            tapf[q] = tab[(bref * 4 + q) * 64];
#pragma unroll 1
        for (int nb = 0; nb < BROW / 16; nb++) {
            const i32x4 px0 = *(const i32x4 *)(p0base + 16 * nb), px1 = *(const i32x4 *)(p1base + 16 * nb);
            // a block that holds the image's first or last column (wave-uniform, one or two per border
            // tile): its own fragments for this iteration, the interior ones back afterwards
            const bool sp = ((special >> nb) & 1u) != 0;
            if (sp) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    tapf[q] = tab[(nb * 4 + q) * 64];
            }
            i32x4 hi = {0, 0, 0, 0}, lo = {a.cinit, a.cinit, a.cinit, a.cinit};
            hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(tapf[0], px0, hi, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(tapf[1], px0, lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(tapf[2], px1, hi, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(tapf[3], px1, lo, 0, 0, 0);
            if (sp) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    tapf[q] = tab[(bref * 4 + q) * 64];
            }
            // byte 2 of (128 hi + lo) is the blurred byte (the sum is below 2^24): four of them into one dword
            uint32_t v[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                v[i] = (uint32_t)(hi[i] * 128 + lo[i]);
            const uint32_t v01 = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u);   // (v0.b2, v1.b2, 0, 0)
            const uint32_t v23 = __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);   // (0, 0, v2.b2, v3.b2)
            brow[nb * 4] = v01 | v23;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's own writes, before it reads them back)
        load_further_modes();
    }

    // window address of the current frame: into the staged tile, or (MB) the lane's bytes of the blurred tile
    constexpr uint32_t FSTEP = MB ? (uint32_t)BROW : (uint32_t)TILE * 4u;
    uint32_t wa = MB ? (uint32_t)(wave * NT * BROW) + (uint32_t)C * (valid ? (uint32_t)lane : 0u)
                     : (uint32_t)(wave * ROWD) * 4u + (uint32_t)taps.A;
    uint32_t half = a.vround;
    float var_max_v = a.k.var_max;   // (the clamp takes one scalar operand, var_min)
    asm volatile("" : "+v"(half), "+v"(var_max_v));   // kept in registers, not re-materialised per use
    const uint32_t ky[3] = {a.vky0, a.vky1, a.vky0};

    // blurred pixel of the current frame as floats (batch: the nine LDS reads issued
    // back to back -- not in the general loop, which has the whole mixture in registers)
    auto frame_input = [&](float (&xin)[C], bool batch) {
        if (MB) {
            const uint8_t *bp = (const uint8_t *)btiles + wa;
#pragma unroll
            for (int c = 0; c < C; c++)
                xin[c] = (float)bp[c];
            wa += FSTEP;
            (void)batch;
            return;
        }
        uint32_t v[C];
        taps.template blur<ROWD>(tiles, wa, ky, half, batch, v);
        wa += FSTEP;   // next frame
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
            wa -= FSTEP;
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
    const uint32_t w0_in = keep[threadIdx.x];
    const int mbyte_was = (int)keep[256 + threadIdx.x];
    const bool w0_dirty = __builtin_amdgcn_ballot_w64(nmodes > 0 && __float_as_uint(M.w[0]) != w0_in) != 0;
    const int mbyte = nmodes | ((nmodes > 0 && __float_as_uint(M.w[0]) == 0x3f800000u) ? kW0One : 0);
    const bool nm_dirty = __builtin_amdgcn_ballot_w64(mbyte != mbyte_was) != 0;
#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m < nmodes) {
            float *pm = planes_ + (size_t)m * F * np;
            if (m > 0 || w0_dirty)
                row_store<float>(pm + rowoff, lane_col(xl) * 4u, M.w[m]);
            store_mode_vec_row<C>(pm, np, rowoff, xl, M.var[m], M.mu[m]);
        }
    }
    if (nm_dirty)
        row_store<uint8_t>(mrow_modes, xl, (uint8_t)mbyte);
}

}  // namespace

// ---- the matrix-pipe blur's B fragments --------------------------------------------------------
// For tile type t (0 interior, 1 first column inside, 2 last column inside, 3 both), byte-channel
// block nb (16 of a tile row's 192) and fragment f = 2 * product + digit: 64 lanes x 16 int8.
// Product 0 multiplies staged rows r, r + 1 (K = 32 j + k), product 1 row r + 2 (K = k < 32); lane l
// holds K = 16 (l >> 4) + 0..15 of column l & 15 (checked on hardware: profiles/micro/mfma_i8_probe.hip).
// Entry = digit (W >> 7 | W & 127) of W = sum over the taps that land on staged byte k of the block's
// 32-byte window for output byte-channel 16 nb + column, REFLECT_101 folded in; pixel x0 - 1 sits at
// byte 1 of a staged row in every tile (the kernel shifts the first tile).  false: a tap product
// does not fit two base-128 digits (W >= 2^14) -- the vector-unit blur runs instead.
static bool mfma_blur_table(const Taps &t, int W, std::vector<int8_t> &tab, int *cinit)
{
    const int gx = (W + TW - 1) / TW;
    tab.assign((size_t)4 * 12 * 4 * 64 * 16, 0);
    long total = 0;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            total += (long)t.vky[j] * t.kx[i];
    *cinit = (int)(128 * total + t.vround);
    for (int type = 0; type < 4; type++) {
        if (((type & 1) && (type & 2)) != (gx == 1))
            if ((type == 3) != (gx == 1))
                continue;   // "both" exists only for a one-tile-wide image, and then it is the only type
        const int x0 = (type & 1) ? 0 : ((type & 2) ? (gx - 1) * TW : TW);
        for (int nb = 0; nb < 12; nb++) {
            // W2[n][j][k]: weight of staged byte k (of the window starting at byte 16 nb) of staged row r + j
            static thread_local long W2[16][3][32];
            memset(W2, 0, sizeof W2);
            for (int n = 0; n < 16; n++) {
                const int o = 16 * nb + n, p = o / 3, c = o % 3, x = x0 + p;
                if (x >= W)
                    continue;   // (a column of a ragged last tile outside the image: never read)
                for (int dx = -1; dx <= 1; dx++) {
                    int xi = x + dx;
                    if (xi < 0)
                        xi = W > 1 ? 1 : 0;
                    if (xi >= W)
                        xi = W > 1 ? W - 2 : 0;
                    const int byte = 1 + 3 * (xi - (x0 - 1)) + c, k = byte - 16 * nb;
                    if (k < 0 || k >= 32)
                        return false;   // (cannot happen for a reach of one pixel)
                    for (int j = 0; j < 3; j++)
                        W2[n][j][k] += (long)t.vky[j] * t.kx[dx + 1];
                }
            }
            for (int l = 0; l < 64; l++) {
                const int n = l & 15, kg = l >> 4;
                for (int jj = 0; jj < 16; jj++) {
                    const int K = 16 * kg + jj;
                    const long w0 = W2[n][K >> 5][K & 31], w1 = K < 32 ? W2[n][2][K] : 0;
                    if (w0 >= 16384 || w1 >= 16384)
                        return false;
                    const size_t base = ((((size_t)type * 12 + nb) * 4) * 64 + l) * 16 + jj;
                    tab[base + 0 * 64 * 16] = (int8_t)(w0 >> 7);
                    tab[base + 1 * 64 * 16] = (int8_t)(w0 & 127);
                    tab[base + 2 * 64 * 16] = (int8_t)(w1 >> 7);
                    tab[base + 3 * 64 * 16] = (int8_t)(w1 & 127);
                }
            }
        }
    }
    return true;
}

// device copies of the tables, one per (taps, width, device), kept for the life of the process
static const void *mfma_blur_table_dev(const Taps &t, int W)
{
    struct Key {
        uint16_t kx[3];
        uint32_t vky[3], vround;
        int W, dev;
        bool operator<(const Key &o) const { return memcmp(this, &o, sizeof(Key)) < 0; }
    };
    static std::mutex mu;
    static std::map<Key, std::pair<void *, int>> cache;
    Key k;
    memset(&k, 0, sizeof k);
    for (int i = 0; i < 3; i++) {
        k.kx[i] = t.kx[i];
        k.vky[i] = t.vky[i];
    }
    k.vround = t.vround;
    k.W = W;
    if (hipGetDevice(&k.dev) != hipSuccess)
        return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(k);
    if (it != cache.end())
        return it->second.first;
    std::vector<int8_t> tab;
    int cinit = 0;
    void *d = nullptr;
    if (mfma_blur_table(t, W, tab, &cinit)) {
        if (hipMalloc(&d, tab.size()) != hipSuccess ||
            hipMemcpy(d, tab.data(), tab.size(), hipMemcpyHostToDevice) != hipSuccess)
            d = nullptr;
    }
    cache[k] = std::make_pair(d, cinit);
    return d;
}

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
    gray_weights(src_rgb, a.g0, a.g1, a.g2);
    a.btab = nullptr;
    a.cinit = 0;
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
    // EXPERIMENTAL, off unless GOF_MFMA_BLUR is set (profiles/r03_mfma_blur.txt): bit-exact and 18 % fewer
    // vector instructions, but the phase adds ~3 us of latency to every wave and the launch is slower today
    static const bool use_mfma = getenv("GOF_MFMA_BLUR") != nullptr;
    if (c == 3 && use_mfma && (a.btab = mfma_blur_table_dev(t, w)) != nullptr) {
        long total = 0;
        for (int j = 0; j < 3; j++)
            for (int i = 0; i < 3; i++)
                total += (long)t.vky[j] * t.kx[i];
        a.cinit = (int)(128 * total + t.vround);
        // blocks whose fragments are not an interior block's: those with the first / last image column
        const int gxn = (w + TW - 1) / TW, olast = 3 * ((w - 1) - (gxn - 1) * TW);
        const uint32_t left = 1u, right = (1u << (olast / 16)) | (1u << ((olast + 2) / 16));
        const uint32_t masks[4] = {0u, left, right, left | right};
        for (int ty = 0; ty < 4; ty++) {
            // an interior block of the type: not special and (in the image's last tile) wholly inside
            // the image; if there is none, every block that matters is special
            const uint32_t inside = (ty & 2) ? (uint32_t)(olast + 3) / 16u : 12u;   // blocks [0, inside) lie inside
            uint32_t ref = 0;
            for (uint32_t r2 = 0; r2 < inside; r2++)
                if (!((masks[ty] >> r2) & 1u))
                    ref = r2;
            a.bspecial[ty] = (masks[ty] & 0xfffu) | (ref << 16);
        }
        hipLaunchKernelGGL((mog2_frames_kernel<3, 0, kMaxT, true>), g, dim3(256), 0, s, a);
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
