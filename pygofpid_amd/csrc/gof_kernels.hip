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
#include "gof_kernels.h"

#include <float.h>
#include <stdlib.h>

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
// the byte of the mode-count plane: count in the low bits, kW0One = "the first mode's
// weight is exactly 1.0f" (kept true by every kernel that writes the byte; a byte
// without the bit only costs a read of the weight plane)
constexpr int kModeCount = 0x7f, kW0One = 0x80;

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

// Model layout in HBM (gof_kernels.h, Mog2State): per mode a weight plane and a VECTOR
// plane that holds {variance, mean[0..]} of a pixel side by side -- float4 for BGR, so the
// 16 bytes a settled pixel updates per frame are ONE global_load_dwordx4 / store_dwordx4
// per lane (1 KiB per wave-instruction) instead of four dword streams; float2 for grey;
// float4 + a fifth plane for 4-channel frames.  `mode` = first float of the mode's planes,
// `pix` = pixel index (wave-uniform row offset + lane).
template <int C>
__device__ __forceinline__ void load_mode_vec(const float *mode, size_t np, size_t pix, float &var,
                                              float (&mu)[C])
{
    if (C == 1) {
        const float2 q = ((const float2 *)(mode + np))[pix];
        var = q.x;
        mu[0] = q.y;
    } else {
        const float4 q = ((const float4 *)(mode + np))[pix];
        var = q.x;
        mu[0] = q.y;
        mu[1 % C] = q.z;
        mu[2 % C] = q.w;
        if (C == 4)
            mu[3 % C] = (mode + 5 * np)[pix];
    }
}

template <int C>
__device__ __forceinline__ void store_mode_vec(float *mode, size_t np, size_t pix, float var,
                                               const float (&mu)[C])
{
    if (C == 1) {
        ((float2 *)(mode + np))[pix] = make_float2(var, mu[0]);
    } else {
        ((float4 *)(mode + np))[pix] = make_float4(var, mu[0], mu[1 % C], mu[2 % C]);
        if (C == 4)
            (mode + 5 * np)[pix] = mu[3 % C];
    }
}

// IEEE-correct a / b for operands in the normal range with a normal quotient:
// the Newton / residual chain hipcc itself emits for a float division, without
// its range scaling (v_div_scale, v_div_fmas' scale) and special-case fix-up
// (v_div_fixup), which are no-ops there.  8 instructions instead of 11; used
// where the denominator is a mixture weight (1e-4 .. 2) or a weight sum.
__device__ __forceinline__ float div_normal(float a, float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float nb = -b;
    const float e0 = __builtin_fmaf(nb, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(nb, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(nb, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}

// detectShadowGMM [OPENCV-RECALL] over the first LIM slots, on the model AFTER
// this frame's update.  Returns true for a shadow pixel.
template <int C, int LIM>
__device__ __forceinline__ bool shadow_test(const PixelModel<C> &M, int nmodes, const float (&x)[C],
                                            const Mog2Consts &k)
{
    bool shadow = false;
    if (k.detect_shadows) {
        float tW = 0.f;
        bool done = false;
#pragma unroll
        for (int m = 0; m < LIM; m++) {
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
    return shadow;
}

// One pixel of MOG2Invoker::operator() on a register-resident model.
// Statement order follows oracle/gof_oracle.c mog2_pixel(); the only
// re-statement is that the fitted mode's new weight is stored in its slot
// BEFORE it is bubbled up, so that it travels with the swaps (OpenCV swaps
// first and then writes gmm[mode - swap_count].weight: same end state).
//
// MAXIN is a compile-time upper bound of the mode count on entry.  The kernel
// calls the MAXIN = 1 instance when no pixel of the wave has a second mode
// (four waves out of five on surveillance footage): same statements, same
// order, same results, but only the slots that can be live (0 on entry, 0..1 on
// exit) are ever touched, which removes most of the unrolled predication.
template <int C, int MAXIN>
__device__ __forceinline__ uint8_t mog2_update(PixelModel<C> &M, int &nmodes_io, const float (&x)[C],
                                               const Mog2Consts &k)
{
    constexpr int LIN = MAXIN < kMaxModes ? MAXIN : kMaxModes;            // live slots on entry
    constexpr int LOUT = MAXIN + 1 < kMaxModes ? MAXIN + 1 : kMaxModes;   // ... after an insertion
    int nmodes = nmodes_io;
    bool background = false, fits = false;
    float totalWeight = 0.f;
    const float negprune = -k.prune;

#pragma unroll
    for (int m = 0; m < LIN; m++) {
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
                    float kk = div_normal(k.alphaT, weight);   // weight >= 0.95 alphaT: normal range
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
        inv = div_normal(1.f, totalWeight);
#pragma unroll
    for (int m = 0; m < LIN; m++)
        if (m < nmodes)
            M.w[m] *= inv;

    if (!fits && k.alphaT > 0.f) {
        int slot = nmodes == k.nmix ? k.nmix - 1 : nmodes++;
        bool first = nmodes == 1;
#pragma unroll
        for (int m = 0; m < LOUT; m++) {
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
        for (int i = LOUT - 1; i > 0; i--) {
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
    return shadow_test<C, LOUT>(M, nmodes, x, k) ? (uint8_t)k.shadow_value : (uint8_t)255;
}

// The MAXIN = 1 instance of mog2_update written without per-lane branches on the
// common path.  Same statements on the same values (every lane evaluates the fit
// arithmetic, selects keep it only where the reference would have executed it);
// the rare parts -- inserting a new mode, the shadow test of a non-background
// pixel -- sit behind wave-uniform branches.  On a wave of pure background this
// is a straight line of ~70 vector instructions and no exec-mask bookkeeping,
// which matters because the multi-frame kernel is bound by VALU + SALU issue.
template <int C>
__device__ __forceinline__ uint8_t mog2_update_single(PixelModel<C> &M, int &nmodes_io,
                                                      const float (&x)[C], const Mog2Consts &k)
{
    int nmodes = nmodes_io;               // 0 or 1 on entry
    const bool has = nmodes > 0;
    const float var = M.var[0];
    float d[C];
    float dist2;
    if (C == 3) {
        d[0] = M.mu[0][0] - x[0];
        d[1] = M.mu[0][1] - x[1];
        d[2] = M.mu[0][2] - x[2];
        dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    } else {
        dist2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            d[c] = M.mu[0][c] - x[c];
            dist2 += d[c] * d[c];
        }
    }
    // mean / variance of a fitted mode 0 (kk = alphaT / weight)
    auto fit_update = [&](bool fit, float kk) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float mnew = M.mu[0][c] - kk * d[c];
            M.mu[0][c] = fit ? mnew : M.mu[0][c];
        }
        float varnew = var + kk * (dist2 - var);
        varnew = varnew > k.var_min ? varnew : k.var_min;
        varnew = varnew > k.var_max ? k.var_max : varnew;
        M.var[0] = fit ? varnew : var;
    };
    bool background, fit;
    {
        const float weight = k.alpha1 * M.w[0] + k.prune;
        background = has && 0.f < k.TB && dist2 < k.Tb * var;
        fit = has && dist2 < k.Tg * var;
        const float wfit = weight + k.alphaT;
        const float kk = div_normal(k.alphaT, wfit);
        fit_update(fit, kk);
        const float w2 = fit ? wfit : weight;
        const bool pruned = has && w2 < -k.prune;
        const float wstore = pruned ? 0.f : w2;
        nmodes -= pruned ? 1 : 0;
        const float total = has ? 0.f + wstore : 0.f;
        float inv = 0.f;
        if (fabsf(total) > FLT_EPSILON)
            inv = div_normal(1.f, total);
        {
            const float wn = nmodes > 0 ? wstore * inv : wstore;
            M.w[0] = has ? wn : M.w[0];
        }
    }

    const bool need = !fit && k.alphaT > 0.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {   // wave-uniform: some pixel opens a new mode
        if (need) {
            const int slot = nmodes == k.nmix ? k.nmix - 1 : nmodes++;
            const bool first = nmodes == 1;
#pragma unroll
            for (int m = 0; m < 2; m++) {
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
            const bool sw = 1 <= nmodes - 1 && !(k.alphaT < M.w[0]);
            swap_modes<C>(M, 1, sw);
        }
    }
    nmodes_io = nmodes;

    uint8_t out = 0;
    if (__builtin_amdgcn_ballot_w64(!background) != 0) {   // wave-uniform: some pixel is not background
        if (!background)
            out = shadow_test<C, 2>(M, nmodes, x, k) ? (uint8_t)k.shadow_value : (uint8_t)255;
    }
    return out;
}

// One frame of a wave in which EVERY pixel's first mode fits (and the pixel is background):
// what a wave on an object's trail sees once the object has passed -- background pixels that
// still carry the object's and its shadow's modes for a few thousand frames.  mog2_update then
// reduces, statement for statement, to: the fitted first mode's weight / mean / variance
// update; the decay (and possible pruning, with the live loop bound) of the further modes;
// the renormalisation.  No mode is opened, none is re-ordered (the fitted first mode only
// grows, the others shrink together), no shadow test runs, every mask byte is 0.  Written as
// a straight line with selects, slots beyond the wave's largest mode count skipped by
// wave-uniform branches.  Returns false -- nothing changed -- if some pixel does not fit.
template <int C>
__device__ __forceinline__ bool mog2_try_first_fits(PixelModel<C> &M, int &nmodes_io, const float (&x)[C],
                                                    const Mog2Consts &k)
{
    const float var = M.var[0];
    float d[C];
    float dist2;
    if (C == 3) {
        d[0] = M.mu[0][0] - x[0];
        d[1] = M.mu[0][1] - x[1];
        d[2] = M.mu[0][2] - x[2];
        dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    } else {
        dist2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            d[c] = M.mu[0][c] - x[c];
            dist2 += d[c] * d[c];
        }
    }
    int nmodes = nmodes_io;
    // every pixel: a first mode, which fits, and background (totalWeight = 0 < TB, dist2 < Tb var)
    const bool ok = nmodes > 0 && dist2 < k.Tg * var && dist2 < k.Tb * var;
    if (__float_as_int(k.TB) <= 0 || __builtin_amdgcn_ballot_w64(!ok) != 0)
        return false;
    const float negprune = -k.prune;
    // m = 0, the fitted mode
    float weight = k.alpha1 * M.w[0] + k.prune;
    weight += k.alphaT;
    const float kk = div_normal(k.alphaT, weight);
#pragma unroll
    for (int c = 0; c < C; c++)
        M.mu[0][c] -= kk * d[c];
    float varnew = var + kk * (dist2 - var);
    varnew = varnew > k.var_min ? varnew : k.var_min;
    varnew = varnew > k.var_max ? k.var_max : varnew;
    M.var[0] = varnew;
    bool pruned = weight < negprune;
    float wstore = pruned ? 0.f : weight;
    nmodes -= pruned ? 1 : 0;
    M.w[0] = wstore;
    float totalWeight = 0.f + wstore;
    // m >= 1: decay, prune; the loop bound is the live mode count
#pragma unroll
    for (int m = 1; m < kMaxModes; m++) {
        if (__builtin_amdgcn_ballot_w64(m < nmodes) == 0)
            break;   // (no pixel of the wave has this slot; the later ones neither)
        const bool live = m < nmodes;
        const float wm = k.alpha1 * M.w[m] + k.prune;
        pruned = live && wm < negprune;
        wstore = pruned ? 0.f : wm;
        nmodes -= pruned ? 1 : 0;
        M.w[m] = live ? wstore : M.w[m];
        totalWeight += live ? wstore : 0.f;
    }
    float inv = 0.f;
    if (fabsf(totalWeight) > FLT_EPSILON)
        inv = div_normal(1.f, totalWeight);
#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m > 0 && __builtin_amdgcn_ballot_w64(m < nmodes) == 0)
            break;
        M.w[m] = m < nmodes ? M.w[m] * inv : M.w[m];
    }
    nmodes_io = nmodes;
    return true;
}

// One frame of a SETTLED wave.  The caller guarantees that every pixel of the wave
// has exactly one mode and that its weight is exactly 1.0f (where a fitted lone
// mode renormalises to, frame after frame).  The weight arithmetic of
// mog2_update_single is then the same for all 64 lanes and the host has already
// evaluated it with those very statements (FrameRates: kk1 = alphaT / wfit, wnfit =
// the renormalised weight of the fitted mode).  If, in addition, every pixel fits
// its mode and is background (the overwhelmingly common frame), what is left per
// lane is the distance and the mean / variance update: the function commits that
// and returns true -- every mask byte of the wave is 0.  Otherwise it changes
// nothing and returns false; the caller redoes the frame with mog2_update_single.
// `stay`: the wave is still settled after a committed frame.
template <int C>
__device__ __forceinline__ bool mog2_try_settled(PixelModel<C> &M, const float (&x)[C],
                                                 const Mog2Consts &k, float var_max_v, bool &stay)
{
    constexpr uint32_t kOne = 0x3f800000u;
    const float var = M.var[0];
    float d[C];
    float dist2;
    if (C == 3) {
        d[0] = M.mu[0][0] - x[0];
        d[1] = M.mu[0][1] - x[1];
        d[2] = M.mu[0][2] - x[2];
        dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    } else {
        dist2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            d[c] = M.mu[0][c] - x[c];
            dist2 += d[c] * d[c];
        }
    }
    const uint64_t all = __builtin_amdgcn_ballot_w64(true);
    // 0.f < TB on the scalar unit (a parameter, never NaN)
    if (__float_as_int(k.TB) <= 0 || __builtin_amdgcn_ballot_w64(dist2 < k.Tg * var) != all)
        return false;
    // Every pixel fits.  It is background if also dist2 < Tb * var; for 0 < Tg <= Tb
    // (positive floats order like their bit patterns; cv2's defaults are 9 and 16)
    // that follows: dist2 >= 0, so a fit means Tg * var > 0, var > 0, and the
    // product is monotonic in its first factor.
    const bool implied = __float_as_int(k.Tg) > 0 && __float_as_int(k.Tb) >= __float_as_int(k.Tg);
    if (!implied && __builtin_amdgcn_ballot_w64(dist2 < k.Tb * var) != all)
        return false;
#pragma unroll
    for (int c = 0; c < C; c++)
        M.mu[0][c] = M.mu[0][c] - k.kk1 * d[c];
    // median of (v, var_min, var_max) instead of two compare + select pairs: the
    // same value for every v, NaN included, given var_min <= var_max (which the host
    // checks before it allows this path)
    M.var[0] = __builtin_amdgcn_fmed3f(var + k.kk1 * (dist2 - var), k.var_min, var_max_v);
    // the fitted weight; 1.0f again (what it was) whenever the wave stays settled
    M.w[0] = k.wnfit;
    stay = __float_as_uint(k.wnfit) == kOne;
    return true;
}

// REFLECT_101 for a reach of one pixel: -1 -> 1, len -> len-2 (len == 1 -> 0)
__device__ __forceinline__ int reflect1(int p, int len)
{
    if (p < 0)
        return len > 1 ? 1 : 0;
    if (p >= len)
        return len > 1 ? len - 2 : 0;
    return p;
}

// Temporal blocking.  MOG2 is a per-pixel recurrence over time; nothing couples
// neighbouring pixels except the blur of the INPUT frame.  One launch therefore
// takes T consecutive frames of every stream: the pixel's model is loaded once,
// updated T times in registers (one mask byte written per frame) and stored
// once, which divides the dominant HBM traffic (the model, ~49 of 55 B/px) by T.
// T = 1 is the per-frame form GOFPID.detect() needs.  Results are identical to T
// single-frame launches: same statements in the same order on the same values.
// MULTI = false is the single-frame instance (T == 1 known at compile time: no
// loop-carried state, 64 VGPRs, 8 waves per SIMD).
// ALN = true: every staged frame row starts at the same byte offset mod 4 (row
// pitch and frame pitch multiples of 4, 4-byte aligned base -- the launcher
// checks), so the lane's window addresses and byte shifts are loop invariants.
// TR = rows per tile = waves per workgroup (64 * TR threads): 4 everywhere except the
// tile-size sweep of the per-frame form (profiles/micro/tile_sweep.py, GOF_T1_TILE_ROWS).
template <int C, bool BLUR3, bool MULTI, bool ALN, int TR = TH>
__global__ __launch_bounds__(64 * TR, (!MULTI && ALN && C <= 3) ? 8 : 1) void mog2_kernel(const uint8_t *__restrict__ frame,
                                                   uint8_t *__restrict__ mask,
                                                   float *__restrict__ planes,
                                                   uint8_t *__restrict__ modes, size_t np, int H,
                                                   int W, int T_arg, size_t frame_stride,
                                                   size_t mask_stride, Mog2Consts k, uint32_t kx012,
                                                   uint32_t ky012, uint8_t *__restrict__ rowflags,
                                                   int flag_pitch, const uint8_t *frame_end)
{
    constexpr int F = 2 + C;  // floats per mode
    const int T = MULTI ? T_arg : 1;
    const int lane = threadIdx.x & 63;
    // wave-uniform values are forced into SGPRs: row index, row bases, plane bases
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TR;
    const int img = blockIdx.z;
    const int x = x0 + lane, y = y0 + wave;
    const size_t N = (size_t)H * W;

    // raw (TR+2) x (TW+2) px input tiles of ALL frames of the launch (<= 10 KB):
    // every global load of the kernel is issued up front, one barrier, and the
    // frame loop below touches only LDS and registers
    constexpr int ROWD = ((TW + 2) * C + 3 + 3) / 4 + 1;   // + misalignment slack + read-ahead of px()
    constexpr int TILE = (TR + 2) * ROWD;
    constexpr int NT = MULTI ? kMaxT : 1;
    __shared__ uint32_t tiles[BLUR3 ? NT * TILE : 1];
    const int sx0 = x0 > 0 ? x0 - 1 : 0;
    const int sx1 = (x0 + TW + 1) < W ? (x0 + TW + 1) : W;
    const int nb = (sx1 - sx0) * C;

    if (BLUR3) {
        // each wave stages rows {wave, wave+4} of every frame: raw bytes, aligned
        // dword loads; borders are resolved when the tile is read
        constexpr int NDW = (ROWD + 63) / 64;
        uint32_t raw[NT][2][NDW];
#pragma unroll
        for (int ti = 0; ti < NT; ti++) {
            if (ti < T) {
                const uint8_t *fimg = frame + (size_t)ti * frame_stride + (size_t)img * N * C;
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int r = wave + TR * rr;
                    const int gy = reflect1(y0 - 1 + (r < TR + 2 ? r : 0), H);
                    const int gyc = gy < H ? gy : H - 1;   // rows below a ragged last tile
                    const uint8_t *a = fimg + ((size_t)gyc * W + sx0) * C;
                    const int mis = (int)((uintptr_t)a & 3);
                    const int ndw = (mis + nb + 3) >> 2;
#pragma unroll
                    for (int q = 0; q < NDW; q++) {
                        const int d = lane + 64 * q;
                        uint32_t v = 0u;
                        if (d < ndw) {
                            const uint8_t *p4 = a - mis + 4 * d;
                            // Aligned frames end on a dword boundary.  Otherwise the dword
                            // holding a row's first / last byte may straddle the caller's
                            // buffer [frame, frame_end): those (at most two per launch) are
                            // assembled from the bytes inside it -- nothing outside is read.
                            if (ALN || (p4 >= frame && p4 + 4 <= frame_end)) {
                                v = *(const uint32_t *)p4;
                            } else {
#pragma unroll
                                for (int b = 0; b < 4; b++)
                                    if (p4 + b >= frame && p4 + b < frame_end)
                                        v |= (uint32_t)p4[b] << (8 * b);
                            }
                        }
                        raw[ti][rr][q] = v;
                    }
                }
            }
        }
#pragma unroll
        for (int ti = 0; ti < NT; ti++) {
            if (ti < T) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int r = wave + TR * rr;
                    if (r < TR + 2) {
#pragma unroll
                        for (int q = 0; q < NDW; q++) {
                            const int d = lane + 64 * q;
                            if (d < ROWD)
                                tiles[ti * TILE + r * ROWD + d] = raw[ti][rr][q];
                        }
                    }
                }
            }
        }
    }

    // ---- model: plane p of this wave's row starts at a wave-uniform address,
    //      the lane only adds x; mode count + mode 0 are loaded speculatively ----
    const bool rowok = y < H;
    const int yy = rowok ? y : 0;
    const bool valid = rowok && x < W;
    const unsigned xl = valid ? (unsigned)x : 0u;           // lane offset within the row
    const size_t rowoff = (size_t)img * N + (size_t)yy * W; // wave-uniform
    float *const planes_ = planes;
    float *const prow = planes + rowoff;
    PixelModel<C> M;
    // the mode-count byte also says (bit 7, kW0One) that the first mode's weight is
    // exactly 1.0f -- the settled state of most pixels -- so that a wave whose pixels
    // all say so does not read the weight plane at all
    const int mbyte_in = valid ? (int)modes[rowoff + xl] : 0;
    int nmodes = mbyte_in & kModeCount;
    load_mode_vec<C>(planes_, np, rowoff + xl, M.var[0], M.mu[0]);
    M.w[0] = 1.f;
    if (__builtin_amdgcn_ballot_w64(valid && !(mbyte_in & kW0One)) != 0)
        M.w[0] = prow[xl];
    const uint32_t w0_in = __float_as_uint(M.w[0]);
    // remaining modes only if some pixel of the wave has one (wave-uniform split)
    if (__builtin_amdgcn_ballot_w64(nmodes > 1) != 0) {
#pragma unroll
        for (int m = 1; m < kMaxModes; m++) {
            if (m < nmodes) {
                const float *pm = planes_ + (size_t)m * F * np;
                M.w[m] = pm[rowoff + xl];
                load_mode_vec<C>(pm, np, rowoff + xl, M.var[m], M.mu[m]);
            } else {
                M.w[m] = 0.f;
                M.var[m] = 0.f;
#pragma unroll
                for (int c = 0; c < C; c++)
                    M.mu[m][c] = 0.f;
            }
        }
    } else {
#pragma unroll
        for (int m = 1; m < kMaxModes; m++) {
            M.w[m] = 0.f;
            M.var[m] = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++)
                M.mu[m][c] = 0.f;
        }
    }

    // the three horizontal taps as bytes (< 256 each, the launcher checks)
    const uint32_t kxw = (kx012 & 1023u) | (((kx012 >> 10) & 1023u) << 8) | ((kx012 >> 20) << 16);
    const int ky[3] = {(int)(ky012 & 1023), (int)((ky012 >> 10) & 1023), (int)(ky012 >> 20)};
    // byte offsets (within a staged row) of the left / centre / right pixel
    const int xc = valid ? x : sx0;
    const int pl = (reflect1(xc - 1, W) - sx0) * C, pc = (xc - sx0) * C,
              pr = (reflect1(xc + 1, W) - sx0) * C;

    // per-frame learning rates (host-computed) parked in LDS: a wave-uniform LDS
    // read per frame instead of a select chain over the kernel arguments
    __shared__ float rates_lds[2 * kMaxT];
    if (MULTI) {
#pragma unroll
        for (int q = 0; q < kMaxT; q++) {
            if (threadIdx.x == (unsigned)q) {
                rates_lds[q] = k.alphaT_f[q];
                rates_lds[kMaxT + q] = k.prune_f[q];
            }
        }
    }
    if (BLUR3 || MULTI)
        __syncthreads();   // the only barrier: all tiles (and the rates) are in LDS

    // invariant window addressing (ALN): byte address of the dword holding the
    // left / centre / right pixel in staged row `wave` of frame 0, and the shifts
    uint32_t aL = 0, aC = 0, aR = 0, sL = 0, sC = 0, sR = 0;
    if (BLUR3 && ALN) {
        const int mis0 = (int)((uintptr_t)(frame + ((size_t)img * N * C) + (size_t)sx0 * C) & 3);
        const uint32_t base = (uint32_t)(wave * ROWD) * 4u;
        aL = base + (uint32_t)((mis0 + pl) & ~3);
        aC = base + (uint32_t)((mis0 + pc) & ~3);
        aR = base + (uint32_t)((mis0 + pr) & ~3);
        sL = (uint32_t)(mis0 + pl) & 3u;
        sC = (uint32_t)(mis0 + pc) & 3u;
        sR = (uint32_t)(mis0 + pr) & 3u;
    }

    // blurred (or raw) pixel of frame ti as floats
    auto frame_input = [&](int ti, float (&xin)[C]) {
        const uint8_t *fimg = frame + (size_t)ti * frame_stride + (size_t)img * N * C;
        if (BLUR3) {
            const uint32_t *tile = tiles + ti * TILE;
            uint32_t acc[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                acc[c] = 32768u;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                uint32_t L, Cn, R;
                if (ALN) {
                    // row j of frame ti: a wave-uniform offset from the lane's frame-0 addresses
                    const uint32_t off = (uint32_t)(ti * TILE + j * ROWD) * 4u;
                    auto win = [&](uint32_t a, uint32_t sh) -> uint32_t {
                        const uint32_t *q = (const uint32_t *)((const char *)tiles + a + off);
                        return __builtin_amdgcn_alignbyte(q[1], q[0], sh);
                    };
                    L = win(aL, sL);
                    Cn = win(aC, sC);
                    R = win(aR, sR);
                } else {
                    const int r = wave + j;
                    const int gy = reflect1(y0 - 1 + r, H);
                    const int gyc = gy < H ? gy : H - 1;
                    const int mis = (int)((uintptr_t)(fimg + ((size_t)gyc * W + sx0) * C) & 3);
                    const uint32_t *row = tile + r * ROWD;
                    // C consecutive bytes starting at byte offset o of the staged row
                    auto px = [&](int o) -> uint32_t {
                        o += mis;
                        const uint32_t lo = row[o >> 2], hi = row[(o >> 2) + 1];
                        return __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(o & 3));
                    };
                    L = px(pl);
                    Cn = px(pc);
                    R = px(pr);
                }
                // horizontal pass: gather (L_c, C_c, R_c, 0) with two v_perm and take the
                // dot product with the byte taps (v_dot4_u32_u8); exact, the taps are < 256
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const uint32_t t = __builtin_amdgcn_perm(Cn, L, 0x0C0C0000u | ((4u + c) << 8) | (uint32_t)c);
                    const uint32_t g = __builtin_amdgcn_perm(R, t, 0x0C000100u | ((4u + c) << 16));
                    // 24-bit multiply-add: the sums are < 2^16, the taps < 2^9 (full-rate
                    // v_mad_u32_u24 instead of the quarter-rate 32-bit multiply)
                    acc[c] = __umul24((uint32_t)ky[j], __builtin_amdgcn_udot4(g, kxw, 0u, false)) + acc[c];
                }
            }
#pragma unroll
            for (int c = 0; c < C; c++)
                xin[c] = (float)(acc[c] >> 16);
        } else {
            const uint8_t *fp = fimg + ((size_t)yy * W + xl) * C;
#pragma unroll
            for (int c = 0; c < C; c++)
                xin[c] = (float)fp[c];
        }
    };
    // per-frame learning rate (host-computed, one wave-uniform LDS read per frame)
    auto frame_consts = [&](int ti) -> Mog2Consts {
        Mog2Consts kt = k;
        if (MULTI) {
            kt.alphaT = rates_lds[ti];
            kt.alpha1 = 1.f - kt.alphaT;
            kt.prune = rates_lds[kMaxT + ti];
        }
        return kt;
    };

    // Two frame loops instead of one with a branch inside: while no pixel of the
    // wave has a second mode (four waves out of five, and it stays that way for
    // whole launches) only slots 0..1 are live, so that loop carries a fraction of
    // the registers; the first frame that sees a second mode anywhere in the wave
    // hands the rest of the launch to the general loop.
    int ti = 0;
    uint32_t nz = 0;
    for (; ti < T; ti++) {
        if (__builtin_amdgcn_ballot_w64(nmodes > 1) != 0)
            break;
        float xin[C];
        frame_input(ti, xin);
        const Mog2Consts kt = frame_consts(ti);
        uint8_t out = 0;
        if (valid) {
            out = mog2_update_single<C>(M, nmodes, xin, kt);
            mask[(size_t)ti * mask_stride + rowoff + xl] = out;
        }
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
    }
    for (; ti < T; ti++) {
        float xin[C];
        frame_input(ti, xin);
        const Mog2Consts kt = frame_consts(ti);
        uint8_t out = 0;
        if (valid) {
            out = mog2_update<C, kMaxModes>(M, nmodes, xin, kt);
            mask[(size_t)ti * mask_stride + rowoff + xl] = out;
        }
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
    }
    // one flag byte per wave and launch for the morphology that follows: bit t =
    // some mask byte of frame t is not zero in this 64-px row segment
    if (rowflags != nullptr && rowok && lane == 0)
        rowflags[((size_t)img * H + y) * flag_pitch + blockIdx.x] = (uint8_t)nz;
    if (!valid)
        return;

    // Planes whose value did not change for ANY pixel of the wave are not written
    // back (wave-uniform test, so a store is always a whole 256-B segment): the
    // weight of a settled single-mode pixel renormalises to exactly 1.0f frame
    // after frame, and the mode count rarely changes.
    const bool w0_dirty = __builtin_amdgcn_ballot_w64(nmodes > 0 && __float_as_uint(M.w[0]) != w0_in) != 0;
    const int mbyte = nmodes | ((nmodes > 0 && __float_as_uint(M.w[0]) == 0x3f800000u) ? kW0One : 0);
    const bool nm_dirty = __builtin_amdgcn_ballot_w64(mbyte != mbyte_in) != 0;
#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m < nmodes) {
            float *pm = planes_ + (size_t)m * F * np;
            if (m > 0 || w0_dirty)
                pm[rowoff + xl] = M.w[m];
            store_mode_vec<C>(pm, np, rowoff + xl, M.var[m], M.mu[m]);
        }
    }
    if (nm_dirty)
        modes[rowoff + xl] = (uint8_t)mbyte;
}

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
// ---- pieces the multi-frame kernels share -------------------------------------
// Geometry of the staged tiles: (TH + 2) x (TW + 2) px of each of up to kMaxT frames,
// raw bytes, LDS rows of a 64-dword pitch per 64 lanes.
template <int C, int NT = kMaxT>
struct FrameTiles {
    static constexpr int NDW = (((TW + 2) * C + 3 + 3) / 4 + 1 + 63) / 64;   // dwords per lane and staged row
    static constexpr int ROWD = NDW * 64;   // LDS row pitch in dwords: every lane writes, no predicate
    static constexpr int TILE = (TH + 2) * ROWD;
    static constexpr int LDS_DWORDS = NT * TILE;
    int sx0;    // first staged image column
    int mis0;   // byte offset of that column's pixel within its aligned dword (same in every row / frame)

    // Issue the loads of rows {wave, wave + 4} of all frames: two running row
    // pointers (every staged row of every frame has the same offset mod 4, which the
    // launcher checks), one shared lane offset, no predicates -- lanes past the
    // row's last dword re-read it, frames past T re-read frame T - 1.
    __device__ __forceinline__ void load(const uint8_t *frame, size_t frame_stride, int H, int W, int T,
                                         int img, int x0, int y0, int wave, int lane,
                                         uint32_t (&raw)[NT][2][NDW])
    {
        const size_t N = (size_t)H * W;
        sx0 = x0 > 0 ? x0 - 1 : 0;
        const int sx1 = (x0 + TW + 1) < W ? (x0 + TW + 1) : W;
        const int nb = (sx1 - sx0) * C;
        const uint8_t *img0 = frame + (size_t)img * N * C;
        mis0 = (int)((uintptr_t)(img0 + (size_t)sx0 * C) & 3);
        const int ndw = (mis0 + nb + 3) >> 2;
        typedef const uint8_t __attribute__((address_space(1))) *GlobalBytes;
        GlobalBytes rowp[2];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = wave + 4 * rr;
            const int gy = reflect1(y0 - 1 + (r < TH + 2 ? r : 0), H);
            const int gyc = gy < H ? gy : H - 1;   // rows below a ragged last tile
            rowp[rr] = (GlobalBytes)(img0 + ((size_t)gyc * W + sx0) * C - mis0);
        }
        // (a full launch, T == NT, advances the pointers unconditionally: a third fewer
        // scalar instructions on the path every wave of the benchmark takes)
        if (T == NT) {
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
#pragma unroll
                    for (int q = 0; q < NDW; q++) {
                        const unsigned d = lane + 64 * q < ndw ? lane + 64 * q : ndw - 1;
                        raw[ti][rr][q] = *(const uint32_t __attribute__((address_space(1))) *)(rowp[rr] + d * 4u);   // SGPR base + 32-bit lane offset
                    }
#ifndef GOF_ABLATE_FRAME0   // (timing ablation: every frame of a launch reads frame 0's rows)
                    if (ti + 1 < NT)
                        rowp[rr] += frame_stride;
#endif
                }
            }
        } else {
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
                const size_t step = ti + 1 < T ? frame_stride : 0;
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
#pragma unroll
                    for (int q = 0; q < NDW; q++) {
                        const unsigned d = lane + 64 * q < ndw ? lane + 64 * q : ndw - 1;
                        raw[ti][rr][q] = *(const uint32_t __attribute__((address_space(1))) *)(rowp[rr] + d * 4u);
                    }
                    rowp[rr] += step;
                }
            }
        }
    }

    __device__ __forceinline__ static void store(uint32_t *tiles, int wave, int lane,
                                                 const uint32_t (&raw)[NT][2][NDW])
    {
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = wave + 4 * rr;
            if (r < TH + 2) {   // one wave-uniform test per staged row, all frames inside
#pragma unroll
                for (int ti = 0; ti < NT; ti++) {
#pragma unroll
                    for (int q = 0; q < NDW; q++)
                        tiles[ti * TILE + r * ROWD + lane + 64 * q] = raw[ti][rr][q];
                }
            }
        }
    }
};

// Lane-invariant blur addressing.  The three pixels a lane needs from a staged row
// (left, centre, right; the border reflection can fold left onto right) lie within
// 12 bytes of the aligned address A below the leftmost of them.  Instead of
// extracting them (alignbyte + byte permutes), the lane carries, per channel, a
// 12-byte tap pattern: the horizontal pass of a row is then three v_dot4_u32_u8 per
// channel on the three aligned dwords, nothing else.  xc: the lane's image column
// (a lane outside the image passes the tile's first column).
template <int C>
struct TapPatterns {
    uint32_t pat[C][3];
    int A;   // byte offset of the window within a staged row

    __device__ __forceinline__ void build(int xc, int W, int sx0, int mis0, uint32_t kxw)
    {
        const int pl = (reflect1(xc - 1, W) - sx0) * C + mis0, pc = (xc - sx0) * C + mis0,
                  pr = (reflect1(xc + 1, W) - sx0) * C + mis0;
        const int pmin = pl < pc ? (pl < pr ? pl : pr) : (pc < pr ? pc : pr);
        A = pmin & ~3;
        // channel 0: byte k0 at pl - A, k1 at pc - A, k2 at pr - A (k0 + k2 where the
        // reflection folds them; the launcher checks that the sum fits a byte),
        // as a 96-bit number (lo, hi); channel c is that number shifted by c bytes
        const uint32_t kx[3] = {kxw & 255u, (kxw >> 8) & 255u, (kxw >> 16) & 255u};
        const int rel[3] = {pl - A, pc - A, pr - A};   // each 0 .. 3 + 2 C
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const uint32_t sh = 8u * (uint32_t)(rel[t] & 7);
            if (rel[t] < 8)
                lo += (uint64_t)kx[t] << sh;
            else
                hi += kx[t] << sh;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint64_t l = c ? lo << (8 * c) : lo;
            const uint32_t h = c ? (hi << (8 * c)) | (uint32_t)(lo >> (64 - 8 * c)) : hi;
            pat[c][0] = (uint32_t)l;
            pat[c][1] = (uint32_t)(l >> 32);
            pat[c][2] = h;
        }
    }

    // blurred pixel from the 3 x 3 aligned dwords at window address wa of a tile:
    // horizontal pass as dot products against the patterns, vertical pass as 24-bit
    // multiply-adds onto the rounding constant (`half`, 2^15 in a VGPR: the
    // multiply-add takes one scalar operand), >> 16 -- the integer of the separable
    // 8.8 blur of blur_kernel (sums of the same products, no intermediate rounding)
    template <int ROWD>
    __device__ __forceinline__ void blur(const uint32_t *tiles, uint32_t wa, const uint32_t (&ky)[3],
                                         uint32_t half, bool batch, uint32_t (&out)[C]) const
    {
#ifdef GOF_ABLATE_BLUR
        // timing ablation only (profiles/): the centre row's first three window bytes instead
        // of the blurred pixel -- bounds what a free blur would buy; results are NOT the path's
        {
            const uint32_t *q = (const uint32_t *)((const char *)tiles + wa + (uint32_t)ROWD * 4u);
            const uint32_t v = q[0];
#pragma unroll
            for (int c = 0; c < C; c++)
                out[c] = (v >> (8 * (c & 3))) & 255u;
            (void)ky;
            (void)half;
            (void)batch;
            return;
        }
#endif
        uint32_t win[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t *q = (const uint32_t *)((const char *)tiles + wa + (uint32_t)(j * ROWD) * 4u);
#pragma unroll
            for (int i = 0; i < 3; i++)
                win[j][i] = q[i];
        }
        // keeps all nine live at one point: the reads are issued back to back
        if (batch)
            asm volatile(""
                         : "+v"(win[0][0]), "+v"(win[0][1]), "+v"(win[0][2]), "+v"(win[1][0]),
                           "+v"(win[1][1]), "+v"(win[1][2]), "+v"(win[2][0]), "+v"(win[2][1]),
                           "+v"(win[2][2]));
#pragma unroll
        for (int c = 0; c < C; c++) {
            // The vertical taps of a Gaussian are symmetric (ky[0] == ky[2]; the launcher
            // checks): rows 0 and 2 share one dot-product chain -- the accumulator carries
            // h0 into h2 -- and one multiply-add, 2 v_mad_u32_u24 per channel instead of 3.
            uint32_t h02 = __builtin_amdgcn_udot4(win[0][0], pat[c][0], 0u, false);
            h02 = __builtin_amdgcn_udot4(win[0][1], pat[c][1], h02, false);
            h02 = __builtin_amdgcn_udot4(win[0][2], pat[c][2], h02, false);
            h02 = __builtin_amdgcn_udot4(win[2][0], pat[c][0], h02, false);
            h02 = __builtin_amdgcn_udot4(win[2][1], pat[c][1], h02, false);
            h02 = __builtin_amdgcn_udot4(win[2][2], pat[c][2], h02, false);
            uint32_t h1 = __builtin_amdgcn_udot4(win[1][0], pat[c][0], 0u, false);
            h1 = __builtin_amdgcn_udot4(win[1][1], pat[c][1], h1, false);
            h1 = __builtin_amdgcn_udot4(win[1][2], pat[c][2], h1, false);
            // acc += ky * h as ONE v_mad_u32_u24 each (h02 < 2^17, ky < 2^10).  The empty asm
            // pins the running sum: without it the products are re-associated into
            // multiply, multiply, add.  (The multiply-add itself must stay
            // compiler-generated: a VALU read of a v_dot4 result needs wait states that
            // only the compiler inserts.)
            uint32_t acc = __umul24(ky[0], h02) + half;
            asm volatile("" : "+v"(acc));
            acc = __umul24(ky[1], h1) + acc;
            asm volatile("" : "+v"(acc));
            out[c] = acc >> 16;
        }
    }
};

struct Mog2FramesArgs {
    const uint8_t *frame;
    uint8_t *mask;
    float *planes;
    uint8_t *modes;
    size_t np, frame_stride, mask_stride;
    int H, W, T;
    uint32_t kxw;     // horizontal taps as bytes (k0 | k1 << 8 | k2 << 16)
    uint32_t ky012;   // vertical taps, 10 bits each
    uint8_t *rowflags;   // [image][row][tile column] bytes, bit t = frame t's mask not all zero there (or NULL)
    int flag_pitch;
    Mog2Consts k;
};

// (7 waves per SIMD for BGR: the issue-bound launch needs every one of them; the float4
// model loads alone would let the allocator drift to 74 registers = 6 waves, 8 % slower)
template <int C>
__global__ __launch_bounds__(256, C <= 3 ? 7 : 1) void mog2_frames_kernel(const Mog2FramesArgs a)
{
    constexpr int F = 2 + C;
    typedef FrameTiles<C> Tiles;
    constexpr int ROWD = Tiles::ROWD, TILE = Tiles::TILE;
    __shared__ uint32_t tiles[Tiles::LDS_DWORDS];

    const int H = a.H, W = a.W, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int img = blockIdx.z;
    const int x = x0 + lane, y = y0 + wave;
    const size_t N = (size_t)H * W;

    // ---- stage the tiles of all T frames (loads now, LDS writes after the model
    //      loads are in flight); borders are resolved at read time ----
    Tiles ft;
    uint32_t raw[kMaxT][2][Tiles::NDW];
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
    const int mbyte_in = (int)a.modes[rowoff + xl];   // mode count | kW0One (see mog2_kernel)
    int nmodes = mbyte_in & kModeCount;
    load_mode_vec<C>(planes_, np, rowoff + xl, M.var[0], M.mu[0]);
    M.w[0] = 1.f;
    if (__builtin_amdgcn_ballot_w64(!(mbyte_in & kW0One)) != 0)
        M.w[0] = prow[xl];
    const uint32_t w0_in = __float_as_uint(M.w[0]);
    // slots 1.. are loaded only if some pixel of the wave has them; otherwise slot 1
    // starts empty (the single-mode loops may fill it) and slots 2.. stay untouched
    // until the general loop is entered
    // the staged rows go to LDS before the further modes are asked for: the 16 staging
    // registers are free again by then (settled waves, four in five, load no further mode)
    Tiles::store(tiles, wave, lane, raw);
    const bool multi_in = __builtin_amdgcn_ballot_w64(nmodes > 1) != 0;
    if (multi_in) {
#pragma unroll
        for (int m = 1; m < kMaxModes; m++) {
            if (m < nmodes) {
                const float *pm = planes_ + (size_t)m * F * np;
                M.w[m] = pm[rowoff + xl];
                load_mode_vec<C>(pm, np, rowoff + xl, M.var[m], M.mu[m]);
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


    // lane-invariant tap patterns; a lane outside the image shadows the tile's first pixel
    TapPatterns<C> taps;
    taps.build(valid ? x : x0, W, ft.sx0, ft.mis0, a.kxw);
    uint32_t wa = (uint32_t)(wave * ROWD) * 4u + (uint32_t)taps.A;   // window address, current frame
    uint32_t half = 32768u;
    float var_max_v = a.k.var_max;   // (the clamp takes one scalar operand, var_min)
    asm volatile("" : "+v"(half), "+v"(var_max_v));   // kept in registers, not re-materialised per use
    const uint32_t ky[3] = {a.ky012 & 1023u, (a.ky012 >> 10) & 1023u, a.ky012 >> 20};

    __syncthreads();   // the only barrier: every tile is in LDS

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
            mrow[xl] = 0;
        mrow += a.mask_stride;
        rates = next;
        ti++;
    }
    for (; ti < T; ti++) {
        if (__builtin_amdgcn_ballot_w64(nmodes > 1) != 0)
            break;
        const u32x8 next = load_rates(ti + 1);
        const Mog2Consts kt = frame_consts(rates);
        float xin[C];
        frame_input(xin, true);
        const uint8_t out = mog2_update_single<C>(M, nmodes, xin, kt);
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
        if (valid)
            mrow[xl] = out;
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
        if (!mog2_try_first_fits<C>(M, nmodes, xin, kt))
            out = mog2_update<C, kMaxModes>(M, nmodes, xin, kt);
        nz |= (__builtin_amdgcn_ballot_w64(out != 0) != 0 ? 1u : 0u) << ti;
        if (valid)
            mrow[xl] = out;
        mrow += a.mask_stride;
        rates = next;
    }
    // one flag byte per wave and launch for the morphology that follows (the settled
    // loop only ever writes zeros, so it contributes no bits)
    if (a.rowflags != nullptr && y < H && lane == 0)
        a.rowflags[((size_t)img * H + y) * a.flag_pitch + blockIdx.x] = (uint8_t)nz;
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
                pm[rowoff + xl] = M.w[m];
            store_mode_vec<C>(pm, np, rowoff + xl, M.var[m], M.mu[m]);
        }
    }
    if (nm_dirty)
        a.modes[rowoff + xl] = (uint8_t)mbyte;
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
    uint32_t kxw, ky012;
};

template <int C, int NT>   // NT: frames staged per launch (1 for the per-frame call, else kMaxT)
__global__ __launch_bounds__(256) void fd_frames_kernel(const FdFramesArgs a)
{
    typedef FrameTiles<C, NT> Tiles;
    constexpr int ROWD = Tiles::ROWD, TILE = Tiles::TILE;
    __shared__ uint32_t tiles[Tiles::LDS_DWORDS];

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

    Tiles::store(tiles, wave, lane, raw);
    TapPatterns<C> taps;
    taps.build(valid ? x : x0, W, ft.sx0, ft.mis0, a.kxw);
    uint32_t wa = (uint32_t)(wave * ROWD) * 4u + (uint32_t)taps.A;
    uint32_t half = 32768u;
    asm volatile("" : "+v"(half));
    const uint32_t ky[3] = {a.ky012 & 1023u, (a.ky012 >> 10) & 1023u, a.ky012 >> 20};
    __syncthreads();

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

// One mode of the model between its HBM layout (weight plane + vector plane, see
// load_mode_vec) and plain structure-of-arrays planes soa[f * n + i], f = 0 weight,
// 1 variance, 2 + c mean[c]: what gof_mog2_export / import hand to the host.
template <int C, bool TO_SOA>
__global__ __launch_bounds__(256) void mode_soa_kernel(float *__restrict__ mode, size_t np, size_t n,
                                                       float *__restrict__ soa)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    float var, mu[C];
    if (TO_SOA) {
        load_mode_vec<C>(mode, np, i, var, mu);
        soa[i] = mode[i];
        soa[n + i] = var;
#pragma unroll
        for (int c = 0; c < C; c++)
            soa[(size_t)(2 + c) * n + i] = mu[c];
    } else {
        var = soa[n + i];
#pragma unroll
        for (int c = 0; c < C; c++)
            mu[c] = soa[(size_t)(2 + c) * n + i];
        mode[i] = soa[i];
        store_mode_vec<C>(mode, np, i, var, mu);
    }
}

__global__ __launch_bounds__(256) void sum_modes_kernel(const uint8_t *__restrict__ modes, size_t n,
                                                        unsigned long long *out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        s += modes[i] & kModeCount;
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
                      int w, int c, int T, const Mog2Consts &k, bool fused_blur3, const Taps &t,
                      hipStream_t s, uint8_t *rowflags, bool *rowflags_written, size_t frame_bytes)
{
    if (rowflags_written)
        *rowflags_written = rowflags != nullptr;   // every MOG2 kernel writes them
    if (n_img < 1 || h < 1 || w < 1 || (h + TH - 1) / TH > 65535 || n_img > 65535 ||
        st.np < (size_t)n_img * h * w || k.nmix < 1 || k.nmix > kMaxModes || T < 1 || T > kMaxT)
        return hipErrorInvalidValue;
    if (fused_blur3 && (t.nkx != 3 || t.nky != 3 || t.kx[0] > 255 || t.kx[1] > 255 || t.kx[2] > 255))
        return hipErrorInvalidValue;   // the fused horizontal pass uses byte taps
    uint32_t kx = 0, ky = 0;
    if (fused_blur3) {
        kx = t.kx[0] | (t.kx[1] << 10) | (t.kx[2] << 20);
        ky = t.ky[0] | (t.ky[1] << 10) | (t.ky[2] << 20);
    }
    // frames [T][n_img][h][w][c], masks [T][n_img][h][w]
    const size_t fstride = (size_t)n_img * h * w * c, mstride = (size_t)n_img * h * w;
    if (frame_bytes < (size_t)T * fstride)
        return hipErrorInvalidValue;
    const uint8_t *frame_end = frame + (size_t)T * fstride;   // the kernels read nothing past it
    dim3 g = grid3(w, h, n_img, TW, TH);
    // invariant window addressing needs every staged row of every frame to start at
    // the same offset mod 4
    const bool aln = fused_blur3 && ((size_t)w * c) % 4 == 0 && fstride % 4 == 0 &&
                     ((size_t)h * w * c) % 4 == 0 && (uintptr_t)frame % 4 == 0;
    // (its tap patterns add the outer taps where the border reflection folds them)
    if (aln && T > 1 && w > 1 && t.kx[0] + t.kx[2] <= 255 && t.ky[0] == t.ky[2]) {   // the multi-frame launch proper
        Mog2FramesArgs a;
        a.frame = frame;
        a.mask = mask;
        a.planes = st.planes;
        a.modes = st.modes;
        a.np = st.np;
        a.frame_stride = fstride;
        a.mask_stride = mstride;
        a.H = h;
        a.W = w;
        a.T = T;
        a.kxw = (uint32_t)t.kx[0] | ((uint32_t)t.kx[1] << 8) | ((uint32_t)t.kx[2] << 16);
        a.ky012 = ky;
        a.rowflags = rowflags;
        a.flag_pitch = rowflag_pitch(w);
        a.k = k;
        switch (c) {
        case 1: hipLaunchKernelGGL(mog2_frames_kernel<1>, g, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(mog2_frames_kernel<3>, g, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(mog2_frames_kernel<4>, g, dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    // A/B knob of the LDS tile-size sweep (BASELINE config 3): rows per tile of the per-frame
    // BGR kernel, 2 | 4 | 8 (64 px wide, one wave per row)
    // Default 2: the sweep (profiles/r02_tile_sweep.txt) has 2 rows ahead of 4 and 8 at
    // 2160p (151 / 155 / 167 us) and level at 1080p -- more, smaller workgroups balance
    // better than the halo rows they re-read cost (those come from L2).
    static const int tile_rows = [] {
        const char *e = getenv("GOF_T1_TILE_ROWS");
        const int v = e ? atoi(e) : 2;
        return v == 2 || v == 8 ? v : TH;
    }();
#define GOF_LAUNCH_MOG2_(CC, BB, MM, AA)                                                         \
    hipLaunchKernelGGL((mog2_kernel<CC, BB, MM, AA>), g, dim3(256), 0, s, frame, mask,           \
                       st.planes, st.modes, st.np, h, w, T, fstride, mstride, k, kx, ky, rowflags,      \
                       rowflag_pitch(w), frame_end)
#define GOF_LAUNCH_MOG2(CC, BB)                                                                  \
    do {                                                                                         \
        if (T == 1 && aln && (BB) && CC == 3 && tile_rows != TH) {                               \
            const dim3 gr = grid3(w, h, n_img, TW, tile_rows);                                   \
            if (tile_rows == 2)                                                                  \
                hipLaunchKernelGGL((mog2_kernel<3, true, false, true, 2>), gr, dim3(128), 0, s, frame, mask, \
                                   st.planes, st.modes, st.np, h, w, T, fstride, mstride, k, kx, ky,  \
                                   rowflags, rowflag_pitch(w), frame_end);                       \
            else                                                                                 \
                hipLaunchKernelGGL((mog2_kernel<3, true, false, true, 8>), gr, dim3(512), 0, s, frame, mask, \
                                   st.planes, st.modes, st.np, h, w, T, fstride, mstride, k, kx, ky,  \
                                   rowflags, rowflag_pitch(w), frame_end);                       \
        } else if (T == 1 && aln)                                                                \
            GOF_LAUNCH_MOG2_(CC, BB, false, (BB));                                               \
        else if (T == 1)                                                                         \
            GOF_LAUNCH_MOG2_(CC, BB, false, false);                                              \
        else                                                                                     \
            GOF_LAUNCH_MOG2_(CC, BB, true, false);                                               \
    } while (0)
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
#undef GOF_LAUNCH_MOG2_
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

bool fd_frames_supported(const uint8_t *frame, int n_img, int h, int w, int c, const Taps &t)
{
    // byte taps whose folded outer pair fits a byte, at least two columns, and every
    // staged row of every frame at the same offset mod 4 (see mog2_apply)
    return t.nkx == 3 && t.nky == 3 && t.box_scale == 0.0 && t.ky[0] == t.ky[2] && t.kx[0] < 256 && t.kx[1] < 256 &&
           t.kx[2] < 256 && t.kx[0] + t.kx[2] <= 255 && w > 1 && ((size_t)w * c) % 4 == 0 &&
           ((size_t)h * w * c) % 4 == 0 && ((size_t)n_img * h * w * c) % 4 == 0 &&
           (uintptr_t)frame % 4 == 0 && (c == 1 || c == 3 || c == 4);
}

hipError_t fd_frames_apply(const uint8_t *frame, uint8_t *mask, const uint32_t *prev_in,
                           uint32_t *prev_out, bool has_prev, int ithr, int n_img, int h, int w, int c,
                           int T, const Taps &t, hipStream_t s, uint8_t *rowflags)
{
    if (n_img < 1 || h < 1 || (h + TH - 1) / TH > 65535 || n_img > 65535 || T < 1 || T > kMaxT ||
        !fd_frames_supported(frame, n_img, h, w, c, t) || !prev_out || (has_prev && !prev_in))
        return hipErrorInvalidValue;
    FdFramesArgs a;
    a.frame = frame;
    a.mask = mask;
    a.prev_in = prev_in;
    a.prev_out = prev_out;
    a.rowflags = rowflags;
    a.frame_stride = (size_t)n_img * h * w * c;
    a.mask_stride = (size_t)n_img * h * w;
    a.H = h;
    a.W = w;
    a.T = T;
    a.has_prev = has_prev ? 1 : 0;
    a.ithr = ithr;
    a.flag_pitch = rowflag_pitch(w);
    a.kxw = (uint32_t)t.kx[0] | ((uint32_t)t.kx[1] << 8) | ((uint32_t)t.kx[2] << 16);
    a.ky012 = t.ky[0] | (t.ky[1] << 10) | (t.ky[2] << 20);
    dim3 g = grid3(w, h, n_img, TW, TH);
#define GOF_LAUNCH_FD(CC)                                                                   \
    do {                                                                                    \
        if (T == 1)                                                                         \
            hipLaunchKernelGGL((fd_frames_kernel<CC, 1>), g, dim3(256), 0, s, a);           \
        else                                                                                \
            hipLaunchKernelGGL((fd_frames_kernel<CC, kMaxT>), g, dim3(256), 0, s, a);       \
    } while (0)
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
    size_t blocks = (npix + 255) / 256;
    if (blocks > 0x7fffffffu)
        return hipErrorInvalidValue;
    const int B2Y = 1868, G2Y = 9617, R2Y = 4899;
    hipLaunchKernelGGL(gray_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, npix, c,
                       rgb ? R2Y : B2Y, G2Y, rgb ? B2Y : R2Y);
    return hipGetLastError();
}

hipError_t mog2_mode_soa(const Mog2State &st, int c, int mode, size_t n, float *soa, bool to_soa,
                         hipStream_t s)
{
    if (n < 1 || n > st.np || mode < 0 || mode >= kMaxModes || !soa)
        return hipErrorInvalidValue;
    float *mp = st.planes + (size_t)mode * (2 + c) * st.np;
    const dim3 g((unsigned)((n + 255) / 256));
#define GOF_SOA(CC)                                                                              \
    do {                                                                                         \
        if (to_soa)                                                                              \
            hipLaunchKernelGGL((mode_soa_kernel<CC, true>), g, dim3(256), 0, s, mp, st.np, n, soa);  \
        else                                                                                     \
            hipLaunchKernelGGL((mode_soa_kernel<CC, false>), g, dim3(256), 0, s, mp, st.np, n, soa); \
    } while (0)
    switch (c) {
    case 1: GOF_SOA(1); break;
    case 3: GOF_SOA(3); break;
    case 4: GOF_SOA(4); break;
    default: return hipErrorInvalidValue;
    }
#undef GOF_SOA
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
