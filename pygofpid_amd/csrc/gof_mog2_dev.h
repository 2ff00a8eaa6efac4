// Device-side pieces the MOG2 kernels share (gof_mog2.hip: per-frame kernel and launcher,
// gof_mog2_frames.hip: the multi-frame launch) and the staging / tap-pattern front end that
// fd_frames_kernel (gof_kernels.hip) borrows.  Internal; everything here has internal linkage.
#pragma once
#include "gof_kernels.h"

#include <float.h>
#include <stdlib.h>

namespace gofk {

namespace {

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

// Path statistics of the multi-frame kernel (profiles/micro/path_stats.py; -DGOF_PATH_STATS builds
// only, never the product library): wave-uniform counts kept in scalar registers, one atomic per
// counter and wave at the end.  The macros vanish in the product build.
#ifdef GOF_PATH_STATS
struct PathStats {
    uint32_t v[32];
};
enum {
    kStSettled, kStSingle, kStFirstFits, kStGeneral, kStGenLanesNotOk, kStGenInsert, kStGenReorder,
    kStGenNotBg, kStGenLanesNotBg, kStGenSumMaxModes, kStGenFitLate, kStGenSlot1, kStGenSlot2, kStGenSlot3,
    kStGenSlot4, kStWaves, kStWavesMulti, kStFit0, kStFit1, kStFit2, kStFit3, kStFit4, kStShadowIters,
    kStGenLanesInsert, kStInsertReorder, kStCount
};
#define GOF_STAT_ARG , PathStats &ps
#define GOF_STAT_PASS , ps
#define GOF_STAT(i, x) ps.v[i] += (uint32_t)(x)
#define GOF_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0 ? 1u : 0u)
#define GOF_POP(c) ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c)))
#else
#define GOF_STAT_ARG
#define GOF_STAT_PASS
#define GOF_STAT(i, x) ((void)0)
#endif

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

// The same through the wave's row: `rowoff` = pixel index of the row's first pixel (wave-uniform, so
// the plane's row base sits in scalar registers) and the lane's column `xl` -- a global access with
// an SGPR base and a 32-bit VGPR byte offset (the offset is formed in 32-bit arithmetic: a row is
// far below 2^32 bytes), no 64-bit per-lane address kept live across the frame loops.
typedef const char __attribute__((address_space(1))) *GlobalRow;
typedef char __attribute__((address_space(1))) *GlobalRowW;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// (the empty asm pins the base as ONE scalar value: otherwise the bases of the mode planes are
// re-associated into a shared per-lane 64-bit address plus scalar deltas, and the access falls
// back to a 64-bit VGPR address; it pins the 32-bit offset in the block of the access, where
// instruction selection has to see its zero extension.  row_base MUST be wave-uniform.)
template <typename T>
__device__ __forceinline__ T row_load(const void *row_base, unsigned byte_off)
{
    GlobalRow b = (GlobalRow)row_base;
    asm("" : "+s"(b), "+v"(byte_off));
    return *(const T __attribute__((address_space(1))) *)(b + byte_off);
}
template <typename T>
__device__ __forceinline__ void row_store(void *row_base, unsigned byte_off, T v)
{
    GlobalRowW b = (GlobalRowW)row_base;
    asm("" : "+s"(b), "+v"(byte_off));
    *(T __attribute__((address_space(1))) *)(b + byte_off) = v;
}

// the lane's column as it is wanted at an access: opaque, so that the scaled offsets (x 4, x 16) are
// formed next to their use instead of living in registers of their own for the whole kernel
__device__ __forceinline__ unsigned lane_col(unsigned xl)
{
    asm("" : "+v"(xl));
    return xl;
}

template <int C>
__device__ __forceinline__ void load_mode_vec_row(const float *mode, size_t np, size_t rowoff, unsigned xl,
                                                  float &var, float (&mu)[C])
{
    if (C == 1) {
        const f32x2 q = row_load<f32x2>((const float2 *)(mode + np) + rowoff, lane_col(xl) * 8u);
        var = q.x;
        mu[0] = q.y;
    } else {
        const f32x4 q = row_load<f32x4>((const float4 *)(mode + np) + rowoff, lane_col(xl) * 16u);
        var = q.x;
        mu[0] = q.y;
        mu[1 % C] = q.z;
        mu[2 % C] = q.w;
        if (C == 4)
            mu[3 % C] = row_load<float>(mode + 5 * np + rowoff, lane_col(xl) * 4u);
    }
}

template <int C>
__device__ __forceinline__ void store_mode_vec_row(float *mode, size_t np, size_t rowoff, unsigned xl, float var,
                                                   const float (&mu)[C])
{
    if (C == 1) {
        row_store<f32x2>((float2 *)(mode + np) + rowoff, lane_col(xl) * 8u, f32x2{var, mu[0]});
    } else {
        row_store<f32x4>((float4 *)(mode + np) + rowoff, lane_col(xl) * 16u, f32x4{var, mu[0], mu[1 % C], mu[2 % C]});
        if (C == 4)
            row_store<float>(mode + 5 * np + rowoff, lane_col(xl) * 4u, mu[3 % C]);
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
                                            const Mog2Consts &k GOF_STAT_ARG)
{
    bool shadow = false;
    if (k.detect_shadows) {
        float tW = 0.f;
        bool done = false;
#pragma unroll
        for (int m = 0; m < LIM; m++) {
#ifdef GOF_PATH_STATS
            ps.v[31] |= (!done && m < nmodes) ? 1u << m : 0u;   // per lane: the slots this lane examined
#endif
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
//
// What a wave rarely does sits behind wave-uniform branches (ballots), so that the common frame
// does not pay for it in selects: re-ordering after a fit (a position of the bubble-up is entered
// only if some lane moves there), opening a new mode, and the re-ordering behind that.
// MED3: the caller guarantees var_min <= var_max (the launcher checks), so that the variance
// clamp is one v_med3_f32 -- the same value as the two compare / select pairs for every input,
// NaN included (mog2_try_settled).
template <int C, int MAXIN, bool MED3 = false>
__device__ __forceinline__ uint8_t mog2_update(PixelModel<C> &M, int &nmodes_io, const float (&x)[C],
                                               const Mog2Consts &k GOF_STAT_ARG)
{
    constexpr int LIN = MAXIN < kMaxModes ? MAXIN : kMaxModes;            // live slots on entry
    constexpr int LOUT = MAXIN + 1 < kMaxModes ? MAXIN + 1 : kMaxModes;   // ... after an insertion
    int nmodes = nmodes_io;
    bool background = false, fits = false;
    float totalWeight = 0.f;
    const float negprune = -k.prune;
#ifdef GOF_PATH_STATS
    {
        int mx = 0;
#pragma unroll
        for (int m = 1; m <= kMaxModes; m++)
            mx = __builtin_amdgcn_ballot_w64(nmodes >= m) != 0 ? m : mx;
        ps.v[kStGenSumMaxModes] += (uint32_t)mx;
    }
    bool st_reorder = false, st_late = false;
#endif

#pragma unroll
    for (int m = 0; m < LIN; m++) {
#ifdef GOF_PATH_STATS
        if (m > 0)
            ps.v[kStGenSlot1 + m - 1] += GOF_ANY(m < nmodes);
#endif
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
                    if (MED3) {
                        varnew = __builtin_amdgcn_fmed3f(varnew, k.var_min, k.var_max);
                    } else {
                        varnew = varnew > k.var_min ? varnew : k.var_min;
                        varnew = varnew > k.var_max ? k.var_max : varnew;
                    }
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
            // bubble the fitted mode up while its weight is not below its predecessor's: a
            // position is entered only if some lane of the wave moves there (rare)
            if (m > 0 && __builtin_amdgcn_ballot_w64(fit_here && !(weight < M.w[m - 1])) != 0) {
                bool go = fit_here;
#pragma unroll
                for (int i = m; i > 0; i--) {
                    go = go && !(weight < M.w[i - 1]);
                    if (i < m && __builtin_amdgcn_ballot_w64(go) == 0)
                        break;
#ifdef GOF_PATH_STATS
                    st_reorder = st_reorder || go;
#endif
                    swap_modes<C>(M, i, go);
                }
            }
#ifdef GOF_PATH_STATS
            st_late = st_late || (fit_here && m > 0);
            ps.v[kStFit0 + m] += GOF_POP(fit_here);
#endif
        }
    }
    GOF_STAT(kStGenReorder, GOF_ANY(st_reorder));
    GOF_STAT(kStGenFitLate, GOF_ANY(st_late));
    GOF_STAT(kStGenInsert, GOF_ANY(!fits && k.alphaT > 0.f));
    GOF_STAT(kStGenLanesInsert, GOF_POP(!fits && k.alphaT > 0.f));
    GOF_STAT(kStGenNotBg, GOF_ANY(!background));
    GOF_STAT(kStGenLanesNotBg, GOF_POP(!background));

    float inv = 0.f;
    if (fabsf(totalWeight) > FLT_EPSILON)
        inv = div_normal(1.f, totalWeight);
    // (slots at or beyond the mode count are dead storage -- never read before a new mode
    // rewrites every field of them, never stored -- so they are scaled along: no per-slot test)
#pragma unroll
    for (int m = 0; m < LIN; m++)
        M.w[m] *= inv;

    // a new mode where nothing fitted: the last slot (replacing the weakest mode when all nmix
    // are in use); the modes in front of it decay once more; then it is bubbled up like a fit
    const bool need = !fits && k.alphaT > 0.f;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {   // wave-uniform: some pixel opens a new mode
        int slot = -1;
        if (need) {
            slot = nmodes == k.nmix ? k.nmix - 1 : nmodes++;   // == the new count - 1
            const float w_new = slot == 0 ? 1.f : k.alphaT;
#pragma unroll
            for (int m = 0; m < LOUT; m++) {
                const bool here = m == slot;
                M.w[m] = here ? w_new : (m < slot ? M.w[m] * k.alpha1 : M.w[m]);
                M.var[m] = here ? k.var_init : M.var[m];
#pragma unroll
                for (int c = 0; c < C; c++)
                    M.mu[m][c] = here ? x[c] : M.mu[m][c];
            }
        }
        bool go = need;
#ifdef GOF_PATH_STATS
        bool st_sw = false;
#endif
#pragma unroll
        for (int i = LOUT - 1; i > 0; i--) {
            // position i is the new mode's only while every swap above it happened
            const bool sw = go && i <= slot && !(k.alphaT < M.w[i - 1]);
            go = i <= slot ? sw : go;
            if (__builtin_amdgcn_ballot_w64(sw) != 0)
                swap_modes<C>(M, i, sw);
#ifdef GOF_PATH_STATS
            st_sw = st_sw || sw;
#endif
        }
        GOF_STAT(kStInsertReorder, GOF_ANY(st_sw));
    }
    nmodes_io = nmodes;

#ifdef GOF_PATH_STATS
    {   // the shadow test, run here so that the per-lane slot bits can be counted wave-uniformly
        PathStats pl;
        pl.v[31] = 0;
        uint8_t out = 0;
        if (!background)
            out = shadow_test<C, LOUT>(M, nmodes, x, k, pl) ? (uint8_t)k.shadow_value : (uint8_t)255;
#pragma unroll
        for (int m = 0; m < kMaxModes; m++)
            ps.v[kStShadowIters] += GOF_ANY(!background && ((pl.v[31] >> m) & 1u));
        return out;
    }
#else
    if (background)
        return 0;
    return shadow_test<C, LOUT>(M, nmodes, x, k) ? (uint8_t)k.shadow_value : (uint8_t)255;
#endif
}

// The MAXIN = 1 instance of mog2_update written without per-lane branches on the
// common path.  Same statements on the same values (every lane evaluates the fit
// arithmetic, selects keep it only where the reference would have executed it);
// the rare parts -- inserting a new mode, the shadow test of a non-background
// pixel -- sit behind wave-uniform branches.  On a wave of pure background this
// is a straight line of ~70 vector instructions and no exec-mask bookkeeping,
// which matters because the multi-frame kernel is bound by VALU + SALU issue.
template <int C, bool MED3 = false>
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
        if (MED3) {
            varnew = __builtin_amdgcn_fmed3f(varnew, k.var_min, k.var_max);
        } else {
            varnew = varnew > k.var_min ? varnew : k.var_min;
            varnew = varnew > k.var_max ? k.var_max : varnew;
        }
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
        {
#ifdef GOF_PATH_STATS
            PathStats ps;   // (not counted: the single-mode loop's shadow tests)
#pragma unroll
            for (int i = 0; i < 32; i++)
                ps.v[i] = 0;
#endif
            out = shadow_test<C, 2>(M, nmodes, x, k GOF_STAT_PASS) ? (uint8_t)k.shadow_value : (uint8_t)255;
        }
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
template <int C, bool MED3 = false>
__device__ __forceinline__ bool mog2_try_first_fits(PixelModel<C> &M, int &nmodes_io, const float (&x)[C],
                                                    const Mog2Consts &k GOF_STAT_ARG)
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
    if (__float_as_int(k.TB) <= 0 || __builtin_amdgcn_ballot_w64(!ok) != 0) {
        GOF_STAT(kStGenLanesNotOk, GOF_POP(!ok));
        return false;
    }
    const float negprune = -k.prune;
    // m = 0, the fitted mode
    float weight = k.alpha1 * M.w[0] + k.prune;
    weight += k.alphaT;
    const float kk = div_normal(k.alphaT, weight);
#pragma unroll
    for (int c = 0; c < C; c++)
        M.mu[0][c] -= kk * d[c];
    float varnew = var + kk * (dist2 - var);
    if (MED3) {
        varnew = __builtin_amdgcn_fmed3f(varnew, k.var_min, k.var_max);
    } else {
        varnew = varnew > k.var_min ? varnew : k.var_min;
        varnew = varnew > k.var_max ? k.var_max : varnew;
    }
    M.var[0] = varnew;
    bool pruned = weight < negprune;
    float wstore = pruned ? 0.f : weight;
    nmodes -= pruned ? 1 : 0;
    M.w[0] = wstore;
    float totalWeight = 0.f + wstore;
    // m >= 1: decay, prune; the loop bound is the live mode count
    int top = 1;   // wave-uniform: slots [0, top) were visited (what lies beyond a pixel's count is dead storage)
#pragma unroll
    for (int m = 1; m < kMaxModes; m++) {
        if (__builtin_amdgcn_ballot_w64(m < nmodes) == 0)
            break;   // (no pixel of the wave has this slot; the later ones neither)
        top = m + 1;
        const bool live = m < nmodes;
        const float wm = k.alpha1 * M.w[m] + k.prune;
        pruned = live && wm < negprune;
        wstore = pruned ? 0.f : wm;
        nmodes -= pruned ? 1 : 0;
        M.w[m] = wstore;
        totalWeight += live ? wstore : 0.f;
    }
    float inv = 0.f;
    if (fabsf(totalWeight) > FLT_EPSILON)
        inv = div_normal(1.f, totalWeight);
#pragma unroll
    for (int m = 0; m < kMaxModes; m++) {
        if (m >= top)
            break;
        M.w[m] *= inv;
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

// Step 0 (cv.cvtColor to grey, pipeline.py:458-459) on staged tiles: the frames arrive with CS (3 | 4)
// channels, FrameTiles<CS, NT> stages their rows as it stages any frame, and this turns every staged
// pixel of every staged row of every frame -- (B * g0 + G * g1 + R * g2 + 2^13) >> 14, gray_kernel's
// statement -- into a grey tile of kGrayRowD dwords per row (pixel sx0 at byte 0), on which the blur
// runs with a zero start offset: no grey image ever exists in HBM.  Between two workgroup barriers
// (the colour tiles complete before, the grey tiles complete after).
constexpr int kGrayRowD = 20;   // dwords per grey tile row: 66 px + the 12-byte window's reach

// Four pixels at a time: their 12 (CS = 3) or 16 (CS = 4) bytes as aligned dwords, the 14-bit weights
// split into low and high bytes, one v_dot4_u32_u8 per (dword, pixel, half) that holds channel bytes of
// the pixel -- grey = (lo + 256 hi + 2^13) >> 14 is the statement above with the products regrouped
// (exact integers either way); the four grey bytes leave as one dword.
template <int CS>
struct GrayQuad {
    uint32_t lo[CS == 3 ? 6 : 1], hi[CS == 3 ? 6 : 1];   // byte patterns (wave-uniform)

    __device__ __forceinline__ static uint32_t pat(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
    {
        return a | (b << 8) | (c << 16) | (d << 24);
    }
    __device__ __forceinline__ void build(int g0, int g1, int g2)
    {
        const uint32_t w[3] = {(uint32_t)g0, (uint32_t)g1, (uint32_t)g2};
        uint32_t l[3], h[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            l[k] = w[k] & 255u;
            h[k] = w[k] >> 8;
        }
        if (CS == 3) {
            // bytes of the quad: c0 c1 c2 c0 | c1 c2 c0 c1 | c2 c0 c1 c2
            lo[0] = pat(l[0], l[1], l[2], 0), hi[0] = pat(h[0], h[1], h[2], 0);   // pixel 0 in dword 0
            lo[1] = pat(0, 0, 0, l[0]), hi[1] = pat(0, 0, 0, h[0]);               // pixel 1 in dword 0
            lo[2] = pat(l[1], l[2], 0, 0), hi[2] = pat(h[1], h[2], 0, 0);         // pixel 1 in dword 1
            lo[3] = pat(0, 0, l[0], l[1]), hi[3] = pat(0, 0, h[0], h[1]);         // pixel 2 in dword 1
            lo[4] = pat(l[2], 0, 0, 0), hi[4] = pat(h[2], 0, 0, 0);               // pixel 2 in dword 2
            lo[5] = pat(0, l[0], l[1], l[2]), hi[5] = pat(0, h[0], h[1], h[2]);   // pixel 3 in dword 2
        } else {
            lo[0] = pat(l[0], l[1], l[2], 0), hi[0] = pat(h[0], h[1], h[2], 0);
        }
    }
    __device__ __forceinline__ static uint32_t finish(uint32_t l0, uint32_t h0, uint32_t l1, uint32_t h1, uint32_t l2,
                                                      uint32_t h2, uint32_t l3, uint32_t h3)
    {
        // (the low sums start from 2^13)
        const uint32_t y0 = (l0 + (h0 << 8)) >> 14, y1 = (l1 + (h1 << 8)) >> 14;
        const uint32_t y2 = (l2 + (h2 << 8)) >> 14, y3 = (l3 + (h3 << 8)) >> 14;
        return y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
    }
    // d[0 .. CS): the quad's dwords
    __device__ __forceinline__ uint32_t convert(const uint32_t (&d)[CS]) const
    {
        constexpr uint32_t R = 1u << 13;
        if constexpr (CS == 3) {
            const uint32_t l0 = __builtin_amdgcn_udot4(d[0], lo[0], R, false), h0 = __builtin_amdgcn_udot4(d[0], hi[0], 0u, false);
            uint32_t l1 = __builtin_amdgcn_udot4(d[0], lo[1], R, false), h1 = __builtin_amdgcn_udot4(d[0], hi[1], 0u, false);
            l1 = __builtin_amdgcn_udot4(d[1], lo[2], l1, false), h1 = __builtin_amdgcn_udot4(d[1], hi[2], h1, false);
            uint32_t l2 = __builtin_amdgcn_udot4(d[1], lo[3], R, false), h2 = __builtin_amdgcn_udot4(d[1], hi[3], 0u, false);
            l2 = __builtin_amdgcn_udot4(d[2], lo[4], l2, false), h2 = __builtin_amdgcn_udot4(d[2], hi[4], h2, false);
            const uint32_t l3 = __builtin_amdgcn_udot4(d[2], lo[5], R, false), h3 = __builtin_amdgcn_udot4(d[2], hi[5], 0u, false);
            return finish(l0, h0, l1, h1, l2, h2, l3, h3);
        } else {
            uint32_t l[4], h[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                l[k] = __builtin_amdgcn_udot4(d[k], lo[0], R, false);
                h[k] = __builtin_amdgcn_udot4(d[k], hi[0], 0u, false);
            }
            return finish(l[0], h[0], l[1], h[1], l[2], h[2], l[3], h[3]);
        }
    }
};

template <int CS, int NT>
__device__ __forceinline__ void gray_tiles_from_staged(const uint32_t *ctiles, uint32_t *tiles,
                                                       const FrameTiles<CS, NT> &ft, int /*x0*/, int /*W*/, int /*wave*/,
                                                       int /*lane*/, int g0, int g1, int g2)
{
    // a task = four consecutive staged pixels of one staged row of one frame: CS aligned dwords in
    // (one more and a byte shift when the row's first pixel does not start its dword), one dword out.
    // 17 quads cover a row's <= 66 staged pixels (what lies beyond them is never read back); the
    // colour row's pitch covers the reads (CS = 3: dwords < 3 * 17 + 1 <= 64).
    constexpr int QUADS = (TW + 2 + 3) / 4, CROWD = FrameTiles<CS, NT>::ROWD, NTASK = NT * (TH + 2) * QUADS;
    static_assert(QUADS * CS + 1 <= CROWD && QUADS <= kGrayRowD, "quads stay inside the tile rows");
    GrayQuad<CS> gq;
    gq.build(g0, g1, g2);
    const uint32_t sh = (uint32_t)ft.mis0;   // bytes (0 for CS = 4: a pixel is a dword)
    for (int id = (int)threadIdx.x; id < NTASK; id += 256) {
        const int row = id / QUADS, k = id - row * QUADS;
        const uint32_t *c = ctiles + (size_t)row * CROWD + k * CS;
        uint32_t d[CS];
        if constexpr (CS == 3) {
            const uint32_t r0 = c[0], r1 = c[1], r2 = c[2], r3 = c[3];
            d[0] = __builtin_amdgcn_alignbyte(r1, r0, sh);
            d[1] = __builtin_amdgcn_alignbyte(r2, r1, sh);
            d[2] = __builtin_amdgcn_alignbyte(r3, r2, sh);
        } else {
#pragma unroll
            for (int q = 0; q < CS; q++)
                d[q] = c[q];
        }
        tiles[(size_t)row * kGrayRowD + k] = gq.convert(d);
    }
}

// the luma weights of source channels 0 / 1 / 2 (cvt_gray's, 14-bit)
inline void gray_weights(bool src_rgb, int &g0, int &g1, int &g2)
{
    const int B2Y = 1868, G2Y = 9617, R2Y = 4899;
    g0 = src_rgb ? R2Y : B2Y;
    g1 = G2Y;
    g2 = src_rgb ? B2Y : R2Y;
}

// Lane-invariant blur addressing.  The three pixels a lane needs from a staged row
// (left, centre, right; the border reflection can fold left onto right) lie within
// 12 bytes of the aligned address A below the leftmost of them.  Instead of
// extracting them (alignbyte + byte permutes), the lane carries, per channel, a
// 12-byte tap pattern: the horizontal pass of a row is then three v_dot4_u32_u8 per
// channel on the three aligned dwords, nothing else.  xc: the lane's image column
// (a lane outside the image passes the tile's first column).
constexpr int kTapRow = 12;                  // dwords per row of the tap table (C <= 4 channels x 3)
constexpr int kTapTableDwords = 12 * kTapRow;   // rows: (interior, first column, last column) x (pmin & 3)

// TapPatterns::build on the host, for every (kind, start offset) a lane can have: byte rel[t] of the
// 12-byte window carries tap t (taps that the border reflection folds onto one byte add up).
inline void tap_table(uint32_t kxw, int C, uint32_t (&table)[kTapTableDwords])
{
    const uint32_t kx[3] = {kxw & 255u, (kxw >> 8) & 255u, (kxw >> 16) & 255u};
    for (int kind = 0; kind < 3; kind++) {
        for (int a = 0; a < 4; a++) {
            // offsets of the left / centre / right pixel from the window's first dword
            const int rel[3] = {kind == 1 ? a + C : a, kind == 1 ? a : a + C, kind == 0 ? a + 2 * C : (kind == 1 ? a + C : a)};
            for (int c = 0; c < 4; c++) {
                uint32_t bytes[16] = {0};
                if (c < C)
                    for (int t = 0; t < 3; t++)
                        bytes[rel[t] + c] += kx[t];
                for (int i = 0; i < 3 && c * 3 + i < kTapRow; i++)
                    table[(kind * 4 + a) * kTapRow + c * 3 + i] =
                        bytes[4 * i] | (bytes[4 * i + 1] << 8) | (bytes[4 * i + 2] << 16) | (bytes[4 * i + 3] << 24);
            }
        }
    }
}

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

    // The same patterns from a table the launcher built (tap_table below): a lane's patterns
    // depend only on where its window starts within a dword (pmin & 3) and on whether it sits on
    // the image's first / last column -- 12 rows of kTapRow dwords, fetched with three loads
    // instead of ~50 vector instructions of shifting per wave.  Needs W > 1.
    __device__ __forceinline__ void load(int xc, int W, int sx0, int mis0,
                                         const uint32_t __attribute__((address_space(4))) *table)
    {
        const bool left = xc == 0, right = xc == W - 1;
        const int pc = (xc - sx0) * C + mis0;
        const int pmin = left ? pc : pc - C;
        A = pmin & ~3;
        const uint32_t row = (left ? 4u : right ? 8u : 0u) + (uint32_t)(pmin & 3);
        const uint32_t __attribute__((address_space(4))) *r = table + row * kTapRow;
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
            for (int i = 0; i < 3; i++)
                pat[c][i] = r[c * 3 + i];
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
        if (batch) {
            // the nine reads back to back as single ds_read_b32 (the compiler pairs them into
            // ds_read2_b32 and then spends four moves per frame untangling the register pairs);
            // the window's LDS address = the tile array's LDS offset (0: it is the kernels' only
            // LDS array) + wa
            const uint32_t lds_wa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)tiles + wa;
            asm volatile("ds_read_b32 %0, %9\n\t"
                         "ds_read_b32 %1, %9 offset:4\n\t"
                         "ds_read_b32 %2, %9 offset:8\n\t"
                         "ds_read_b32 %3, %9 offset:%c10\n\t"
                         "ds_read_b32 %4, %9 offset:%c11\n\t"
                         "ds_read_b32 %5, %9 offset:%c12\n\t"
                         "ds_read_b32 %6, %9 offset:%c13\n\t"
                         "ds_read_b32 %7, %9 offset:%c14\n\t"
                         "ds_read_b32 %8, %9 offset:%c15\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(win[0][0]), "=&v"(win[0][1]), "=&v"(win[0][2]), "=&v"(win[1][0]),
                           "=&v"(win[1][1]), "=&v"(win[1][2]), "=&v"(win[2][0]), "=&v"(win[2][1]),
                           "=&v"(win[2][2])
                         : "v"(lds_wa), "n"(ROWD * 4), "n"(ROWD * 4 + 4), "n"(ROWD * 4 + 8), "n"(ROWD * 8),
                           "n"(ROWD * 8 + 4), "n"(ROWD * 8 + 8)
                         : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const uint32_t *q = (const uint32_t *)((const char *)tiles + wa + (uint32_t)(j * ROWD) * 4u);
#pragma unroll
                for (int i = 0; i < 3; i++)
                    win[j][i] = q[i];
            }
        }
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



inline dim3 grid3(int w, int h, int n, int tw, int th)
{
    return dim3((unsigned)((w + tw - 1) / tw), (unsigned)((h + th - 1) / th), (unsigned)n);
}

}  // namespace

}  // namespace gofk
