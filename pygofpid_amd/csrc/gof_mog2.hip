// MOG2, one frame per launch (and every layout the multi-frame launch does not cover),
// the launcher that picks between the two, and the model helpers (export planes, mode sum).
// Compiled with -ffp-contract=off: the update must round after every multiply and add
// exactly like the scalar C++ it restates.
#include "gof_mog2_dev.h"

namespace gofk {

namespace {

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
                                                   uint32_t vky0, uint32_t vky1, uint32_t vky2, uint32_t vround,
                                                   uint8_t *__restrict__ rowflags,
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
    // vertical multipliers and rounding constant of the fused 3x3 form (Taps::vky / vround: the
    // Gaussian's taps and 2^15, or the box filter's 7282 and 4 * 7282)
    const uint32_t ky[3] = {vky0, vky1, vky2};
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
                acc[c] = vround;
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
                    acc[c] = __umul24(ky[j], __builtin_amdgcn_udot4(g, kxw, 0u, false)) + acc[c];
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

}  // namespace

hipError_t mog2_apply(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h,
                      int w, int c, int T, const Mog2Consts &k, bool fused_blur3, const Taps &t,
                      hipStream_t s, uint8_t *rowflags, bool *rowflags_written, size_t frame_bytes)
{
    if (rowflags_written)
        *rowflags_written = rowflags != nullptr;   // every MOG2 kernel writes them
    if (n_img < 1 || h < 1 || w < 1 || (h + TH - 1) / TH > 65535 || n_img > 65535 ||
        st.np < (size_t)n_img * h * w || k.nmix < 1 || k.nmix > kMaxModes || T < 1 || T > kMaxT)
        return hipErrorInvalidValue;
    if (fused_blur3 && !t.fused3_ok)
        return hipErrorInvalidValue;   // the fused horizontal pass uses byte taps (Taps::fused3_ok)
    uint32_t kx = 0;
    if (fused_blur3)
        kx = t.kx[0] | (t.kx[1] << 10) | (t.kx[2] << 20);
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
    // (its variance clamp is one v_med3_f32, which needs var_min <= var_max; anything else
    // takes the general kernel below)
    if (aln && T > 1 && w > 1 && t.kx[0] + t.kx[2] <= 255 && t.vky[0] == t.vky[2] &&
        k.var_min <= k.var_max) {   // the multi-frame launch proper
        const hipError_t e = mog2_frames_launch(frame, mask, st, n_img, h, w, c, T, k, t, s, rowflags);
        if (e != hipErrorNotSupported)
            return e;
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
                       st.planes, st.modes, st.np, h, w, T, fstride, mstride, k, kx, t.vky[0], t.vky[1], t.vky[2], t.vround, rowflags, \
                       rowflag_pitch(w), frame_end)
#define GOF_LAUNCH_MOG2(CC, BB)                                                                  \
    do {                                                                                         \
        if (T == 1 && aln && (BB) && CC == 3 && tile_rows != TH) {                               \
            const dim3 gr = grid3(w, h, n_img, TW, tile_rows);                                   \
            if (tile_rows == 2)                                                                  \
                hipLaunchKernelGGL((mog2_kernel<3, true, false, true, 2>), gr, dim3(128), 0, s, frame, mask, \
                                   st.planes, st.modes, st.np, h, w, T, fstride, mstride, k, kx, t.vky[0], t.vky[1], t.vky[2], t.vround,  \
                                   rowflags, rowflag_pitch(w), frame_end);                       \
            else                                                                                 \
                hipLaunchKernelGGL((mog2_kernel<3, true, false, true, 8>), gr, dim3(512), 0, s, frame, mask, \
                                   st.planes, st.modes, st.np, h, w, T, fstride, mstride, k, kx, t.vky[0], t.vky[1], t.vky[2], t.vround,  \
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
