// Internal launch interface between the C ABI (gof_api.cpp) and the gfx950
// kernels (gof_kernels.hip).  Not installed; the public surface is
// include/gofpid_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gofk {

constexpr int kMaxModes = 5;
constexpr int kMaxKsize = 31;

// Scalars of one frame of a multi-frame launch: one 32-byte record, fetched by
// the kernel with a single scalar load from its argument segment.
//   alphaT / alpha1 / prune: the learning-rate terms (alpha1 = 1.f - alphaT).
//   kk1 / wnfit / wnmiss: the same frame's weight arithmetic for a pixel whose
//   only mode has weight exactly 1.0f (the settled state: a fitted lone mode
//   renormalises to 1.0f), evaluated on the host with the kernel's own float
//   statements.  When every pixel of a wave is in that state the kernel takes
//   these instead of recomputing a wave-uniform division chain per lane:
//     weight1 = alpha1 * 1.f + prune; wfit1 = weight1 + alphaT; kk1 = alphaT / wfit1;
//     wnfit = wfit1 * (1.f / wfit1); wnmiss = weight1 * (1.f / weight1)
//   uni_ok == 0 switches the shortcut off for this frame (a rate at which the
//   lone mode would be pruned, or a weight sum not above FLT_EPSILON).
struct alignas(32) FrameRates {
    float alphaT, alpha1, prune;
    float kk1, wnfit, wnmiss;
    uint32_t uni_ok, pad_;
};

// Per-frame MOG2 scalars (cv2 BackgroundSubtractorMOG2Impl::apply computes the
// rates on the host once per frame; restated in gof_api.cpp).
struct Mog2Consts {
    FrameRates fr[8];   // the (up to kMaxT) frames of one launch
    uint32_t uni_mask;  // bit i: fr[i].uni_ok
    // learning rates of the (up to kMaxT) frames of one launch, computed on the
    // host exactly as cv2 does [BackgroundSubtractorMOG2Impl::apply]:
    //   lr = 1.0 / min(2 n, history) in double; alphaT = float(lr);
    //   prune = float(-lr * ct);  (alpha1 = 1.f - alphaT is formed in the kernel)
    float alphaT_f[8], prune_f[8];
    float alphaT, alpha1, prune;   // scratch: the current frame's values
    // scratch for the settled-wave shortcut (see FrameRates): the current frame's values
    float kk1, wnfit, wnmiss;
    uint32_t uni_ok;
    float Tb, TB, Tg;
    float var_init, var_min, var_max, tau;
    int nmix;
    int detect_shadows;
    int shadow_value;
};

// most frames one launch steps through (temporal blocking)
constexpr int kMaxT = 8;

// Model in HBM, np = plane pitch in floats (>= streams*h*w, a multiple of 64).  Mode m
// occupies the (2+C)*np floats from state + m*(2+C)*np:
//   [0, np)            weight plane, one float per pixel
//   [np, ...)          VECTOR plane, {variance, mean[0], ...} of a pixel side by side:
//                      float2 (C = 1) or float4 {var, mean0, mean1, mean2} (C = 3, 4)
//   [5 np, 6 np)       mean[3] plane (C = 4 only)
// so that a pixel's variance and means travel as one 8- / 16-byte access per lane.
struct Mog2State {
    float *planes;
    uint8_t *modes;
    size_t np;
};

struct Taps {
    uint16_t kx[kMaxKsize];
    uint16_t ky[kMaxKsize];
    int nkx, nky;
    // 0: 8.8 fixed-point taps, out = (sum + 2^15) >> 16.  != 0: normalised box
    // filter (all taps 1), out = rint(double(sum) * box_scale) as cv.blur does.
    double box_scale;
    // What the kernels that fuse a 3x3 blur into the detector compute (fused3_ok):
    //   out = (sum_j vky[j] * (sum_i kx[i] * p) + vround) >> 16
    // Gaussian: vky = ky, vround = 2^15.  3x3 box: kx = (1,1,1), vky = 7282 (= ceil(2^16 / 9)),
    // vround = 4 * 7282, i.e. ((S + 4) * 7282) >> 16 = floor((S + 4) / 9) = rint(S / 9) for
    // every 9-pixel sum S <= 2295 (no ties at an odd area; checked exhaustively by the host).
    uint32_t vky[3];
    uint32_t vround;
    bool fused3_ok;
};

struct SeBits {            // structuring element, bit dx of rows[dy] = member
    uint32_t rows[kMaxKsize];
    int kh, kw;
};

struct RectChain {         // fused chain of rectangular erode/dilate ops
    int n;
    int op[8];
    int rx[8], ry[8];      // half sizes (odd kernels only)
};

// images: n_img dense (h, w, c) uint8 frames back to back.
hipError_t blur_sep_q8(const uint8_t *src, uint8_t *dst, int n_img, int h, int w, int c,
                       const Taps &t, hipStream_t s);

// mask[T][n_img][h][w] <- T successive MOG2 updates of state with
// frame[T][n_img][h][w][c] (the model is loaded and stored once per launch);
// frame ti uses k.alphaT_f[ti] / k.prune_f[ti].
// fused_blur3: each frame is first blurred with the 3x3 taps t (REFLECT_101)
// inside the same kernel (LDS-staged tile), no intermediate image.
// rowflags (may be NULL): n_img * h * rowflag_pitch(w) bytes.  When the multi-frame
// kernel runs it writes, per image row and 64-px tile column, a byte whose bit t
// says "the mask of frame t is not all zero there" (conservative) and sets
// *rowflags_written; morph_rect_chain uses that to skip reading all-zero regions.
// frame_bytes: bytes readable from `frame` (>= T * n_img*h*w*c); no kernel reads a
// byte outside [frame, frame + T * n_img*h*w*c), whatever the alignment of `frame`.
inline int rowflag_pitch(int w) { return (w + 63) / 64; }
hipError_t mog2_apply(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h,
                      int w, int c, int T, const Mog2Consts &k, bool fused_blur3, const Taps &t,
                      hipStream_t s, uint8_t *rowflags, bool *rowflags_written, size_t frame_bytes);

// the multi-frame launch behind mog2_apply (gof_mog2_frames.hip); mog2_apply decides when it applies
// src_c != 0: step 0 fused -- the frames have src_c (3 | 4) channels, BGR order unless src_rgb, and are
// converted to grey inside the kernel (c == 1); then T == 1 is accepted as well
hipError_t mog2_frames_launch(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h,
                              int w, int c, int T, const Mog2Consts &k, const Taps &t, hipStream_t s,
                              uint8_t *rowflags, int src_c = 0, bool src_rgb = false);
// can mog2_convert_apply take these frames?  (3x3 byte taps, symmetric vertical multipliers,
// every staged row of every colour frame at the same offset mod 4, var_min <= var_max)
bool mog2_convert_supported(const uint8_t *frame, int n_img, int h, int w, int src_c, const Taps &t,
                            const Mog2Consts &k);
// steps 0-2 in one kernel: colour frames [T][n_img][h][w][src_c] -> grey -> 3x3 blur -> MOG2 (one-channel
// model st), T <= kMaxT
hipError_t mog2_convert_apply(const uint8_t *frame, uint8_t *mask, const Mog2State &st, int n_img, int h, int w,
                              int src_c, bool src_rgb, int T, const Mog2Consts &k, const Taps &t, hipStream_t s,
                              uint8_t *rowflags);

hipError_t fd_apply(const uint8_t *cur, const uint8_t *prev, bool has_prev, int ithr,
                    uint8_t *mask, int n_img, int h, int w, int c, hipStream_t s);

// GaussianBlur 3x3 (taps t) + frame differencing for T <= kMaxT consecutive frames
// [T][n_img][h][w][c] in one launch; the blurred last frame is handed from launch
// to launch through prev_in / prev_out (one dword per pixel, n_img*h*w of them each,
// distinct buffers; the kernels' own format).
// fd_frames_supported: 3x3 byte taps, aligned rows / frames (as the fused MOG2 path).
// src_c != 0: step 0 inside the kernel -- the frames arrive with src_c (3 | 4) channels, BGR(A) or
// RGB(A) order, and are converted to grey (c == 1) on the staged tiles.
bool fd_frames_supported(const uint8_t *frame, int n_img, int h, int w, int c, const Taps &t, int src_c = 0);
hipError_t fd_frames_apply(const uint8_t *frame, uint8_t *mask, const uint32_t *prev_in,
                           uint32_t *prev_out, bool has_prev, int ithr, int n_img, int h, int w, int c,
                           int T, const Taps &t, hipStream_t s, uint8_t *rowflags = nullptr, int src_c = 0,
                           bool src_rgb = false);

hipError_t morph_generic(int op, const uint8_t *src, uint8_t *dst, int n_img, int h, int w,
                         const SeBits &se, hipStream_t s);

// all ops rectangular with odd sizes: one kernel, intermediates stay on chip.
// chain_valued: the caller guarantees that the mask's values form a chain under
// bitwise inclusion (every mask MOG2 / FD produce does) -> AND/OR fast path.
// rowflags / streams (optional): the images are [t][stream] masks of a detector
// launch sequence whose row flags are rowflags[t / 8][stream][h][pitch], bit t % 8.
hipError_t morph_rect_chain(const uint8_t *src, uint8_t *dst, int n_img, int h, int w,
                            const RectChain &ch, bool chain_valued, hipStream_t s,
                            const uint8_t *rowflags = nullptr, int streams = 1);
bool morph_rect_chain_supported(const RectChain &ch);

// One mode between the HBM layout and structure-of-arrays planes soa[f * n + i] (f = 0
// weight, 1 variance, 2 + ch mean[ch]; n pixels): the form export / import exchange.
hipError_t mog2_mode_soa(const Mog2State &st, int c, int mode, size_t n, float *soa, bool to_soa,
                         hipStream_t s);

// (h,w,c in {3,4}) -> (h,w) luma, BGR order unless rgb
hipError_t cvt_gray(const uint8_t *src, uint8_t *dst, int n_img, int h, int w, int c, bool rgb,
                    hipStream_t s);

// ---- K-nearest-neighbours subtractor (gof_knn.hip) ---------------------------------------
// Model in HBM, pitch np >= n_img * h * w dwords: samples = 3 nN sample planes (x 2 for 4-channel
// frames): a pixel's sample is ONE dword (channel bytes + the flag in byte 3); next / index: a dword
// per pixel, bytes (short, mid, long).
struct KnnState {
    uint32_t *samples, *next, *index;
    size_t np;
    int nN, knn;
    float Tb, tau;
    int detect_shadows, shadow_value;
};
// per-frame scalars, computed on the host as cv2 does (gof_api.cpp knn_next_frame)
struct KnnFrame {
    int short_counter, mid_counter, long_counter;   // counters of this frame
    int n_short, n_mid, n_long;                     // cycle lengths of this frame
    uint32_t frame_no;                              // 1-based frame count: key of the update draws
};
// blur3 != NULL: the frames are blurred 3x3 (GaussianBlur / cv.blur, Taps::fused3_ok) inside the kernel;
// the caller checks knn_blur3_supported first
hipError_t knn_apply(const uint8_t *frame, uint8_t *mask, const KnnState &st, int n_img, int h, int w, int c,
                     const KnnFrame &f, hipStream_t s, const Taps *blur3 = nullptr);
bool knn_blur3_supported(const uint8_t *frame, int n_img, int h, int w, int c, const Taps &t);
// byte planes for parity tests: d_samples[3 nN][c + 1][n], d_state[6][n] (device buffers)
hipError_t knn_export(const KnnState &st, int c, size_t n, uint8_t *d_samples, uint8_t *d_state, hipStream_t s);

// ---- blob extraction (gof_blobs.hip) ----------------------------------------------------
// One entry per border (crack cycle): exact shoelace sums of the border polygon, pixel
// changes along it (= its number of contour points), bounding box.
struct BlobAcc {
    long long a00, a10, a01;   // sum cross, sum cross*(x0+x1), sum cross*(y0+y1) over the border's edges
    int xmin, ymin, xmax, ymax;
    int trans;                 // pixel changes around the cycle
    uint32_t root;             // slot of the root crack (north crack of the raster-first pixel)
    int out;                   // index of the blob record, -1 if the border is not returned
    int pad_;
};
// Per-image output block, packed: header, n_blobs records, n_points points (x | y << 16).
struct BlobHeader {
    uint32_t n_blobs, n_points, flags, n_cracks;
};
struct BlobRecord {            // == gof_blob_record (include/gofpid_hip.h)
    int x0, y0;                // start pixel: the border's first point
    int xmin, ymin, xmax, ymax;
    int n_points, point_offset;
    long long a00, a10, a01;
};
constexpr uint32_t kBlobHeaderWords = sizeof(BlobHeader) / 4, kBlobRecordWords = sizeof(BlobRecord) / 4;
// 0..3 counts / flags (zero between extractions: the finishing kernel clears them behind itself), 4..11 phase
// stamps of the finishing kernel (100 MHz ticks), 12..15 the counts / flags of the last extraction (diagnostics)
constexpr int kBlobCounters = 16;
enum { kCntCracks = 0, kCntRoots = 1, kCntFlags = 2, kCntSegs = 3, kCntLast = 12 };
enum { kFlagCrackOverflow = 1, kFlagBlobOverflow = 2, kFlagPointOverflow = 4, kFlagInternal = 8 };

struct BlobBuffers {           // device scratch for n_img masks of h x w; cap = crack slots per image
    int n_img, h, w;
    uint32_t cap;              // multiple of 4
    uint32_t out_words;        // dwords of one image's output block
    uint32_t *cbase;           // [n_img][h*w]  slot << 4 | side bits, for pixels that own cracks
    uint32_t *ckey;            // [n_img][cap]  pixel * 4 + side
    uint32_t *csucc, *cparent, *caux, *cpos, *cslot;   // [n_img][cap] each
    uint32_t *rsel, *rnxt, *rsum, *rnxt2, *rsum2;   // [n_img][cap] each: list-ranking scratch
    BlobAcc *acc;              // [n_img][cap / 4]
    uint32_t *segs;            // [n_img][seg_cap]  tiles (256 px x 16 rows) with foreground: ty << 16 | tx
    uint32_t seg_cap;          // >= ceil(h / 16) * ceil(w / 256)
    uint32_t *counters;        // [n_img][kBlobCounters]; all zero before an extraction (blob_extract's contract)
    uint32_t *out;             // [n_img][out_words]
    // host-visible (pinned, device-mapped) mirror of the head of every image's output block, written by the
    // finishing kernel itself: no copy operation behind the kernels.  NULL: no mirror.
    uint32_t *host_out;        // [n_img][host_words]
    uint32_t host_words;
};

// External contours (RETR_EXTERNAL, CHAIN_APPROX_NONE) of n_img masks with polygon area
// >= min_area: records + ordered points into B.out, one block per image.  Enqueues only.
// B.counters must be zero on entry (they are after a completed extraction; after a failed launch the
// caller clears them).
hipError_t blob_extract(const uint8_t *mask, const BlobBuffers &B, double min_area, hipStream_t s);

// sum of modes[] (for m-bar); out is a device uint64 accumulator, zeroed by caller.
hipError_t sum_modes(const uint8_t *modes, size_t n, unsigned long long *out, hipStream_t s);

}  // namespace gofk
