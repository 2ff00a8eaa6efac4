// C ABI of libgofpid_hip.so (include/gofpid_hip.h): contexts, the MOG2 / FD
// state objects, the fused pipeline object and the host-buffer conveniences.
// All arithmetic is in gof_kernels.hip; this file only owns memory, computes
// the per-frame scalars cv2 computes on the host, and validates shapes before
// any launch.
#include "gofpid_hip.h"

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <cmath>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "gof_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define GOF_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(e_ == hipErrorOutOfMemory ? GOF_ENOMEM                           \
                        : e_ == hipErrorInvalidValue ? GOF_EINVAL : GOF_EHIP,            \
                        "%s: %s", #expr, hipGetErrorString(e_));                         \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap)
            return GOF_OK;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        // + 256: dword-granular tile loads may touch the dword holding the last byte
        GOF_HIP(hipMalloc(&p, bytes + 256));
        cap = bytes;
        return GOF_OK;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

bool valid_channels(int c) { return c == 1 || c == 3 || c == 4; }

}  // namespace

struct gof_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf in, out, tmp0, tmp1;   // scratch of the stateless host entry points
};

struct gof_mog2 {
    gof_ctx *ctx = nullptr;
    gof_mog2_params p{};
    int streams = 0, h = 0, w = 0, c = 0;
    int nframes = 0;
    size_t np = 0;           // plane pitch in floats
    size_t state_bytes = 0;
    void *state = nullptr;   // [K*(2+c)] float planes of pitch np, then modes[np]
    // the stream the last apply was enqueued on: export / import / state_copy /
    // mean_modes run on ctx->stream and wait for it first when it is another one
    hipStream_t last = nullptr;
    DevBuf in, out;
    DevBuf soa;              // one mode as plain planes: the staging form of export / import
    gofk::Mog2State view() const
    {
        gofk::Mog2State s;
        s.planes = (float *)state;
        s.modes = (uint8_t *)state + (size_t)p.nmixtures * (2 + c) * np * sizeof(float);
        s.np = np;
        return s;
    }
};

struct gof_knn {
    gof_ctx *ctx = nullptr;
    gof_knn_params p{};
    int streams = 0, h = 0, w = 0, c = 0;
    int nframes = 0;
    int short_counter = 0, mid_counter = 0, long_counter = 0;   // cv2's nShortCounter ... (class members)
    size_t np = 0;            // plane pitch in dwords
    size_t state_bytes = 0;
    void *state = nullptr;    // [3 nN * ws] sample planes, then the next plane, then the index plane
    hipStream_t last = nullptr;
    DevBuf in, out, exp;
    int ws() const { return c <= 3 ? 1 : 2; }
    gofk::KnnState view() const
    {
        gofk::KnnState s;
        s.samples = (uint32_t *)state;
        s.next = s.samples + (size_t)3 * p.nsamples * ws() * np;
        s.index = s.next + np;
        s.np = np;
        s.nN = p.nsamples;
        s.knn = p.knn;
        s.Tb = p.dist2_threshold;
        s.tau = p.tau;
        s.detect_shadows = p.detect_shadows;
        s.shadow_value = p.shadow_value;
        return s;
    }
};

struct gof_fd {
    gof_ctx *ctx = nullptr;
    int streams = 0, h = 0, w = 0, c = 0;
    double threshold = 5;
    bool has_prev = false;
    int cur = 0;
    DevBuf frame[2];         // ping-pong: frame[cur] is X_t, frame[cur^1] is X_{t-1}
    DevBuf out;
};

struct gof_pipeline {
    gof_ctx *ctx = nullptr;
    gof_pipeline_desc d{};
    std::vector<std::vector<uint8_t>> se;   // owned copies of the structuring elements
    gofk::Taps taps{};
    bool fuse_blur3 = false;   // MOG2: the 3x3 blur runs inside the detector kernel
    bool fuse_fd3 = false;     // FD: likewise, when the frames are aligned (checked per call)
    bool chain_ok = false;
    gofk::RectChain chain{};
    std::vector<gofk::SeBits> sebits;
    gof_mog2 *mog2 = nullptr;
    gof_fd *fd = nullptr;
    gof_knn *knn = nullptr;
    DevBuf in, out, blurred, m0, m1;
    // Host-frame batches are double-buffered (slot 0 = in / out, slot 1 = in1 / out1): the frames
    // of one batch go up on a copy stream while the kernels of the batch before run on the
    // context's stream (gof_pipeline_submit_host_blobs_ptrs / gof_pipeline_wait_slot).
    DevBuf in1, out1;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    gof_blobs *slot_blobs[2] = {nullptr, nullptr};
    int last_slot = 0;         // the slot whose `out` holds the masks of the last host batch
    DevBuf gray;       // step 0 (gof_pipeline_set_convert): the converted frames
    int cvt_c = 0;     // channels of the incoming frames when the conversion is on (else 0)
    bool cvt_rgb = false;
    DevBuf rowflags;   // per launch chunk: [stream][h][tile column] bytes, see gofk::mog2_apply
    // fused blur + FD: the blurred last frame of the previous launch, one dword per
    // pixel (ping-pong), and which path this pipeline took on its first call
    DevBuf fd_prev[2];
    int fd_cur = 0;
    bool fd_has_prev = false;
    int fd_mode = 0;   // 0 undecided, 1 fused kernel, 2 separate blur / difference kernels
    unsigned long long *d_sum = nullptr;
    // profiling: events[3*i+0..2] = before detector, after detector, after morphology
    bool prof = false;
    std::vector<hipEvent_t> events;
    size_t prof_used = 0;
    int prof_stride = 1;       // bracket every prof_stride-th apply
    size_t prof_seen = 0;      // applies since profiling was enabled
};

struct gof_blobs {
    gof_ctx *ctx = nullptr;
    int n_img = 0, h = 0, w = 0;
    gofk::BlobBuffers B{};
    void *dev = nullptr;            // one allocation behind every pointer of B
    uint32_t *host = nullptr;       // pinned: [n_img][head_words], the head of every image's block
    uint32_t head_words = 0;
    std::vector<uint32_t> spill;    // an image's whole block when it exceeds the head
    int spill_image = -1;
    hipStream_t last = nullptr;
    bool pending = false;
    bool dirty = false;             // a launch failed mid-extraction: the counters need clearing first
    int n_last = 0;
    DevBuf in;                      // masks of gof_blobs_extract_host
};

namespace {

int make_taps(int kw, int kh, double sx, double sy, gofk::Taps *t)
{
    if (kw < 1 || kh < 1 || !(kw & 1) || !(kh & 1) || kw > GOF_MAX_KSIZE || kh > GOF_MAX_KSIZE)
        return fail(GOF_EINVAL, "Gaussian ksize must be odd and in [1, %d], got (%d, %d)",
                    GOF_MAX_KSIZE, kw, kh);
    if (sy <= 0)
        sy = sx;
    if (gof_gauss_kernel_q8(kw, sx, t->kx) || gof_gauss_kernel_q8(kh, sy, t->ky))
        return fail(GOF_EINVAL, "bad Gaussian kernel");
    t->nkx = kw;
    t->nky = kh;
    t->box_scale = 0.0;
    // the detector kernels can take the 3x3 form themselves when the horizontal taps are bytes
    // (a degenerate sigma gives [0, 256, 0], which goes through the generic blur instead)
    t->fused3_ok = kw == 3 && kh == 3 && t->kx[0] < 256 && t->kx[1] < 256 && t->kx[2] < 256;
    for (int j = 0; j < 3; j++)
        t->vky[j] = kh == 3 ? t->ky[j] : 0;
    t->vround = 32768u;
    return GOF_OK;
}

// cv.blur(src, (kw, kh)): normalised box filter, REFLECT_101 [OPENCV-RECALL]
int make_box(int kw, int kh, gofk::Taps *t)
{
    if (kw < 1 || kh < 1 || !(kw & 1) || !(kh & 1) || kw > GOF_MAX_KSIZE || kh > GOF_MAX_KSIZE)
        return fail(GOF_EINVAL, "box ksize must be odd and in [1, %d], got (%d, %d)",
                    GOF_MAX_KSIZE, kw, kh);
    for (int i = 0; i < gofk::kMaxKsize; i++)
        t->kx[i] = t->ky[i] = 1;
    t->nkx = kw;
    t->nky = kh;
    t->box_scale = 1.0 / ((double)kw * (double)kh);
    // 3x3: the fused form ((S + 4) * 7282) >> 16 -- accepted only if it reproduces
    // rint(double(S) * box_scale) for every possible sum (it does; the loop is the proof)
    t->fused3_ok = kw == 3 && kh == 3;
    t->vky[0] = t->vky[1] = t->vky[2] = 7282u;
    t->vround = 4u * 7282u;
    for (uint32_t S = 0; t->fused3_ok && S <= 9u * 255u; S++)
        if (((S * 7282u + t->vround) >> 16) != (uint32_t)std::nearbyint((double)S * t->box_scale))
            t->fused3_ok = false;
    return GOF_OK;
}

int make_sebits(const uint8_t *k, int kh, int kw, gofk::SeBits *se)
{
    if (kh < 1 || kw < 1 || kh > GOF_MAX_KSIZE || kw > GOF_MAX_KSIZE)
        return fail(GOF_EINVAL, "structuring element must be within %dx%d, got (%d, %d)",
                    GOF_MAX_KSIZE, GOF_MAX_KSIZE, kh, kw);
    memset(se, 0, sizeof(*se));
    se->kh = kh;
    se->kw = kw;
    for (int y = 0; y < kh; y++)
        for (int x = 0; x < kw; x++)
            if (!k || k[y * kw + x])
                se->rows[y] |= 1u << x;
    return GOF_OK;
}

bool se_is_full_odd_rect(const gofk::SeBits &se)
{
    if (!(se.kh & 1) || !(se.kw & 1))
        return false;
    uint32_t full = se.kw == 32 ? 0xffffffffu : ((1u << se.kw) - 1);
    for (int y = 0; y < se.kh; y++)
        if (se.rows[y] != full)
            return false;
    return true;
}

int check_geom(int n, int h, int w, int c)
{
    if (n < 1 || h < 1 || w < 1 || !valid_channels(c))
        return fail(GOF_EINVAL, "bad geometry: images=%d h=%d w=%d c=%d (c must be 1, 3 or 4)", n,
                    h, w, c);
    if ((double)n * h * w * 4.0 > 1.0e12)
        return fail(GOF_EINVAL, "geometry too large");
    return GOF_OK;
}

hipStream_t pick(gof_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }

// Host-visible accesses to the model (export, import, state_copy, mean_modes) run on
// the context's stream; applies may have been enqueued on a caller's stream.
hipError_t drain_applies(gof_mog2 *m)
{
    if (m->last && m->last != m->ctx->stream)
        return hipStreamSynchronize(m->last);
    return hipSuccess;
}

// Constants of the next T (<= kMaxT) frames; advances the frame counter by T.
// cv2 BackgroundSubtractorMOG2Impl::apply with learningRate = -1 [OPENCV-RECALL]:
//   ++nframes; lr = 1/min(2*nframes, history) in double;
//   alphaT = float(lr); prune = float(-lr * fCT)
gofk::Mog2Consts next_consts(gof_mog2 *m, int T)
{
    gofk::Mog2Consts k;
    k.uni_mask = 0u;
    for (int i = 0; i < gofk::kMaxT; i++) {
        k.alphaT_f[i] = 0.f;
        k.prune_f[i] = 0.f;
        k.fr[i] = gofk::FrameRates{0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
    }
    for (int i = 0; i < T; i++) {
        ++m->nframes;
        const int d = 2 * m->nframes < m->p.history ? 2 * m->nframes : m->p.history;
        const double lr = 1.0 / (double)d;
        k.alphaT_f[i] = (float)lr;
        k.prune_f[i] = (float)(-lr * m->p.ct);
        // the update of a lone mode of weight 1.0f, statement for statement as
        // mog2_update_single evaluates it (float arithmetic, no contraction)
        const float aT = k.alphaT_f[i], pr = k.prune_f[i];
        const float a1 = 1.f - aT;
        const float weight1 = a1 * 1.f + pr;
        const float wfit1 = weight1 + aT;
        const float tfit = 0.f + wfit1, tmiss = 0.f + weight1;
        // (the settled path orders Tg / Tb / TB by their bit patterns: finite values only)
        const bool ok = std::isfinite(m->p.var_threshold) && std::isfinite(m->p.var_threshold_gen) &&
                        std::isfinite(m->p.background_ratio) && std::isfinite(m->p.var_init) &&
                        m->p.var_min <= m->p.var_max && std::isnormal(wfit1) && std::isnormal(weight1) && !(wfit1 < -pr) &&
                        !(weight1 < -pr) && std::fabs(tfit) > FLT_EPSILON &&
                        std::fabs(tmiss) > FLT_EPSILON;
        gofk::FrameRates &f = k.fr[i];
        f.alphaT = aT;
        f.alpha1 = a1;
        f.prune = pr;
        if (ok) {
            f.kk1 = aT / wfit1;
            f.wnfit = wfit1 * (1.f / tfit);
            f.wnmiss = weight1 * (1.f / tmiss);
            f.uni_ok = 1u;
            k.uni_mask |= 1u << i;
        }
    }
    k.alphaT = k.alphaT_f[0];
    k.alpha1 = 1.f - k.alphaT;
    k.prune = k.prune_f[0];
    k.kk1 = k.fr[0].kk1;
    k.wnfit = k.fr[0].wnfit;
    k.wnmiss = k.fr[0].wnmiss;
    k.uni_ok = k.fr[0].uni_ok;
    k.Tb = m->p.var_threshold;
    k.TB = m->p.background_ratio;
    k.Tg = m->p.var_threshold_gen;
    k.var_init = m->p.var_init;
    k.var_min = m->p.var_min;
    k.var_max = m->p.var_max;
    k.tau = m->p.tau;
    k.nmix = m->p.nmixtures;
    k.detect_shadows = m->p.detect_shadows;
    k.shadow_value = m->p.shadow_value;
    return k;
}

}  // namespace

extern "C" {

int gof_abi_version(void) { return GOF_ABI_VERSION; }

const char *gof_strerror(int code)
{
    switch (code) {
    case GOF_OK: return "ok";
    case GOF_EINVAL: return "invalid argument";
    case GOF_EHIP: return "HIP runtime error";
    case GOF_ENOMEM: return "out of memory";
    case GOF_ENODEV: return "no usable device";
    default: return "unknown error";
    }
}

const char *gof_last_error(void) { return g_err; }

int gof_device_count(int *count)
{
    if (!count)
        return fail(GOF_EINVAL, "count is NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(GOF_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return GOF_OK;
}

int gof_create(int device, gof_ctx **out)
{
    if (!out)
        return fail(GOF_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(GOF_ENODEV, "no HIP device visible");
    if (device < 0 || device >= n)
        return fail(GOF_EINVAL, "device %d out of range [0, %d)", device, n);
    GOF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GOF_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GOF_ENODEV, "device %d is %s; this library carries gfx950 code only", device,
                    prop.gcnArchName);
    gof_ctx *c = new (std::nothrow) gof_ctx;
    if (!c)
        return fail(GOF_ENOMEM, "host allocation failed");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(GOF_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return GOF_OK;
}

void gof_destroy(gof_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->in.release();
    ctx->out.release();
    ctx->tmp0.release();
    ctx->tmp1.release();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int gof_synchronize(gof_ctx *ctx)
{
    if (!ctx)
        return fail(GOF_EINVAL, "ctx is NULL");
    GOF_HIP(hipStreamSynchronize(ctx->stream));
    return GOF_OK;
}

void *gof_stream(gof_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

/* ---- blur ---------------------------------------------------------------- */

int gof_gauss_kernel_q8(int n, double sigma, uint16_t *taps)
{
    // cv2 getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED [OPENCV-RECALL],
    // side taps rounded to 1/256 with the error carried forward,
    // centre tap = 256 - 2*sum(side); sigma <= 0 uses cv2's small fixed tables.
    if (n < 1 || !(n & 1) || n > 255 || !taps)
        return GOF_EINVAL;
    if (sigma <= 0) {
        static const uint16_t t1[] = {256}, t3[] = {64, 128, 64}, t5[] = {16, 64, 96, 64, 16},
                              t7[] = {8, 28, 56, 72, 56, 28, 8},
                              t9[] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
        const uint16_t *t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : n == 9 ? t9 : nullptr;
        if (t) {
            memcpy(taps, t, (size_t)n * sizeof(uint16_t));
            return GOF_OK;
        }
        sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    }
    const int half = n / 2;
    std::vector<double> g((size_t)n);
    double sum = 0;
    const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        double x = i - half;
        g[(size_t)i] = exp(s2 * x * x);
        sum += g[(size_t)i];
    }
    double err = 0;
    long acc = 0;
    for (int i = 0; i < half; i++) {
        double adj = g[(size_t)i] / sum * 256.0 + err;
        long v = lrint(adj);
        err = adj - (double)v;
        taps[i] = taps[n - 1 - i] = (uint16_t)v;
        acc += 2 * v;
    }
    taps[half] = (uint16_t)(256 - acc);
    return GOF_OK;
}

int gof_gaussian_blur_dev(gof_ctx *ctx, const void *d_src, void *d_dst, int n_images, int h, int w,
                          int c, int kw, int kh, double sigma_x, double sigma_y, void *stream)
{
    if (!ctx || !d_src || !d_dst)
        return fail(GOF_EINVAL, "NULL argument");
    if (d_src == d_dst)
        return fail(GOF_EINVAL, "blur cannot run in place");
    int rc = check_geom(n_images, h, w, c);
    if (rc)
        return rc;
    gofk::Taps t;
    if ((rc = make_taps(kw, kh, sigma_x, sigma_y, &t)))
        return rc;
    GOF_HIP(hipSetDevice(ctx->device));
    GOF_HIP(gofk::blur_sep_q8((const uint8_t *)d_src, (uint8_t *)d_dst, n_images, h, w, c, t,
                              pick(ctx, stream)));
    return GOF_OK;
}

int gof_box_blur_dev(gof_ctx *ctx, const void *d_src, void *d_dst, int n_images, int h, int w,
                     int c, int kw, int kh, void *stream)
{
    if (!ctx || !d_src || !d_dst)
        return fail(GOF_EINVAL, "NULL argument");
    if (d_src == d_dst)
        return fail(GOF_EINVAL, "blur cannot run in place");
    int rc = check_geom(n_images, h, w, c);
    if (rc)
        return rc;
    gofk::Taps t;
    if ((rc = make_box(kw, kh, &t)))
        return rc;
    GOF_HIP(hipSetDevice(ctx->device));
    GOF_HIP(gofk::blur_sep_q8((const uint8_t *)d_src, (uint8_t *)d_dst, n_images, h, w, c, t,
                              pick(ctx, stream)));
    return GOF_OK;
}

int gof_box_blur_host(gof_ctx *ctx, const uint8_t *src, uint8_t *dst, int h, int w, int c, int kw,
                      int kh)
{
    if (!ctx || !src || !dst)
        return fail(GOF_EINVAL, "NULL argument");
    int rc = check_geom(1, h, w, c);
    if (rc)
        return rc;
    size_t bytes = (size_t)h * w * c;
    GOF_HIP(hipSetDevice(ctx->device));
    if ((rc = ctx->in.reserve(bytes)) || (rc = ctx->out.reserve(bytes)))
        return rc;
    GOF_HIP(hipMemcpyAsync(ctx->in.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gof_box_blur_dev(ctx, ctx->in.p, ctx->out.p, 1, h, w, c, kw, kh, nullptr)))
        return rc;
    GOF_HIP(hipMemcpyAsync(dst, ctx->out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GOF_HIP(hipStreamSynchronize(ctx->stream));
    return GOF_OK;
}

int gof_gaussian_blur_host(gof_ctx *ctx, const uint8_t *src, uint8_t *dst, int h, int w, int c,
                           int kw, int kh, double sigma_x, double sigma_y)
{
    if (!ctx || !src || !dst)
        return fail(GOF_EINVAL, "NULL argument");
    int rc = check_geom(1, h, w, c);
    if (rc)
        return rc;
    size_t bytes = (size_t)h * w * c;
    GOF_HIP(hipSetDevice(ctx->device));
    if ((rc = ctx->in.reserve(bytes)) || (rc = ctx->out.reserve(bytes)))
        return rc;
    GOF_HIP(hipMemcpyAsync(ctx->in.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = gof_gaussian_blur_dev(ctx, ctx->in.p, ctx->out.p, 1, h, w, c, kw, kh, sigma_x, sigma_y,
                               nullptr);
    if (rc)
        return rc;
    GOF_HIP(hipMemcpyAsync(dst, ctx->out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GOF_HIP(hipStreamSynchronize(ctx->stream));
    return GOF_OK;
}

/* ---- colour conversion --------------------------------------------------- */

int gof_cvt_gray_dev(gof_ctx *ctx, const void *d_src, void *d_dst, int n_images, int h, int w,
                     int c, int rgb_order, void *stream)
{
    if (!ctx || !d_src || !d_dst)
        return fail(GOF_EINVAL, "NULL argument");
    if (n_images < 1 || h < 1 || w < 1 || (c != 3 && c != 4))
        return fail(GOF_EINVAL, "cvtColor to gray needs 3 or 4 channels, got images=%d h=%d w=%d c=%d",
                    n_images, h, w, c);
    GOF_HIP(hipSetDevice(ctx->device));
    GOF_HIP(gofk::cvt_gray((const uint8_t *)d_src, (uint8_t *)d_dst, n_images, h, w, c,
                           rgb_order != 0, pick(ctx, stream)));
    return GOF_OK;
}

int gof_cvt_gray_host(gof_ctx *ctx, const uint8_t *src, uint8_t *dst, int h, int w, int c,
                      int rgb_order)
{
    if (!ctx || !src || !dst)
        return fail(GOF_EINVAL, "NULL argument");
    if (h < 1 || w < 1 || (c != 3 && c != 4))
        return fail(GOF_EINVAL, "cvtColor to gray needs 3 or 4 channels");
    size_t npx = (size_t)h * w;
    int rc;
    GOF_HIP(hipSetDevice(ctx->device));
    if ((rc = ctx->in.reserve(npx * c)) || (rc = ctx->out.reserve(npx)))
        return rc;
    GOF_HIP(hipMemcpyAsync(ctx->in.p, src, npx * c, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gof_cvt_gray_dev(ctx, ctx->in.p, ctx->out.p, 1, h, w, c, rgb_order, nullptr)))
        return rc;
    GOF_HIP(hipMemcpyAsync(dst, ctx->out.p, npx, hipMemcpyDeviceToHost, ctx->stream));
    GOF_HIP(hipStreamSynchronize(ctx->stream));
    return GOF_OK;
}

/* ---- MOG2 ---------------------------------------------------------------- */

void gof_mog2_default_params(gof_mog2_params *p)
{
    if (!p)
        return;
    p->history = 500;
    p->nmixtures = 5;
    p->var_threshold = 16.f;
    p->background_ratio = 0.9f;
    p->var_threshold_gen = 9.f;
    p->var_init = 15.f;
    p->var_min = 4.f;
    p->var_max = 75.f;
    p->ct = 0.05f;
    p->tau = 0.5f;
    p->detect_shadows = 1;
    p->shadow_value = 127;
}

int gof_mog2_create(gof_ctx *ctx, int streams, int h, int w, int c, const gof_mog2_params *p,
                    gof_mog2 **out)
{
    if (!ctx || !out)
        return fail(GOF_EINVAL, "NULL argument");
    *out = nullptr;
    int rc = check_geom(streams, h, w, c);
    if (rc)
        return rc;
    gof_mog2_params pp;
    if (p)
        pp = *p;
    else
        gof_mog2_default_params(&pp);
    if (pp.nmixtures < 1 || pp.nmixtures > GOF_MAX_MODES || pp.history < 1 ||
        pp.shadow_value < 0 || pp.shadow_value > 255)
        return fail(GOF_EINVAL, "bad MOG2 parameters (nmixtures in [1,%d], history >= 1)",
                    GOF_MAX_MODES);
    gof_mog2 *m = new (std::nothrow) gof_mog2;
    if (!m)
        return fail(GOF_ENOMEM, "host allocation failed");
    m->ctx = ctx;
    m->p = pp;
    m->streams = streams;
    m->h = h;
    m->w = w;
    m->c = c;
    size_t n = (size_t)streams * h * w;
    m->np = (n + 63) & ~(size_t)63;   // planes start on 256-B boundaries
    m->state_bytes = (size_t)pp.nmixtures * (2 + c) * m->np * sizeof(float) + m->np;
    GOF_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&m->state, m->state_bytes);
    if (e != hipSuccess) {
        delete m;
        return fail(GOF_ENOMEM, "hipMalloc(%zu B of MOG2 state): %s", m->state_bytes,
                    hipGetErrorString(e));
    }
    // Zero-fill and WAIT: the *_dev entry points may be given any caller stream, which is
    // not ordered against ctx->stream (created non-blocking); the model must be complete
    // before create returns.  Not on the hot path.
    e = hipMemsetAsync(m->state, 0, m->state_bytes, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(m->state);
        delete m;
        return fail(GOF_EHIP, "zero-filling the MOG2 state: %s", hipGetErrorString(e));
    }
    *out = m;
    return GOF_OK;
}

void gof_mog2_destroy(gof_mog2 *m)
{
    if (!m)
        return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->state)
        (void)hipFree(m->state);
    m->in.release();
    m->out.release();
    m->soa.release();
    delete m;
}

int gof_mog2_nframes(const gof_mog2 *m) { return m ? m->nframes : -1; }

int gof_mog2_set_nframes(gof_mog2 *m, int nframes)
{
    if (!m || nframes < 0)
        return fail(GOF_EINVAL, "bad argument");
    m->nframes = nframes;
    return GOF_OK;
}

int gof_mog2_apply_dev(gof_mog2 *m, const void *d_frames, void *d_masks, void *stream)
{
    if (!m || !d_frames || !d_masks)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(m->ctx->device));
    gofk::Mog2Consts k = next_consts(m, 1);
    gofk::Taps none{};
    m->last = pick(m->ctx, stream);
    hipError_t e = gofk::mog2_apply((const uint8_t *)d_frames, (uint8_t *)d_masks, m->view(),
                                    m->streams, m->h, m->w, m->c, 1, k, false, none,
                                    m->last, nullptr, nullptr, (size_t)m->streams * m->h * m->w * m->c);
    if (e != hipSuccess) {
        --m->nframes;
        return fail(GOF_EHIP, "mog2 launch: %s", hipGetErrorString(e));
    }
    return GOF_OK;
}

int gof_mog2_apply_host(gof_mog2 *m, const uint8_t *frames, uint8_t *masks)
{
    if (!m || !frames || !masks)
        return fail(GOF_EINVAL, "NULL argument");
    size_t npx = (size_t)m->streams * m->h * m->w;
    int rc;
    GOF_HIP(hipSetDevice(m->ctx->device));
    if ((rc = m->in.reserve(npx * m->c)) || (rc = m->out.reserve(npx)))
        return rc;
    hipStream_t s = m->ctx->stream;
    GOF_HIP(hipMemcpyAsync(m->in.p, frames, npx * m->c, hipMemcpyHostToDevice, s));
    if ((rc = gof_mog2_apply_dev(m, m->in.p, m->out.p, nullptr)))
        return rc;
    GOF_HIP(hipMemcpyAsync(masks, m->out.p, npx, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    return GOF_OK;
}

int gof_mog2_export(gof_mog2 *m, float *weight, float *var, float *mean, uint8_t *modes)
{
    if (!m || !weight || !var || !mean || !modes)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(m->ctx->device));
    GOF_HIP(drain_applies(m));
    hipStream_t s = m->ctx->stream;
    const size_t n = (size_t)m->streams * m->h * m->w, np = m->np;
    const int K = m->p.nmixtures, C = m->c, F = 2 + C;
    // the device keeps {variance, means} interleaved per pixel (gof_kernels.h): one mode
    // at a time is spread into plain planes on the device and copied out plane by plane
    int rc;
    if ((rc = m->soa.reserve((size_t)F * n * sizeof(float))))
        return rc;
    float *soa = (float *)m->soa.p;
    (void)np;
    for (int k = 0; k < K; k++) {
        GOF_HIP(gofk::mog2_mode_soa(m->view(), C, k, n, soa, true, s));
        GOF_HIP(hipMemcpyAsync(weight + (size_t)k * n, soa, n * sizeof(float), hipMemcpyDeviceToHost, s));
        GOF_HIP(hipMemcpyAsync(var + (size_t)k * n, soa + n, n * sizeof(float), hipMemcpyDeviceToHost, s));
        GOF_HIP(hipMemcpyAsync(mean + (size_t)k * C * n, soa + 2 * n, (size_t)C * n * sizeof(float),
                               hipMemcpyDeviceToHost, s));
        GOF_HIP(hipStreamSynchronize(s));   // the staging planes are reused by the next mode
    }
    GOF_HIP(hipMemcpyAsync(modes, m->view().modes, n, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; i++)   // bit 7 is the kernels' "first weight is 1.0f" note
        modes[i] &= 0x7f;
    return GOF_OK;
}

int gof_mog2_import(gof_mog2 *m, const float *weight, const float *var, const float *mean,
                    const uint8_t *modes, int nframes)
{
    if (!m || !weight || !var || !mean || !modes || nframes < 0)
        return fail(GOF_EINVAL, "bad argument");
    GOF_HIP(hipSetDevice(m->ctx->device));
    GOF_HIP(drain_applies(m));
    hipStream_t s = m->ctx->stream;
    const size_t n = (size_t)m->streams * m->h * m->w, np = m->np;
    const int K = m->p.nmixtures, C = m->c, F = 2 + C;
    for (size_t i = 0; i < n; i++)
        if (modes[i] > K)
            return fail(GOF_EINVAL, "modes[%zu] = %d exceeds nmixtures %d", i, modes[i], K);
    int rc;
    if ((rc = m->soa.reserve((size_t)F * n * sizeof(float))))
        return rc;
    float *soa = (float *)m->soa.p;
    (void)np;
    for (int k = 0; k < K; k++) {
        GOF_HIP(hipMemcpyAsync(soa, weight + (size_t)k * n, n * sizeof(float), hipMemcpyHostToDevice, s));
        GOF_HIP(hipMemcpyAsync(soa + n, var + (size_t)k * n, n * sizeof(float), hipMemcpyHostToDevice, s));
        GOF_HIP(hipMemcpyAsync(soa + 2 * n, mean + (size_t)k * C * n, (size_t)C * n * sizeof(float),
                               hipMemcpyHostToDevice, s));
        GOF_HIP(gofk::mog2_mode_soa(m->view(), C, k, n, soa, false, s));
        GOF_HIP(hipStreamSynchronize(s));
    }
    GOF_HIP(hipMemcpyAsync(m->view().modes, modes, n, hipMemcpyHostToDevice, s));
    GOF_HIP(hipStreamSynchronize(s));
    m->nframes = nframes;
    return GOF_OK;
}

int gof_mog2_device_state(gof_mog2 *m, void **d_state, size_t *bytes)
{
    if (!m || !d_state || !bytes)
        return fail(GOF_EINVAL, "NULL argument");
    *d_state = m->state;
    *bytes = m->state_bytes;
    return GOF_OK;
}

int gof_mog2_state_copy(gof_mog2 *m, void *d_buf, int to_model)
{
    if (!m || !d_buf)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(m->ctx->device));
    GOF_HIP(drain_applies(m));
    hipStream_t s = m->ctx->stream;
    if (to_model)
        GOF_HIP(hipMemcpyAsync(m->state, d_buf, m->state_bytes, hipMemcpyDeviceToDevice, s));
    else
        GOF_HIP(hipMemcpyAsync(d_buf, m->state, m->state_bytes, hipMemcpyDeviceToDevice, s));
    GOF_HIP(hipStreamSynchronize(s));
    return GOF_OK;
}

/* ---- frame differencing -------------------------------------------------- */

// ---- KNN (pipeline.py:233-234) -------------------------------------------------------------
void gof_knn_default_params(gof_knn_params *p)
{
    if (!p)
        return;
    p->history = 500;
    p->nsamples = 7;
    p->dist2_threshold = 20.0f * 20.0f;
    p->knn = 2;   // cv2: max(1, cvRound(0.1 * nN * 3 + 0.40))
    p->tau = 0.5f;
    p->detect_shadows = 1;
    p->shadow_value = 127;
}

int gof_knn_create(gof_ctx *ctx, int streams, int h, int w, int c, const gof_knn_params *p, gof_knn **out)
{
    if (!ctx || !out)
        return fail(GOF_EINVAL, "NULL argument");
    *out = nullptr;
    int rc = check_geom(streams, h, w, c);
    if (rc)
        return rc;
    gof_knn_params pp;
    if (p)
        pp = *p;
    else
        gof_knn_default_params(&pp);
    if (pp.nsamples < 1 || pp.nsamples > 16 || pp.knn < 1 || pp.history < 1 || pp.shadow_value < 0 ||
        pp.shadow_value > 255)
        return fail(GOF_EINVAL, "bad KNN parameters (nsamples in [1,16], knn >= 1, history >= 1)");
    gof_knn *k = new (std::nothrow) gof_knn;
    if (!k)
        return fail(GOF_ENOMEM, "host allocation failed");
    k->ctx = ctx;
    k->p = pp;
    k->streams = streams;
    k->h = h;
    k->w = w;
    k->c = c;
    const size_t n = (size_t)streams * h * w;
    k->np = (n + 63) & ~(size_t)63;
    k->state_bytes = ((size_t)3 * pp.nsamples * k->ws() + 2) * k->np * sizeof(uint32_t);
    GOF_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&k->state, k->state_bytes);
    if (e != hipSuccess) {
        const size_t b = k->state_bytes;
        delete k;
        return fail(GOF_ENOMEM, "hipMalloc(%zu B of KNN samples): %s", b, hipGetErrorString(e));
    }
    // cv2 starts from an all-zero model with every flag clear; complete before create returns
    e = hipMemsetAsync(k->state, 0, k->state_bytes, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(k->state);
        delete k;
        return fail(GOF_EHIP, "zero-filling the KNN model: %s", hipGetErrorString(e));
    }
    *out = k;
    return GOF_OK;
}

void gof_knn_destroy(gof_knn *k)
{
    if (!k)
        return;
    (void)hipSetDevice(k->ctx->device);
    if (k->last && k->last != k->ctx->stream)
        (void)hipStreamSynchronize(k->last);
    (void)hipStreamSynchronize(k->ctx->stream);
    if (k->state)
        (void)hipFree(k->state);
    k->in.release();
    k->out.release();
    k->exp.release();
    delete k;
}

int gof_knn_nframes(const gof_knn *k) { return k ? k->nframes : -1; }

// cv2's BackgroundSubtractorKNNImpl::apply, the scalar part: frame count, learning rate
// 1 / min(2 n, history) in double, the three cycle lengths, the counters (this frame's values go to
// the kernel, the members advance) [OPENCV-RECALL]; restated identically in oracle/gof_oracle.c
static gofk::KnnFrame knn_next_frame(gof_knn *k)
{
    k->nframes++;
    const double lr = 1.0 / (double)std::min(2 * k->nframes, k->p.history);
    const int nN = k->p.nsamples;
    const int Kshort = (int)(log(0.7) / log(1 - lr)) + 1;
    const int Kmid = (int)(log(0.4) / log(1 - lr)) - Kshort + 1;
    const int Klong = (int)(log(0.1) / log(1 - lr)) - Kshort - Kmid + 1;
    gofk::KnnFrame f;
    f.n_short = Kshort / nN + 1;
    f.n_mid = Kmid / nN + 1;
    f.n_long = Klong / nN + 1;
    f.short_counter = k->short_counter;
    f.mid_counter = k->mid_counter;
    f.long_counter = k->long_counter;
    f.frame_no = (uint32_t)k->nframes;
    if (++k->short_counter >= f.n_short)
        k->short_counter = 0;
    if (++k->mid_counter >= f.n_mid)
        k->mid_counter = 0;
    if (++k->long_counter >= f.n_long)
        k->long_counter = 0;
    return f;
}

// (blur3: the pipeline's 3x3 blur runs inside the KNN kernel)
static int knn_apply_frame(gof_knn *k, const void *d_frames, void *d_masks, void *stream, const gofk::Taps *blur3)
{
    if (!k || !d_frames || !d_masks)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(k->ctx->device));
    hipStream_t s = pick(k->ctx, stream);
    if (k->last && k->last != s)
        GOF_HIP(hipStreamSynchronize(k->last));   // the model is a recurrence: one stream at a time
    k->last = s;
    const gofk::KnnFrame f = knn_next_frame(k);
    GOF_HIP(gofk::knn_apply((const uint8_t *)d_frames, (uint8_t *)d_masks, k->view(), k->streams, k->h, k->w,
                            k->c, f, s, blur3));
    return GOF_OK;
}

int gof_knn_apply_dev(gof_knn *k, const void *d_frames, void *d_masks, void *stream)
{
    return knn_apply_frame(k, d_frames, d_masks, stream, nullptr);
}

int gof_knn_apply_host(gof_knn *k, const uint8_t *frames, uint8_t *masks)
{
    if (!k || !frames || !masks)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(k->ctx->device));
    const size_t npx = (size_t)k->streams * k->h * k->w;
    int rc;
    if ((rc = k->in.reserve(npx * k->c)) || (rc = k->out.reserve(npx)))
        return rc;
    hipStream_t s = k->ctx->stream;
    GOF_HIP(hipMemcpyAsync(k->in.p, frames, npx * k->c, hipMemcpyHostToDevice, s));
    if ((rc = gof_knn_apply_dev(k, k->in.p, k->out.p, nullptr)))
        return rc;
    GOF_HIP(hipMemcpyAsync(masks, k->out.p, npx, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    return GOF_OK;
}

int gof_knn_export(gof_knn *k, uint8_t *samples, uint8_t *state)
{
    if (!k || !samples || !state)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(k->ctx->device));
    hipStream_t s = k->ctx->stream;
    if (k->last && k->last != s)
        GOF_HIP(hipStreamSynchronize(k->last));
    const size_t n = (size_t)k->streams * k->h * k->w;
    const size_t nb_s = (size_t)3 * k->p.nsamples * (k->c + 1) * n, nb_t = 6 * n;
    int rc;
    if ((rc = k->exp.reserve(nb_s + nb_t)))
        return rc;
    uint8_t *d = (uint8_t *)k->exp.p;
    GOF_HIP(gofk::knn_export(k->view(), k->c, n, d, d + nb_s, s));
    GOF_HIP(hipMemcpyAsync(samples, d, nb_s, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipMemcpyAsync(state, d + nb_s, nb_t, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    return GOF_OK;
}

int gof_fd_create(gof_ctx *ctx, int streams, int h, int w, int c, double threshold, gof_fd **out)
{
    if (!ctx || !out)
        return fail(GOF_EINVAL, "NULL argument");
    *out = nullptr;
    int rc = check_geom(streams, h, w, c);
    if (rc)
        return rc;
    if (!(threshold == threshold))
        return fail(GOF_EINVAL, "threshold is NaN");
    gof_fd *f = new (std::nothrow) gof_fd;
    if (!f)
        return fail(GOF_ENOMEM, "host allocation failed");
    f->ctx = ctx;
    f->streams = streams;
    f->h = h;
    f->w = w;
    f->c = c;
    f->threshold = threshold;
    *out = f;
    return GOF_OK;
}

void gof_fd_destroy(gof_fd *f)
{
    if (!f)
        return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    f->frame[0].release();
    f->frame[1].release();
    f->out.release();
    delete f;
}

namespace {
// cv.threshold on uint8 floors a float threshold [OPENCV-RECALL]; clamp so the
// int compare stays meaningful for any double.
int fd_ithr(double thr)
{
    double f = floor(thr);
    return f < -1 ? -1 : (f > 255 ? 255 : (int)f);
}
}  // namespace

/* The device entry point keeps its own copy of X_t as next call's X_{t-1}: the
 * reference keeps a *reference* to X (segmentation.py:66), which is the same
 * thing as long as the caller does not overwrite X, and safer when it does. */
int gof_fd_apply_dev(gof_fd *f, const void *d_frames, void *d_masks, void *stream)
{
    if (!f || !d_frames || !d_masks)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(f->ctx->device));
    hipStream_t s = pick(f->ctx, stream);
    size_t bytes = (size_t)f->streams * f->h * f->w * f->c;
    int rc;
    if ((rc = f->frame[0].reserve(bytes)) || (rc = f->frame[1].reserve(bytes)))
        return rc;
    const void *prev = f->frame[f->cur ^ 1].p;
    GOF_HIP(gofk::fd_apply((const uint8_t *)d_frames, (const uint8_t *)prev, f->has_prev,
                           fd_ithr(f->threshold), (uint8_t *)d_masks, f->streams, f->h, f->w, f->c,
                           s));
    GOF_HIP(hipMemcpyAsync(f->frame[f->cur].p, d_frames, bytes, hipMemcpyDeviceToDevice, s));
    f->cur ^= 1;
    f->has_prev = true;
    return GOF_OK;
}

int gof_fd_apply_host(gof_fd *f, const uint8_t *frames, uint8_t *masks)
{
    if (!f || !frames || !masks)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(f->ctx->device));
    hipStream_t s = f->ctx->stream;
    size_t npx = (size_t)f->streams * f->h * f->w, bytes = npx * f->c;
    int rc;
    if ((rc = f->frame[0].reserve(bytes)) || (rc = f->frame[1].reserve(bytes)) ||
        (rc = f->out.reserve(npx)))
        return rc;
    // upload straight into the ping-pong slot: no device-to-device copy
    void *cur = f->frame[f->cur].p;
    const void *prev = f->frame[f->cur ^ 1].p;
    GOF_HIP(hipMemcpyAsync(cur, frames, bytes, hipMemcpyHostToDevice, s));
    GOF_HIP(gofk::fd_apply((const uint8_t *)cur, (const uint8_t *)prev, f->has_prev,
                           fd_ithr(f->threshold), (uint8_t *)f->out.p, f->streams, f->h, f->w,
                           f->c, s));
    GOF_HIP(hipMemcpyAsync(masks, f->out.p, npx, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    f->cur ^= 1;
    f->has_prev = true;
    return GOF_OK;
}

/* ---- morphology ---------------------------------------------------------- */

int gof_morph_dev(gof_ctx *ctx, int op, const void *d_src, void *d_dst, int n_images, int h, int w,
                  const uint8_t *kernel, int kh, int kw, void *stream)
{
    if (!ctx || !d_src || !d_dst)
        return fail(GOF_EINVAL, "NULL argument");
    if (d_src == d_dst)
        return fail(GOF_EINVAL, "morphology cannot run in place");
    if (op != GOF_ERODE && op != GOF_DILATE)
        return fail(GOF_EINVAL, "op must be GOF_ERODE or GOF_DILATE");
    int rc = check_geom(n_images, h, w, 1);
    if (rc)
        return rc;
    gofk::SeBits se;
    if ((rc = make_sebits(kernel, kh, kw, &se)))
        return rc;
    GOF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (se_is_full_odd_rect(se)) {
        gofk::RectChain ch{};
        ch.n = 1;
        ch.op[0] = op;
        ch.rx[0] = kw / 2;
        ch.ry[0] = kh / 2;
        if (gofk::morph_rect_chain_supported(ch)) {
            GOF_HIP(gofk::morph_rect_chain((const uint8_t *)d_src, (uint8_t *)d_dst, n_images, h, w,
                                           ch, false, s));   // arbitrary grey levels
            return GOF_OK;
        }
    }
    GOF_HIP(gofk::morph_generic(op, (const uint8_t *)d_src, (uint8_t *)d_dst, n_images, h, w, se, s));
    return GOF_OK;
}

int gof_morph_host(gof_ctx *ctx, int op, const uint8_t *src, uint8_t *dst, int h, int w,
                   const uint8_t *kernel, int kh, int kw)
{
    if (!ctx || !src || !dst)
        return fail(GOF_EINVAL, "NULL argument");
    int rc = check_geom(1, h, w, 1);
    if (rc)
        return rc;
    size_t bytes = (size_t)h * w;
    GOF_HIP(hipSetDevice(ctx->device));
    if ((rc = ctx->in.reserve(bytes)) || (rc = ctx->out.reserve(bytes)))
        return rc;
    GOF_HIP(hipMemcpyAsync(ctx->in.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gof_morph_dev(ctx, op, ctx->in.p, ctx->out.p, 1, h, w, kernel, kh, kw, nullptr)))
        return rc;
    GOF_HIP(hipMemcpyAsync(dst, ctx->out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GOF_HIP(hipStreamSynchronize(ctx->stream));
    return GOF_OK;
}

/* ---- blob extraction ------------------------------------------------------- */

int gof_blobs_create(gof_ctx *ctx, int n_images, int h, int w, int crack_capacity, gof_blobs **out)
{
    if (!ctx || !out)
        return fail(GOF_EINVAL, "NULL argument");
    *out = nullptr;
    int rc = check_geom(n_images, h, w, 1);
    if (rc)
        return rc;
    if (h > 65535 || w > 65535 || n_images > 65535 || (size_t)h * w > ((size_t)1 << 30) || crack_capacity < 0)
        return fail(GOF_EINVAL, "blob extraction takes masks up to 65535 x 65535 (2^30 pixels)");
    const size_t N = (size_t)h * w;
    size_t cap = crack_capacity ? (size_t)crack_capacity : N / 2;
    if (cap < 4096)
        cap = 4096;
    if (cap > 2 * N + 16)
        cap = 2 * N + 16;           // no mask has more border edges than that
    cap = (cap + 3) & ~(size_t)3;
    if (cap > ((size_t)1 << 28))
        return fail(GOF_EINVAL, "crack capacity too large");
    gof_blobs *b = new (std::nothrow) gof_blobs;
    if (!b)
        return fail(GOF_ENOMEM, "host allocation failed");
    b->ctx = ctx;
    b->n_img = n_images;
    b->h = h;
    b->w = w;
    gofk::BlobBuffers &B = b->B;
    B.n_img = n_images;
    B.h = h;
    B.w = w;
    B.cap = (uint32_t)cap;
    // header + up to 4096 records + one point per four crack slots
    B.out_words = gofk::kBlobHeaderWords + 4096 * gofk::kBlobRecordWords + (uint32_t)(cap / 4);
    const size_t n = (size_t)n_images;
    const size_t acc_words = n * (cap / 4) * sizeof(gofk::BlobAcc) / 4;
    B.seg_cap = (uint32_t)h * (uint32_t)(4 * ((w + 255) / 256));
    const size_t words = acc_words + n * N + 12 * n * cap + n * B.seg_cap + n * gofk::kBlobCounters +
                         n * B.out_words;
    const size_t bytes = words * 4 + 256;
    GOF_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&b->dev, bytes);
    if (e != hipSuccess) {
        delete b;
        return fail(GOF_ENOMEM, "hipMalloc(%zu B of blob scratch): %s", bytes, hipGetErrorString(e));
    }
    uint32_t *q = (uint32_t *)b->dev;
    auto take = [&](size_t cnt) {
        uint32_t *r = q;
        q += cnt;
        return r;
    };
    B.acc = (gofk::BlobAcc *)take(acc_words);   // 8-byte aligned: first
    B.cbase = take(n * N);
    B.ckey = take(n * cap);
    B.csucc = take(n * cap);
    B.cparent = take(n * cap);
    B.caux = take(n * cap);
    B.cpos = take(n * cap);
    B.cslot = take(n * cap);
    B.rsel = take(n * cap);
    B.rnxt = take(n * cap);
    B.rsum = take(n * cap);
    B.rnxt2 = take(n * cap);
    B.rsum2 = take(n * cap);
    B.segs = take(n * B.seg_cap);
    B.counters = take(n * gofk::kBlobCounters);
    B.out = take(n * B.out_words);
    b->head_words = B.out_words < 8192u ? B.out_words : 8192u;   // 32 KB per image travel unasked
    e = hipHostMalloc((void **)&b->host, n * b->head_words * 4, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipFree(b->dev);
        delete b;
        return fail(GOF_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e));
    }
    // the finishing kernel writes the head of every image's block straight into that pinned buffer
    // (no copy operation behind an extraction) and clears the counters behind itself (no memset in
    // front of one): they start out zero here
    void *mirror = nullptr;
    e = hipHostGetDevicePointer(&mirror, b->host, 0);
    if (e == hipSuccess)
        e = hipMemset(B.counters, 0, n * gofk::kBlobCounters * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(b->dev);
        (void)hipHostFree(b->host);
        delete b;
        return fail(GOF_EHIP, "blob scratch set-up: %s", hipGetErrorString(e));
    }
    B.host_out = (uint32_t *)mirror;
    B.host_words = b->head_words;
    *out = b;
    return GOF_OK;
}

void gof_blobs_destroy(gof_blobs *b)
{
    if (!b)
        return;
    (void)hipSetDevice(b->ctx->device);
    if (b->last)
        (void)hipStreamSynchronize(b->last);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->dev)
        (void)hipFree(b->dev);
    if (b->host)
        (void)hipHostFree(b->host);
    b->in.release();
    delete b;
}

int gof_blobs_extract_dev(gof_blobs *b, const void *d_masks, int n_images, double min_area, void *stream)
{
    if (!b || !d_masks)
        return fail(GOF_EINVAL, "NULL argument");
    if (n_images < 1 || n_images > b->n_img)
        return fail(GOF_EINVAL, "n_images must be in [1, %d], got %d", b->n_img, n_images);
    if (!(min_area == min_area))
        return fail(GOF_EINVAL, "min_area is NaN");
    GOF_HIP(hipSetDevice(b->ctx->device));
    hipStream_t s = pick(b->ctx, stream);
    gofk::BlobBuffers B = b->B;
    B.n_img = n_images;
    if (b->dirty) {   // (an earlier launch failure left the counters half-used)
        GOF_HIP(hipMemsetAsync(B.counters, 0, (size_t)b->n_img * gofk::kBlobCounters * sizeof(uint32_t), s));
        b->dirty = false;
    }
    const hipError_t ee = gofk::blob_extract((const uint8_t *)d_masks, B, min_area, s);
    if (ee != hipSuccess) {
        b->dirty = true;
        return fail(GOF_EHIP, "blob extraction launch: %s", hipGetErrorString(ee));
    }
    // (the records arrive in b->host with the finishing kernel's own stores: nothing to copy)
    b->last = s;
    b->pending = true;
    b->n_last = n_images;
    b->spill_image = -1;
    return GOF_OK;
}

int gof_blobs_extract_host(gof_blobs *b, const uint8_t *masks, int n_images, double min_area)
{
    if (!b || !masks)
        return fail(GOF_EINVAL, "NULL argument");
    if (n_images < 1 || n_images > b->n_img)
        return fail(GOF_EINVAL, "n_images must be in [1, %d], got %d", b->n_img, n_images);
    GOF_HIP(hipSetDevice(b->ctx->device));
    const size_t bytes = (size_t)n_images * b->h * b->w;
    int rc;
    if ((rc = b->in.reserve(bytes)))
        return rc;
    hipStream_t s = b->ctx->stream;
    GOF_HIP(hipMemcpyAsync(b->in.p, masks, bytes, hipMemcpyHostToDevice, s));
    if ((rc = gof_blobs_extract_dev(b, b->in.p, n_images, min_area, nullptr)))
        return rc;
    GOF_HIP(hipStreamSynchronize(s));
    b->pending = false;
    return GOF_OK;
}

namespace {
int blobs_wait(gof_blobs *b, int image)
{
    if (!b)
        return fail(GOF_EINVAL, "NULL argument");
    if (image < 0 || image >= b->n_last)
        return fail(GOF_EINVAL, "image %d out of range [0, %d)", image, b->n_last);
    if (b->pending) {
        GOF_HIP(hipSetDevice(b->ctx->device));
        GOF_HIP(hipStreamSynchronize(b->last));
        b->pending = false;
    }
    return GOF_OK;
}
}  // namespace

/* diagnostics (profiles/micro): the device counters of one image of the last extraction --
 * tiles, cracks, roots, flags, then phase stamps of the finishing kernel; not in the header */
extern "C" int gof_blobs_debug_counters(gof_blobs *b, int image, uint32_t *out16)
{
    int rc = blobs_wait(b, image);
    if (rc || !out16)
        return rc ? rc : GOF_EINVAL;
    GOF_HIP(hipMemcpy(out16, b->B.counters + (size_t)image * gofk::kBlobCounters,
                      gofk::kBlobCounters * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++)   // (the live counters are cleared by the finishing kernel: its copies)
        out16[i] = out16[gofk::kCntLast + i];
    return GOF_OK;
}

int gof_blobs_counts(gof_blobs *b, int image, int *n_blobs, int *n_points, int *flags)
{
    if (!n_blobs || !n_points || !flags)
        return fail(GOF_EINVAL, "NULL argument");
    int rc = blobs_wait(b, image);
    if (rc)
        return rc;
    const gofk::BlobHeader *h = (const gofk::BlobHeader *)(b->host + (size_t)image * b->head_words);
    *n_blobs = (int)h->n_blobs;
    *n_points = (int)h->n_points;
    *flags = (int)h->flags;
    return GOF_OK;
}

int gof_blobs_fetch(gof_blobs *b, int image, gof_blob_record *records, int max_records,
                    int32_t *points_xy, int max_points)
{
    int rc = blobs_wait(b, image);
    if (rc)
        return rc;
    const uint32_t *blk = b->host + (size_t)image * b->head_words;
    const gofk::BlobHeader h = *(const gofk::BlobHeader *)blk;
    if (h.flags != 0)
        return fail(GOF_EINVAL, "image %d has no result (flags %u): fall back to the mask", image, h.flags);
    if ((int)h.n_blobs > max_records || (int)h.n_points > max_points || (h.n_blobs && !records) ||
        (h.n_points && !points_xy))
        return fail(GOF_EINVAL, "image %d holds %u blobs / %u points, buffers take %d / %d", image,
                    h.n_blobs, h.n_points, max_records, max_points);
    const size_t words = gofk::kBlobHeaderWords + (size_t)h.n_blobs * gofk::kBlobRecordWords + h.n_points;
    if (words > b->head_words) {   // the rest of a large result, on demand
        if (b->spill_image != image) {
            try {
                b->spill.resize(words);
            } catch (const std::bad_alloc &) {
                return fail(GOF_ENOMEM, "host allocation failed");
            }
            GOF_HIP(hipSetDevice(b->ctx->device));
            GOF_HIP(hipMemcpyAsync(b->spill.data(), b->B.out + (size_t)image * b->B.out_words, words * 4,
                                   hipMemcpyDeviceToHost, b->last));
            GOF_HIP(hipStreamSynchronize(b->last));
            b->spill_image = image;
        }
        blk = b->spill.data();
    }
    static_assert(sizeof(gof_blob_record) == sizeof(gofk::BlobRecord), "record layouts differ");
    if (h.n_blobs)
        memcpy(records, blk + gofk::kBlobHeaderWords, (size_t)h.n_blobs * sizeof(gof_blob_record));
    const uint32_t *pts = blk + gofk::kBlobHeaderWords + (size_t)h.n_blobs * gofk::kBlobRecordWords;
    for (uint32_t i = 0; i < h.n_points; i++) {
        points_xy[2 * i] = (int32_t)(pts[i] & 0xffffu);
        points_xy[2 * i + 1] = (int32_t)(pts[i] >> 16);
    }
    return GOF_OK;
}

/* ---- fused pipeline ------------------------------------------------------ */

int gof_pipeline_create(gof_ctx *ctx, const gof_pipeline_desc *desc, gof_pipeline **out)
{
    if (!ctx || !desc || !out)
        return fail(GOF_EINVAL, "NULL argument");
    *out = nullptr;
    int rc = check_geom(desc->streams, desc->h, desc->w, desc->c);
    if (rc)
        return rc;
    if (desc->n_morph < 0 || desc->n_morph > GOF_MAX_MORPH)
        return fail(GOF_EINVAL, "n_morph must be in [0, %d]", GOF_MAX_MORPH);
    if (desc->detector != GOF_DET_MOG2 && desc->detector != GOF_DET_FD && desc->detector != GOF_DET_KNN)
        return fail(GOF_EINVAL, "Unknown method for foreground detection");
    if (desc->blur != GOF_BLUR_NONE && desc->blur != GOF_BLUR_GAUSS && desc->blur != GOF_BLUR_BOX)
        return fail(GOF_EINVAL, "unknown blur kind %d", desc->blur);
    gof_pipeline *p = new (std::nothrow) gof_pipeline;
    if (!p)
        return fail(GOF_ENOMEM, "host allocation failed");
    p->ctx = ctx;
    p->d = *desc;
    auto bail = [&](int code) {
        gof_pipeline_destroy(p);
        return code;
    };
    if (desc->blur == GOF_BLUR_GAUSS) {
        if ((rc = make_taps(desc->blur_kw, desc->blur_kh, desc->blur_sigma_x, desc->blur_sigma_y,
                            &p->taps)))
            return bail(rc);
        // fused into the detector kernel when the taps allow (Taps::fused3_ok)
        p->fuse_blur3 = desc->detector == GOF_DET_MOG2 && p->taps.fused3_ok;
        p->fuse_fd3 = desc->detector == GOF_DET_FD && p->taps.fused3_ok;   // + geometry: fd_frames_supported()
    } else if (desc->blur == GOF_BLUR_BOX) {
        if ((rc = make_box(desc->blur_kw, desc->blur_kh, &p->taps)))
            return bail(rc);
        // cv.blur 3x3 (what the reference's tests configure, tests/test_pipeline.py:68-75) likewise
        const bool no_box_fusion = getenv("GOF_NO_FUSE_BOX") != nullptr;   // A/B knob (profiles/)
        p->fuse_blur3 = desc->detector == GOF_DET_MOG2 && p->taps.fused3_ok && !no_box_fusion;
        p->fuse_fd3 = desc->detector == GOF_DET_FD && p->taps.fused3_ok && !no_box_fusion;
    }
    bool all_rect = desc->n_morph > 0;
    p->chain.n = desc->n_morph;
    for (int i = 0; i < desc->n_morph; i++) {
        const gof_morph_op &o = desc->morph[i];
        if (o.op != GOF_ERODE && o.op != GOF_DILATE)
            return bail(fail(GOF_EINVAL, "morph[%d].op must be GOF_ERODE or GOF_DILATE", i));
        gofk::SeBits se;
        if ((rc = make_sebits(o.kernel, o.kh, o.kw, &se)))
            return bail(rc);
        p->sebits.push_back(se);
        all_rect = all_rect && se_is_full_odd_rect(se);
        p->chain.op[i] = o.op;
        p->chain.rx[i] = o.kw / 2;
        p->chain.ry[i] = o.kh / 2;
        p->d.morph[i].kernel = nullptr;   // the bits are owned now; drop the borrowed pointer
    }
    p->chain_ok = all_rect && gofk::morph_rect_chain_supported(p->chain);
    if (desc->detector == GOF_DET_MOG2) {
        if ((rc = gof_mog2_create(ctx, desc->streams, desc->h, desc->w, desc->c, &desc->mog2,
                                  &p->mog2)))
            return bail(rc);
    } else if (desc->detector == GOF_DET_KNN) {
        // cv.createBackgroundSubtractorKNN() as the reference calls it: all defaults (pipeline.py:233-234)
        if ((rc = gof_knn_create(ctx, desc->streams, desc->h, desc->w, desc->c, nullptr, &p->knn)))
            return bail(rc);
    } else {
        if ((rc = gof_fd_create(ctx, desc->streams, desc->h, desc->w, desc->c, desc->fd_threshold,
                                &p->fd)))
            return bail(rc);
    }
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess)
        e = hipMalloc((void **)&p->d_sum, sizeof(unsigned long long));
    if (e != hipSuccess)
        return bail(fail(GOF_EHIP, "pipeline scratch: %s", hipGetErrorString(e)));
    *out = p;
    return GOF_OK;
}

void gof_pipeline_destroy(gof_pipeline *p)
{
    if (!p)
        return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    gof_mog2_destroy(p->mog2);
    gof_fd_destroy(p->fd);
    gof_knn_destroy(p->knn);
    if (p->copy_stream) {
        (void)hipStreamSynchronize(p->copy_stream);
        (void)hipStreamDestroy(p->copy_stream);
    }
    for (int i = 0; i < 2; i++)
        for (hipEvent_t e : {p->ev_up[i], p->ev_free[i], p->ev_done[i]})
            if (e)
                (void)hipEventDestroy(e);
    p->in.release();
    p->out.release();
    p->in1.release();
    p->out1.release();
    p->blurred.release();
    p->m0.release();
    p->m1.release();
    p->gray.release();
    p->rowflags.release();
    p->fd_prev[0].release();
    p->fd_prev[1].release();
    if (p->d_sum)
        (void)hipFree(p->d_sum);
    for (hipEvent_t e : p->events)
        (void)hipEventDestroy(e);
    delete p;
}

gof_mog2 *gof_pipeline_mog2(gof_pipeline *p) { return p ? p->mog2 : nullptr; }
gof_knn *gof_pipeline_knn(gof_pipeline *p) { return p ? p->knn : nullptr; }

int gof_pipeline_set_convert(gof_pipeline *p, int c_in, int rgb_order)
{
    if (!p)
        return fail(GOF_EINVAL, "NULL argument");
    if (c_in != 0 && (c_in != 3 && c_in != 4))
        return fail(GOF_EINVAL, "conversion to grey takes 3- or 4-channel frames, got %d", c_in);
    if (c_in != 0 && p->d.c != 1)
        return fail(GOF_EINVAL, "a converting pipeline works on one channel; it was created with c = %d", p->d.c);
    p->cvt_c = c_in;
    p->cvt_rgb = rgb_order != 0;
    return GOF_OK;
}

// channels of the frames an apply call is handed
static int in_channels(const gof_pipeline *p) { return p->cvt_c ? p->cvt_c : p->d.c; }

int gof_pipeline_apply_dev_many(gof_pipeline *p, const void *d_frames, void *d_masks, int n_frames,
                                void *stream)
{
    if (!p || !d_frames || !d_masks)
        return fail(GOF_EINVAL, "NULL argument");
    if (n_frames < 1 || n_frames > 4096)
        return fail(GOF_EINVAL, "n_frames must be in [1, 4096], got %d", n_frames);
    GOF_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = pick(p->ctx, stream);
    const gof_pipeline_desc &d = p->d;
    const int T = n_frames;
    const size_t npx = (size_t)d.streams * d.h * d.w;   // pixels of ONE frame of every stream
    if ((size_t)T * d.streams > 65535)
        return fail(GOF_EINVAL, "n_frames * streams must not exceed 65535");
    int rc;

    // step 0 (pipeline.py:458-459), when switched on: BGR / RGB -> grey, on the device
    const uint8_t *x = (const uint8_t *)d_frames;
    // ... fused into the MOG2 kernel's staging when the 3x3 blur is (no grey image in HBM, one
    // detector launch); otherwise its own kernel
    bool cvt_fused = false;
    static const bool no_step0_fusion = getenv("GOF_NO_FUSE_STEP0") != nullptr;   // A/B knob (profiles/)
    static const bool no_knn_fusion = getenv("GOF_NO_FUSE_KNN_BLUR") != nullptr;  // A/B knob (profiles/)
    if (p->cvt_c && d.c == 1 && !no_step0_fusion) {
        if (p->mog2 && p->fuse_blur3) {
            gofk::Mog2Consts kc{};
            kc.var_min = p->mog2->p.var_min;
            kc.var_max = p->mog2->p.var_max;
            cvt_fused = gofk::mog2_convert_supported(x, d.streams, d.h, d.w, p->cvt_c, p->taps, kc);
        } else if (p->fd && p->fuse_fd3 && p->fd_mode != 2 && T > 1) {
            // multi-frame calls only: one-frame calls are faster through gray_quad_kernel (the staged
            // conversion works on 1.5 x the pixels, with a second barrier; profiles/r03_step0_box_fusion.txt),
            // and so is KNN, one launch per frame.  (The grey frames must suit the fused kernel as well:
            // a call whose colour frames do not -- a misaligned pointer -- goes through the grey buffer
            // and then the same kernel; the hand-over format between launches is the same.)
            cvt_fused = gofk::fd_frames_supported(x, d.streams, d.h, d.w, 1, p->taps, p->cvt_c) &&
                        gofk::fd_frames_supported(nullptr, d.streams, d.h, d.w, 1, p->taps);
            if (cvt_fused)
                p->fd_mode = 1;
        }
    }
    if (p->cvt_c && !cvt_fused) {
        if ((rc = p->gray.reserve((size_t)T * npx)))
            return rc;
        GOF_HIP(gofk::cvt_gray(x, (uint8_t *)p->gray.p, T * d.streams, d.h, d.w, p->cvt_c, p->cvt_rgb, s));
        x = (const uint8_t *)p->gray.p;
    }
    const int xc = cvt_fused ? p->cvt_c : d.c;   // channels of the frames at x
    // step 1 (pipeline.py:462-467) -- unless it is fused into step 2
    // FD: the fused kernel needs aligned frames.  The choice is made on the first
    // call and kept (the two paths keep the previous frame in different formats); a
    // later misaligned pointer is copied to an aligned buffer first.
    if (p->fuse_fd3 && p->fd && p->fd_mode == 0)
        p->fd_mode = gofk::fd_frames_supported(x, d.streams, d.h, d.w, d.c, p->taps) ? 1 : 2;
    const bool fd_fused = p->fd_mode == 1;
    if (fd_fused && !cvt_fused && !gofk::fd_frames_supported(x, d.streams, d.h, d.w, d.c, p->taps)) {
        if ((rc = p->in.reserve((size_t)T * npx * d.c)))
            return rc;
        GOF_HIP(hipMemcpyAsync(p->in.p, x, (size_t)T * npx * d.c, hipMemcpyDeviceToDevice, s));
        x = (const uint8_t *)p->in.p;
    }
    // KNN: the 3x3 blur runs inside the KNN kernel when the frames are aligned (as the fused FD kernel)
    const bool knn_fused = p->knn && d.blur != GOF_BLUR_NONE && p->taps.fused3_ok && !no_knn_fusion &&
                           gofk::knn_blur3_supported(x, d.streams, d.h, d.w, d.c, p->taps);
    if (d.blur != GOF_BLUR_NONE && !p->fuse_blur3 && !fd_fused && !knn_fused) {
        if ((rc = p->blurred.reserve((size_t)T * npx * d.c)))
            return rc;
        GOF_HIP(gofk::blur_sep_q8(x, (uint8_t *)p->blurred.p, T * d.streams, d.h, d.w, d.c, p->taps, s));
        x = (const uint8_t *)p->blurred.p;
    }

    const bool prof = p->prof && (p->prof_seen++ % (size_t)p->prof_stride) == 0 &&
                      3 * (p->prof_used + 1) <= p->events.size();
    hipEvent_t *ev = prof ? &p->events[3 * p->prof_used] : nullptr;
    if (prof)
        GOF_HIP(hipEventRecord(ev[0], s));

    // step 2 (pipeline.py:470)
    uint8_t *det_out = (uint8_t *)d_masks;
    uint8_t *rf = nullptr;   // row flags of the detector masks (MOG2 multi-frame launches only)
    if (d.n_morph > 0) {
        if ((rc = p->m0.reserve((size_t)T * npx)))
            return rc;
        det_out = (uint8_t *)p->m0.p;
    }
    if (p->mog2) {
        // temporal blocking: up to kMaxT frames per launch, the model moves once
        // row flags for the morphology (which 64-px mask segments are all zero), one
        // plane per launch chunk; only worth having when a fused chain follows
        const size_t rf_chunk = (size_t)d.streams * d.h * gofk::rowflag_pitch(d.w);
        const int chunks = (T + gofk::kMaxT - 1) / gofk::kMaxT;
        // (one-frame calls go without: there the flag store costs the detector what the
        // skipping saves the morphology)
        if (d.n_morph > 0 && p->chain_ok && T > 1) {
            if ((rc = p->rowflags.reserve(rf_chunk * chunks)))
                return rc;
            rf = (uint8_t *)p->rowflags.p;
        }
        for (int t0 = 0; t0 < T; t0 += gofk::kMaxT) {
            const int tn = T - t0 < gofk::kMaxT ? T - t0 : gofk::kMaxT;
            gofk::Mog2Consts k = next_consts(p->mog2, tn);
            uint8_t *rfc = rf ? rf + (size_t)(t0 / gofk::kMaxT) * rf_chunk : nullptr;
            bool written = false;
            p->mog2->last = s;
            hipError_t e;
            if (cvt_fused) {
                e = gofk::mog2_convert_apply(x + (size_t)t0 * npx * p->cvt_c, det_out + (size_t)t0 * npx,
                                             p->mog2->view(), d.streams, d.h, d.w, p->cvt_c, p->cvt_rgb, tn, k,
                                             p->taps, s, rfc);
                written = rfc != nullptr;
            } else {
                e = gofk::mog2_apply(x + (size_t)t0 * npx * d.c, det_out + (size_t)t0 * npx,
                                     p->mog2->view(), d.streams, d.h, d.w, d.c, tn, k,
                                     p->fuse_blur3, p->taps, s, rfc, &written,
                                     (size_t)(T - t0) * npx * d.c);
            }
            if (e != hipSuccess) {
                p->mog2->nframes -= tn;
                return fail(GOF_EHIP, "mog2 launch: %s", hipGetErrorString(e));
            }
            if (rfc && !written)   // (every MOG2 kernel writes them today; keep the contract explicit)
                GOF_HIP(hipMemsetAsync(rfc, 0xFF, rf_chunk, s));   // "nothing is known to be zero"
        }
    } else if (p->knn) {
        // one launch per frame (the refresh counters and the draws' key advance per frame)
        for (int t = 0; t < T; t++)
            if ((rc = knn_apply_frame(p->knn, x + (size_t)t * npx * d.c, det_out + (size_t)t * npx, s,
                                      knn_fused ? &p->taps : nullptr)))
                return rc;
    } else if (fd_fused) {
        // blur + difference for up to kMaxT frames per launch; the blurred last frame is
        // handed from launch to launch in fd_prev
        const size_t bytes = npx * xc;
        if ((rc = p->fd_prev[0].reserve(npx * 4)) || (rc = p->fd_prev[1].reserve(npx * 4)))
            return rc;
        const size_t rf_chunk = (size_t)d.streams * d.h * gofk::rowflag_pitch(d.w);
        if (d.n_morph > 0 && p->chain_ok && T > 1) {
            if ((rc = p->rowflags.reserve(rf_chunk * ((T + gofk::kMaxT - 1) / gofk::kMaxT))))
                return rc;
            rf = (uint8_t *)p->rowflags.p;
        }
        for (int t0 = 0; t0 < T; t0 += gofk::kMaxT) {
            const int tn = T - t0 < gofk::kMaxT ? T - t0 : gofk::kMaxT;
            GOF_HIP(gofk::fd_frames_apply(x + (size_t)t0 * bytes, det_out + (size_t)t0 * npx,
                                          (const uint32_t *)p->fd_prev[p->fd_cur ^ 1].p,
                                          (uint32_t *)p->fd_prev[p->fd_cur].p, p->fd_has_prev,
                                          fd_ithr(p->fd->threshold), d.streams, d.h, d.w, d.c, tn,
                                          p->taps, s,
                                          rf ? rf + (size_t)(t0 / gofk::kMaxT) * rf_chunk : nullptr,
                                          cvt_fused ? p->cvt_c : 0, p->cvt_rgb));
            p->fd_cur ^= 1;
            p->fd_has_prev = true;
        }
    } else {
        for (int t = 0; t < T; t++)
            if ((rc = gof_fd_apply_dev(p->fd, x + (size_t)t * npx * d.c, det_out + (size_t)t * npx, s)))
                return rc;
    }
    if (prof)
        GOF_HIP(hipEventRecord(ev[1], s));

    // step 3 (pipeline.py:473-475): all T * streams masks in one launch per kernel
    if (d.n_morph > 0) {
        if (p->chain_ok) {
            // detector masks are chain-valued ({0, shadow, 255} / {0, 84, 85} / {0, 255})
            GOF_HIP(gofk::morph_rect_chain(det_out, (uint8_t *)d_masks, T * d.streams, d.h, d.w,
                                           p->chain, true, s, rf, d.streams));
        } else {
            if (d.n_morph > 1 && (rc = p->m1.reserve((size_t)T * npx)))
                return rc;
            const uint8_t *src = det_out;
            for (int i = 0; i < d.n_morph; i++) {
                uint8_t *dst = i == d.n_morph - 1 ? (uint8_t *)d_masks
                               : (src == (const uint8_t *)p->m0.p ? (uint8_t *)p->m1.p
                                                                   : (uint8_t *)p->m0.p);
                GOF_HIP(gofk::morph_generic(d.morph[i].op, src, dst, T * d.streams, d.h, d.w,
                                            p->sebits[(size_t)i], s));
                src = dst;
            }
        }
    }
    if (prof) {
        GOF_HIP(hipEventRecord(ev[2], s));
        p->prof_used++;
    }
    return GOF_OK;
}

int gof_pipeline_apply_dev(gof_pipeline *p, const void *d_frames, void *d_masks, void *stream)
{
    return gof_pipeline_apply_dev_many(p, d_frames, d_masks, 1, stream);
}

int gof_pipeline_apply_host_many(gof_pipeline *p, const uint8_t *frames, uint8_t *masks,
                                 int n_frames)
{
    if (!p || !frames || !masks)
        return fail(GOF_EINVAL, "NULL argument");
    if (n_frames < 1 || n_frames > 4096)
        return fail(GOF_EINVAL, "n_frames must be in [1, 4096], got %d", n_frames);
    GOF_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const size_t npx = (size_t)n_frames * p->d.streams * p->d.h * p->d.w;
    const size_t cin = (size_t)in_channels(p);
    int rc;
    if ((rc = p->in.reserve(npx * cin)) || (rc = p->out.reserve(npx)))
        return rc;
    GOF_HIP(hipMemcpyAsync(p->in.p, frames, npx * cin, hipMemcpyHostToDevice, s));
    if ((rc = gof_pipeline_apply_dev_many(p, p->in.p, p->out.p, n_frames, nullptr)))
        return rc;
    GOF_HIP(hipMemcpyAsync(masks, p->out.p, npx, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    p->last_slot = 0;
    return GOF_OK;
}

int gof_pipeline_apply_host(gof_pipeline *p, const uint8_t *frames, uint8_t *masks)
{
    return gof_pipeline_apply_host_many(p, frames, masks, 1);
}

// One host batch into a slot: upload on the copy stream (from pageable memory this call blocks
// until the frames have left the caller's buffers -- while the context's stream runs the kernels
// of the batch submitted before), then steps 1-3 and the blob extraction enqueued behind it.
// inline_copy: the synchronous entry points -- nothing to overlap with, so the upload goes on the context's own
// stream and the kernels follow it without the copy stream's event (no cross-queue dependency to resolve before
// the first kernel starts).
static int submit_host_blobs(gof_pipeline *p, const uint8_t *const *frames, const uint8_t *packed, int n_frames,
                             gof_blobs *b, double min_area, int slot, bool inline_copy = false)
{
    if (!p || !b || (!frames && !packed))
        return fail(GOF_EINVAL, "NULL argument");
    if (slot != 0 && slot != 1)
        return fail(GOF_EINVAL, "slot must be 0 or 1, got %d", slot);
    if (n_frames < 1 || n_frames > 4096)
        return fail(GOF_EINVAL, "n_frames must be in [1, 4096], got %d", n_frames);
    if (b->ctx != p->ctx || b->h != p->d.h || b->w != p->d.w || b->n_img < n_frames * p->d.streams)
        return fail(GOF_EINVAL, "blob scratch (%d images of %dx%d) does not fit %d frames of %d streams of %dx%d",
                    b->n_img, b->h, b->w, n_frames, p->d.streams, p->d.h, p->d.w);
    if (frames)
        for (int i = 0; i < n_frames; i++)
            if (!frames[i])
                return fail(GOF_EINVAL, "frames[%d] is NULL", i);
    GOF_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    if (!p->copy_stream) {
        GOF_HIP(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            GOF_HIP(hipEventCreateWithFlags(&p->ev_up[i], hipEventDisableTiming));
            GOF_HIP(hipEventCreateWithFlags(&p->ev_free[i], hipEventDisableTiming));
            GOF_HIP(hipEventCreateWithFlags(&p->ev_done[i], hipEventDisableTiming));
        }
    }
    DevBuf &in = slot ? p->in1 : p->in, &out = slot ? p->out1 : p->out;
    const size_t per = (size_t)p->d.streams * p->d.h * p->d.w, cin = (size_t)in_channels(p);
    // the slot's buffers are free once the batch that used them last has been through its kernels
    // (and its caller has had its records: a slot is waited for before it is submitted again)
    if (p->slot_blobs[slot]) {
        GOF_HIP(hipEventSynchronize(p->ev_done[slot]));
        p->slot_blobs[slot]->pending = false;
        p->slot_blobs[slot] = nullptr;
    }
    int rc;
    if ((rc = in.reserve((size_t)n_frames * per * cin)) || (rc = out.reserve((size_t)n_frames * per)))
        return rc;
    hipStream_t up = inline_copy ? s : p->copy_stream;
    if (packed) {
        GOF_HIP(hipMemcpyAsync(in.p, packed, (size_t)n_frames * per * cin, hipMemcpyHostToDevice, up));
    } else {
        for (int i = 0; i < n_frames; i++)
            GOF_HIP(hipMemcpyAsync((uint8_t *)in.p + (size_t)i * per * cin, frames[i], per * cin,
                                   hipMemcpyHostToDevice, up));
    }
    if (!inline_copy) {
        GOF_HIP(hipEventRecord(p->ev_up[slot], p->copy_stream));
        GOF_HIP(hipStreamWaitEvent(s, p->ev_up[slot], 0));
    }
    if ((rc = gof_pipeline_apply_dev_many(p, in.p, out.p, n_frames, nullptr)))
        return rc;
    if ((rc = gof_blobs_extract_dev(b, out.p, n_frames * p->d.streams, min_area, nullptr)))
        return rc;
    GOF_HIP(hipEventRecord(p->ev_done[slot], s));
    p->slot_blobs[slot] = b;
    p->last_slot = slot;
    return GOF_OK;
}

int gof_pipeline_submit_host_blobs_ptrs(gof_pipeline *p, const uint8_t *const *frames, int n_frames,
                                        gof_blobs *b, double min_area, int slot)
{
    if (!frames)
        return fail(GOF_EINVAL, "NULL argument");
    return submit_host_blobs(p, frames, nullptr, n_frames, b, min_area, slot);
}

int gof_pipeline_wait_slot(gof_pipeline *p, int slot)
{
    if (!p || (slot != 0 && slot != 1))
        return fail(GOF_EINVAL, "bad argument");
    if (!p->slot_blobs[slot])
        return GOF_OK;   // nothing in flight there
    GOF_HIP(hipSetDevice(p->ctx->device));
    GOF_HIP(hipEventSynchronize(p->ev_done[slot]));
    p->slot_blobs[slot]->pending = false;
    p->slot_blobs[slot] = nullptr;
    return GOF_OK;
}

int gof_pipeline_apply_host_blobs(gof_pipeline *p, const uint8_t *frames, int n_frames, gof_blobs *b,
                                  double min_area)
{
    if (!frames)
        return fail(GOF_EINVAL, "NULL argument");
    int rc = submit_host_blobs(p, nullptr, frames, n_frames, b, min_area, 0, true);
    return rc ? rc : gof_pipeline_wait_slot(p, 0);
}

int gof_pipeline_apply_host_blobs_ptrs(gof_pipeline *p, const uint8_t *const *frames, int n_frames,
                                       gof_blobs *b, double min_area)
{
    if (!frames)
        return fail(GOF_EINVAL, "NULL argument");
    int rc = submit_host_blobs(p, frames, nullptr, n_frames, b, min_area, 0, true);
    return rc ? rc : gof_pipeline_wait_slot(p, 0);
}

int gof_pipeline_fetch_masks(gof_pipeline *p, uint8_t *masks, int n_frames)
{
    return gof_pipeline_fetch_masks_slot(p, masks, n_frames, p ? p->last_slot : 0);
}

int gof_pipeline_fetch_masks_slot(gof_pipeline *p, uint8_t *masks, int n_frames, int slot)
{
    if (!p || !masks || n_frames < 1 || (slot != 0 && slot != 1))
        return fail(GOF_EINVAL, "bad argument");
    const size_t npx = (size_t)n_frames * p->d.streams * p->d.h * p->d.w;
    const DevBuf &out = slot ? p->out1 : p->out;
    if (!out.p || out.cap < npx)
        return fail(GOF_EINVAL, "no masks of %d frame(s) on the device", n_frames);
    GOF_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    GOF_HIP(hipMemcpyAsync(masks, out.p, npx, hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    return GOF_OK;
}

int gof_pipeline_profile_enable(gof_pipeline *p, int enable, int capacity)
{
    if (!p || capacity < 0 || capacity > (1 << 20))
        return fail(GOF_EINVAL, "bad argument");
    GOF_HIP(hipSetDevice(p->ctx->device));
    p->prof = enable != 0;
    p->prof_used = 0;
    p->prof_seen = 0;
    while (p->prof && p->events.size() < 3 * (size_t)capacity) {
        hipEvent_t e;
        GOF_HIP(hipEventCreate(&e));
        p->events.push_back(e);
    }
    return GOF_OK;
}

int gof_pipeline_profile_stride(gof_pipeline *p, int stride)
{
    if (!p || stride < 1)
        return fail(GOF_EINVAL, "bad argument");
    p->prof_stride = stride;
    return GOF_OK;
}

int gof_pipeline_profile_query(gof_pipeline *p, double *detector_ms, double *morph_ms,
                               int *n_applies)
{
    if (!p || !detector_ms || !morph_ms || !n_applies)
        return fail(GOF_EINVAL, "NULL argument");
    GOF_HIP(hipSetDevice(p->ctx->device));
    double a = 0, b = 0;
    for (size_t i = 0; i < p->prof_used; i++) {
        float ms = 0;
        GOF_HIP(hipEventSynchronize(p->events[3 * i + 2]));
        GOF_HIP(hipEventElapsedTime(&ms, p->events[3 * i], p->events[3 * i + 1]));
        a += ms;
        GOF_HIP(hipEventElapsedTime(&ms, p->events[3 * i + 1], p->events[3 * i + 2]));
        b += ms;
    }
    *detector_ms = a;
    *morph_ms = b;
    *n_applies = (int)p->prof_used;
    p->prof_used = 0;
    return GOF_OK;
}

int gof_pipeline_profile_span(gof_pipeline *p, double *span_ms)
{
    if (!p || !span_ms)
        return fail(GOF_EINVAL, "NULL argument");
    *span_ms = 0.0;
    if (p->prof_used == 0)
        return GOF_OK;
    GOF_HIP(hipSetDevice(p->ctx->device));
    float ms = 0;
    GOF_HIP(hipEventSynchronize(p->events[3 * (p->prof_used - 1) + 2]));
    GOF_HIP(hipEventElapsedTime(&ms, p->events[0], p->events[3 * (p->prof_used - 1) + 2]));
    *span_ms = ms;
    return GOF_OK;
}

int gof_pipeline_profile_read(gof_pipeline *p, float *detector_ms, float *morph_ms, int capacity,
                              int *n_applies)
{
    if (!p || !detector_ms || !morph_ms || !n_applies || capacity < 0)
        return fail(GOF_EINVAL, "bad argument");
    GOF_HIP(hipSetDevice(p->ctx->device));
    for (size_t i = 0; i < p->prof_used; i++) {
        GOF_HIP(hipEventSynchronize(p->events[3 * i + 2]));
        if (i >= (size_t)capacity)
            continue;
        GOF_HIP(hipEventElapsedTime(&detector_ms[i], p->events[3 * i], p->events[3 * i + 1]));
        GOF_HIP(hipEventElapsedTime(&morph_ms[i], p->events[3 * i + 1], p->events[3 * i + 2]));
    }
    *n_applies = (int)p->prof_used;
    p->prof_used = 0;
    return GOF_OK;
}

int gof_pipeline_mean_modes(gof_pipeline *p, double *mean_modes)
{
    if (!p || !mean_modes)
        return fail(GOF_EINVAL, "NULL argument");
    *mean_modes = 0.0;
    if (!p->mog2)
        return GOF_OK;
    GOF_HIP(hipSetDevice(p->ctx->device));
    GOF_HIP(drain_applies(p->mog2));
    hipStream_t s = p->ctx->stream;
    const size_t n = (size_t)p->d.streams * p->d.h * p->d.w;
    unsigned long long sum = 0;
    GOF_HIP(hipMemsetAsync(p->d_sum, 0, sizeof(sum), s));
    GOF_HIP(gofk::sum_modes(p->mog2->view().modes, n, p->d_sum, s));
    GOF_HIP(hipMemcpyAsync(&sum, p->d_sum, sizeof(sum), hipMemcpyDeviceToHost, s));
    GOF_HIP(hipStreamSynchronize(s));
    *mean_modes = (double)sum / (double)n;
    return GOF_OK;
}

int gof_pipeline_algorithmic_bytes(gof_pipeline *p, double mean_modes, int n_frames,
                                   double *total_bytes, double *detector_bytes)
{
    if (!p || !total_bytes || !detector_bytes || n_frames < 1)
        return fail(GOF_EINVAL, "bad argument");
    const gof_pipeline_desc &d = p->d;
    const double npx = (double)d.streams * d.h * d.w;
    const double T = n_frames;
    // detector: every frame in and every mask out; the MOG2 model moves once per
    // launch of up to kMaxT frames (temporal blocking), FD state once per frame
    // (a converting pipeline whose 3x3 blur is fused reads the colour frames in the detector kernel)
    const bool cvt_fused = p->cvt_c && d.c == 1 &&
                           ((p->mog2 && p->fuse_blur3) || (p->fd && p->fd_mode == 1 && n_frames > 1));
    double det = T * ((cvt_fused ? p->cvt_c : d.c) + 1.0);
    if (p->mog2) {
        const double launches = (double)((n_frames + gofk::kMaxT - 1) / gofk::kMaxT);
        det += launches * 2.0 * (1.0 + 4.0 * (2 + d.c) * mean_modes);   // model read + write
    } else {
        det += T * 2.0 * d.c;                                  // X_{t-1} read, X_t kept
    }
    double tot = det;
    if (p->cvt_c && !cvt_fused)
        tot += T * (p->cvt_c + 1.0);                           // step 0 as its own kernel
    if (d.blur != GOF_BLUR_NONE && !p->fuse_blur3 && !(p->fd && p->fd_mode == 1) && !(p->knn && p->taps.fused3_ok))
        tot += T * 2.0 * d.c;                                  // blurred image written + read
    const int morph_kernels = d.n_morph == 0 ? 0 : (p->chain_ok ? 1 : d.n_morph);
    tot += T * 2.0 * morph_kernels;                            // each reads a mask, writes a mask
    *total_bytes = npx * tot;
    *detector_bytes = npx * det;
    return GOF_OK;
}

}  // extern "C"
