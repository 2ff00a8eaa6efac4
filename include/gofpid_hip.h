/*
 * gofpid_hip.h -- C ABI of libgofpid_hip.so, the MI355X (gfx950) implementation
 * of PyGOFPID's per-frame foreground path (steps 1-3 of GOFPID.detect).
 *
 * The reference has no FFI of its own: its plug-in surface is Python duck
 * typing over cv2 calls.  Each entry point below replaces one of those calls
 * and cites the reference line that makes it; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature (a stream is passed as
 *     void*; NULL means the context's own stream).
 *   - images are dense, C-contiguous uint8: frames (h, w, c) interleaved,
 *     masks (h, w).  "batch" objects take `streams` such images back to back.
 *   - every function returns GOF_OK (0) or an error code; gof_last_error()
 *     gives the message for the calling thread.  GOF_EINVAL is what the Python
 *     layer turns into ValueError, everything else into RuntimeError.
 *   - *_host entry points borrow host buffers for the duration of the call and
 *     are synchronous.  *_dev entry points take device pointers, enqueue on the
 *     stream and return without synchronising.
 *   - streams: `stream == NULL` means the CONTEXT's own stream (gof_stream(ctx), created
 *     non-blocking), NOT HIP's legacy default stream 0: a caller whose producer
 *     kernels run on stream 0 must either pass gof_stream(ctx) to them or
 *     synchronise before the call.  All *_dev calls on ONE object (gof_mog2, gof_fd,
 *     gof_pipeline) must be issued on ONE stream, or the caller orders the streams
 *     itself: the objects keep device state and scratch that successive calls
 *     read and write without further synchronisation.  Objects are complete when
 *     their create call returns (the zero-fill of a model is waited for), and the
 *     host-visible model accesses (export, import, state_copy, mean_modes) first wait
 *     for the stream the last apply was enqueued on.
 *   - device buffers need no particular alignment and no slack: no kernel reads or
 *     writes a byte outside the buffers' stated sizes.  The fast paths need frames
 *     whose base, row pitch (w*c) and image size are multiples of 4 bytes (true for
 *     every common geometry); anything else runs a slower general path with the
 *     same results.
 *   - a context is bound to one device and is NOT thread-safe; distinct
 *     contexts are independent (the reference runs one GOFPID per video,
 *     examples/dataset_ilids.py:126-140).
 */
#ifndef GOFPID_HIP_H
#define GOFPID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: blob extraction (gof_blobs_*), gof_pipeline_set_convert, per-apply profile series; the
 *    device layout behind gof_mog2_device_state / state_copy changed (a model buffer of
 *    version 1 must not be copied into a version 2 library: re-export / import instead) */
#define GOF_ABI_VERSION 3

enum {
    GOF_OK = 0,
    GOF_EINVAL = 1,   /* bad shape / parameter           -> ValueError   */
    GOF_EHIP = 2,     /* a HIP runtime call failed       -> RuntimeError */
    GOF_ENOMEM = 3,   /* device or host allocation failed                */
    GOF_ENODEV = 4    /* no usable gfx950 device                         */
};

enum { GOF_ERODE = 0, GOF_DILATE = 1 };
enum { GOF_DET_MOG2 = 0, GOF_DET_FD = 1, GOF_DET_KNN = 2 };
enum { GOF_BLUR_NONE = 0, GOF_BLUR_GAUSS = 1, GOF_BLUR_BOX = 2 };

#define GOF_MAX_MODES 5      /* cv2 MOG2 default nmixtures                 */
#define GOF_MAX_MORPH 8      /* ops in one pipeline's mat_morph list       */
#define GOF_MAX_KSIZE 31     /* largest blur / structuring-element side    */

typedef struct gof_ctx gof_ctx;
typedef struct gof_mog2 gof_mog2;
typedef struct gof_knn gof_knn;
typedef struct gof_fd gof_fd;
typedef struct gof_pipeline gof_pipeline;

/* cv.createBackgroundSubtractorMOG2() defaults (pipeline.py:232). */
typedef struct gof_mog2_params {
    int history;              /* 500  */
    int nmixtures;            /* 5    */
    float var_threshold;      /* 16   Tb */
    float background_ratio;   /* 0.9  TB */
    float var_threshold_gen;  /* 9    Tg */
    float var_init;           /* 15   */
    float var_min;            /* 4    */
    float var_max;            /* 75   */
    float ct;                 /* 0.05 */
    float tau;                /* 0.5  */
    int detect_shadows;       /* 1    */
    int shadow_value;         /* 127  */
} gof_mog2_params;

/* cv.createBackgroundSubtractorKNN() defaults [OPENCV-RECALL bgfg_KNN.cpp]. */
typedef struct gof_knn_params {
    int history;              /* 500  */
    int nsamples;             /* 7    samples per set (short / mid / long term) */
    float dist2_threshold;    /* 400  */
    int knn;                  /* 2    flagged neighbours that make a pixel background */
    float tau;                /* 0.5  shadow threshold */
    int detect_shadows;       /* 1    */
    int shadow_value;         /* 127  */
} gof_knn_params;

/* One mat_morph entry (pipeline.py:177-186, 473-475). */
typedef struct gof_morph_op {
    int op;                   /* GOF_ERODE | GOF_DILATE                    */
    int kh, kw;               /* structuring-element size                  */
    const uint8_t *kernel;    /* (kh, kw) uint8, non-zero = member; NULL = all ones */
} gof_morph_op;

/* Steps 1-3 of GOFPID.detect as one object (pipeline.py:461-476). */
typedef struct gof_pipeline_desc {
    int streams;              /* independent video streams stepped together (>= 1) */
    int h, w, c;              /* frame geometry, c in {1, 3, 4}            */
    int blur;                 /* GOF_BLUR_NONE | GOF_BLUR_GAUSS | GOF_BLUR_BOX */
    int blur_kw, blur_kh;     /* odd                                        */
    double blur_sigma_x, blur_sigma_y;
    int detector;             /* GOF_DET_MOG2 | GOF_DET_FD | GOF_DET_KNN (cv2 defaults) */
    gof_mog2_params mog2;
    double fd_threshold;
    int n_morph;
    gof_morph_op morph[GOF_MAX_MORPH];
} gof_pipeline_desc;

/* ---- library / context ------------------------------------------------- */

int gof_abi_version(void);
const char *gof_strerror(int code);
const char *gof_last_error(void);
int gof_device_count(int *count);

/* One context per (process, device): owns a HIP stream and scratch buffers. */
int gof_create(int device, gof_ctx **out);
void gof_destroy(gof_ctx *ctx);
int gof_synchronize(gof_ctx *ctx);
void *gof_stream(gof_ctx *ctx);        /* the context's hipStream_t */

/* ---- step 0 (optional): colour conversion (replaces cv.cvtColor(X, code) for
 *      COLOR_BGR2GRAY / COLOR_RGB2GRAY, pipeline.py:458-459) ---------------- */
int gof_cvt_gray_host(gof_ctx *ctx, const uint8_t *src, uint8_t *dst, int h, int w, int c,
                      int rgb_order);
int gof_cvt_gray_dev(gof_ctx *ctx, const void *d_src, void *d_dst, int n_images, int h, int w,
                     int c, int rgb_order, void *stream);

/* ---- step 1: blur  (replaces cv.GaussianBlur, pipeline.py:463-467) ------ */
/* cv.GaussianBlur(src, (kw, kh), sigma_x, sigmaY = sigma_y) on uint8 with
 * BORDER_REFLECT_101.  NB the reference passes blur['borderType'] as the 3rd
 * positional argument, which cv2 binds to sigmaX. */
int gof_gaussian_blur_host(gof_ctx *ctx, const uint8_t *src, uint8_t *dst, int h, int w, int c,
                           int kw, int kh, double sigma_x, double sigma_y);
int gof_gaussian_blur_dev(gof_ctx *ctx, const void *d_src, void *d_dst, int n_images, int h,
                          int w, int c, int kw, int kh, double sigma_x, double sigma_y,
                          void *stream);
/* cv.blur(src, (kw, kh)) on uint8: normalised box filter, BORDER_REFLECT_101,
 * out = round-half-even(sum / (kw*kh)).  The reference's tests call it as
 * blur['fun'](X, ksize, borderType) (tests/test_pipeline.py:68-75): cv2 binds that
 * third positional argument to `dst`, so the requested border is ignored.
 * Only odd kw / kh are accepted (GOF_EINVAL otherwise): cv2 also takes even sizes
 * (anchor at k/2), which the reference never uses (3x3, tests/test_pipeline.py:68-75). */
int gof_box_blur_host(gof_ctx *ctx, const uint8_t *src, uint8_t *dst, int h, int w, int c, int kw,
                      int kh);
int gof_box_blur_dev(gof_ctx *ctx, const void *d_src, void *d_dst, int n_images, int h, int w,
                     int c, int kw, int kh, void *stream);
/* the 8.8 fixed-point taps the blur uses (host only, for tests) */
int gof_gauss_kernel_q8(int n, double sigma, uint16_t *taps);

/* ---- step 2a: MOG2  (replaces cv.createBackgroundSubtractorMOG2() and its
 *      .apply(X), pipeline.py:232 and :470) --------------------------------- */
void gof_mog2_default_params(gof_mog2_params *p);
int gof_mog2_create(gof_ctx *ctx, int streams, int h, int w, int c, const gof_mog2_params *p,
                    gof_mog2 **out);
void gof_mog2_destroy(gof_mog2 *m);
int gof_mog2_apply_host(gof_mog2 *m, const uint8_t *frames, uint8_t *masks);
int gof_mog2_apply_dev(gof_mog2 *m, const void *d_frames, void *d_masks, void *stream);
int gof_mog2_nframes(const gof_mog2 *m);
/* Model as structure-of-arrays planes, n = streams*h*w: weight[K][n], var[K][n],
 * mean[K][c][n] float32, modes[n] uint8 (K = nmixtures).  Entries of slots >=
 * modes[i] are dead storage.  (On the device a pixel's variance and means sit side by
 * side, one 16-byte access per lane; export / import convert.)  export -> host,
 * import <- host: parity tests, checkpoint/restore and warm start of a replica. */
int gof_mog2_export(gof_mog2 *m, float *weight, float *var, float *mean, uint8_t *modes);
int gof_mog2_import(gof_mog2 *m, const float *weight, const float *var, const float *mean,
                    const uint8_t *modes, int nframes);
/* raw device view of the same planes (one allocation) for an RCCL broadcast */
int gof_mog2_device_state(gof_mog2 *m, void **d_state, size_t *bytes);
int gof_mog2_set_nframes(gof_mog2 *m, int nframes);
/* copy the whole model to (to_model == 0) or from (to_model != 0) a caller-owned
 * device buffer of gof_mog2_device_state()'s size; synchronous */
int gof_mog2_state_copy(gof_mog2 *m, void *d_buf, int to_model);

/* ---- step 2c: KNN  (replaces cv.createBackgroundSubtractorKNN() and its .apply(X),
 *      pipeline.py:233-234 and :470; exercised by tests/test_pipeline.py:98-109) ------------
 * cv2's model (three sets of nsamples samples per pixel, sample = channel bytes + flag byte),
 * rule (kNN flagged neighbours within dist2Threshold -> background, shadow test with tau, else
 * foreground) and staggered refresh.  cv2 draws every pixel's next refresh slot with rand() in a
 * row-parallel loop, so its own masks are not reproducible run to run; here the draw is a hash of
 * (pixel, frame, set) -- identical results on every run and bit-identical to the CPU restatement
 * this repository tests against; parity with cv2 itself is unpinnable by construction.
 * frames: streams x (h, w, c) uint8; masks: streams x (h, w) in {0, shadow_value, 255}. */
void gof_knn_default_params(gof_knn_params *p);
int gof_knn_create(gof_ctx *ctx, int streams, int h, int w, int c, const gof_knn_params *p, gof_knn **out);
void gof_knn_destroy(gof_knn *k);
int gof_knn_apply_host(gof_knn *k, const uint8_t *frames, uint8_t *masks);
int gof_knn_apply_dev(gof_knn *k, const void *d_frames, void *d_masks, void *stream);
int gof_knn_nframes(const gof_knn *k);
/* the model as byte planes (host buffers), n = streams*h*w: samples[3 nsamples][c + 1][n] (sets in
 * the order short, mid, long; last byte of a sample = its flag), state[6][n] = slot of the next
 * short / mid / long refresh, write index of the short / mid / long set */
int gof_knn_export(gof_knn *k, uint8_t *samples, uint8_t *state);

/* ---- step 2b: frame differencing (replaces FrameDifferencing.apply,
 *      segmentation.py:40-70) ---------------------------------------------- */
int gof_fd_create(gof_ctx *ctx, int streams, int h, int w, int c, double threshold, gof_fd **out);
void gof_fd_destroy(gof_fd *f);
int gof_fd_apply_host(gof_fd *f, const uint8_t *frames, uint8_t *masks);
int gof_fd_apply_dev(gof_fd *f, const void *d_frames, void *d_masks, void *stream);

/* ---- step 3: morphology (replaces cv.erode / cv.dilate, pipeline.py:475) - */
int gof_morph_host(gof_ctx *ctx, int op, const uint8_t *src, uint8_t *dst, int h, int w,
                   const uint8_t *kernel, int kh, int kw);
int gof_morph_dev(gof_ctx *ctx, int op, const void *d_src, void *d_dst, int n_images, int h,
                  int w, const uint8_t *kernel, int kh, int kw, void *stream);

/* ---- steps 1-3 fused (replaces pipeline.py:461-476 as a whole) ---------- */
int gof_pipeline_create(gof_ctx *ctx, const gof_pipeline_desc *desc, gof_pipeline **out);
void gof_pipeline_destroy(gof_pipeline *p);
/* frames: streams x (h, w, c); masks: streams x (h, w) */
int gof_pipeline_apply_host(gof_pipeline *p, const uint8_t *frames, uint8_t *masks);
int gof_pipeline_apply_dev(gof_pipeline *p, const void *d_frames, void *d_masks, void *stream);
/* n_frames consecutive frames of every stream in one call (temporal blocking:
 * the MOG2 model is read and written once per 8 frames instead of once per
 * frame; the masks are identical to n_frames single-frame calls).
 * frames: n_frames x streams x (h, w, c); masks: n_frames x streams x (h, w). */
int gof_pipeline_apply_host_many(gof_pipeline *p, const uint8_t *frames, uint8_t *masks,
                                 int n_frames);
int gof_pipeline_apply_dev_many(gof_pipeline *p, const void *d_frames, void *d_masks, int n_frames,
                                void *stream);
/* Step 0 inside the pipeline (replaces cv.cvtColor(X, COLOR_BGR2GRAY | COLOR_RGB2GRAY) in
 * front of it, pipeline.py:458-459): frames arrive with c_in (3 or 4) channels and are
 * converted to grey on the device before step 1, with no round trip through the host.
 * The pipeline must have been created with c == 1; every apply then takes
 * streams x (h, w, c_in) frames.  c_in == 0 switches the conversion off again. */
int gof_pipeline_set_convert(gof_pipeline *p, int c_in, int rgb_order);
gof_mog2 *gof_pipeline_mog2(gof_pipeline *p);
gof_knn *gof_pipeline_knn(gof_pipeline *p);    /* the pipeline's KNN model, or NULL */
/* mean number of Gaussian modes in use per pixel (m-bar of the byte model) */
int gof_pipeline_mean_modes(gof_pipeline *p, double *mean_modes);
/* Algorithmic HBM bytes of one apply of n_frames frames, given m-bar:
 *   *detector_bytes: the dominant kernel alone (fused blur + MOG2), per pixel
 *       n_frames*(c + 1) + ceil(n_frames/8) * 2*(1 + 4*(2+c)*mbar)
 *       = 6 + 40*mbar for one BGR frame
 *   *total_bytes: every kernel of the step (adds 2 B/px/frame per morphology
 *       kernel and 2*c B/px/frame for a blur that is not fused). */
int gof_pipeline_algorithmic_bytes(gof_pipeline *p, double mean_modes, int n_frames,
                                   double *total_bytes, double *detector_bytes);

/* ---- step 4 helper, host only (NOT part of the GPU hot path) -------------
 * cv.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE) as the reference's
 * helpers.find_contours calls it (helpers.py:139-143): Suzuki-Abe border
 * following on the host.  Returns library-allocated arrays (release each with
 * gof_free): points_xy = (x, y) int32 pairs of all contours back to back,
 * offsets[k]..offsets[k+1] = the point range of contour k. */
int gof_find_external_contours(const uint8_t *mask, int h, int w, int32_t **points_xy,
                               int32_t **offsets, int *n_contours);
void gof_free(void *p);

/* ---- step 4 front end on the GPU: external contours + blob statistics ----------
 * Replaces helpers.find_contours(mask) = cv.findContours(mask, RETR_EXTERNAL,
 * CHAIN_APPROX_NONE) [helpers.py:125-150], the filter cv.contourArea(contour) >=
 * area_min [pipeline.py:555-572] and the cv.moments / np.max evaluations of
 * get_bottom / get_center [helpers.py:153-168]: only the records and the contour
 * points of blobs that pass the filter leave the device (KBs, not the 2 MB mask).
 * Results equal gof_find_external_contours + the same filter: same contours, same
 * order, same points.  a00/a10/a01 are the exact integer shoelace sums of the contour
 * polygon: cv.contourArea = |a00| / 2; cv.moments: s = a00 < 0 ? -1 : 1, m00 = s a00 / 2,
 * m10 = s a10 / 6, m01 = s a01 / 6; cv.boundingRect = (xmin, ymin, xmax - xmin + 1,
 * ymax - ymin + 1). */
typedef struct gof_blobs gof_blobs;
typedef struct gof_blob_record {
    int32_t x0, y0;                     /* first contour point (raster-first pixel) */
    int32_t xmin, ymin, xmax, ymax;
    int32_t n_points, point_offset;     /* its points: points_xy[2*point_offset ...] */
    int64_t a00, a10, a01;
} gof_blob_record;
enum {                                  /* gof_blobs_counts flags: the image's result is not */
    GOF_BLOBS_CRACK_OVERFLOW = 1,       /* available (mask too intricate for the scratch);    */
    GOF_BLOBS_BLOB_OVERFLOW = 2,        /* the caller falls back to the mask itself            */
    GOF_BLOBS_POINT_OVERFLOW = 4,
    GOF_BLOBS_INTERNAL = 8              /* a device-side loop bound was hit (a bug: report it) */
};
/* scratch for n_images masks of (h, w); crack_capacity = border-edge slots per image
 * (0: h*w/2, enough for any mask a morphological open/close leaves) */
int gof_blobs_create(gof_ctx *ctx, int n_images, int h, int w, int crack_capacity, gof_blobs **out);
void gof_blobs_destroy(gof_blobs *b);
/* Enqueue the extraction for n_images masks (device, back to back) with
 * cv.contourArea >= min_area, and the copy of the result heads to the host. */
int gof_blobs_extract_dev(gof_blobs *b, const void *d_masks, int n_images, double min_area,
                          void *stream);
/* The same from host masks (n_images x (h, w) uint8, uploaded first); synchronous. */
int gof_blobs_extract_host(gof_blobs *b, const uint8_t *masks, int n_images, double min_area);
/* Wait for the last extraction; counts and flags of one image. */
int gof_blobs_counts(gof_blobs *b, int image, int *n_blobs, int *n_points, int *flags);
/* Records (n_blobs of them) and contour points ((x, y) int32 pairs, n_points of them,
 * contour k = points [point_offset, point_offset + n_points) of record k). */
int gof_blobs_fetch(gof_blobs *b, int image, gof_blob_record *records, int max_records,
                    int32_t *points_xy, int max_points);
/* steps 1-3 for n_frames frames of every stream from host buffers + the extraction on
 * the resulting masks, which stay on the device (gof_pipeline_fetch_masks downloads them):
 * image index = frame * streams + stream.  b must hold n_frames * streams images. */
int gof_pipeline_apply_host_blobs(gof_pipeline *p, const uint8_t *frames, int n_frames,
                                  gof_blobs *b, double min_area);
/* the same with the n_frames frames (each streams x (h, w, c)) in separate host buffers:
 * no host-side gathering copy (a video reader hands out one array per frame) */
int gof_pipeline_apply_host_blobs_ptrs(gof_pipeline *p, const uint8_t *const *frames, int n_frames,
                                       gof_blobs *b, double min_area);
/* The same, asynchronously, for a caller that pipelines batches (GOFPID.detect_many: the frames of
 * batch k + 1 go up while the kernels of batch k run and the host works on the records of batch k):
 * enqueue a batch into slot 0 or 1 -- returns once the frames have left the caller's buffers (the
 * upload runs on a copy stream of the library beside the kernels of the batch submitted before);
 * steps 1-3 and the extraction into b are enqueued behind it, not waited for.  Batches must be
 * submitted in stream order by one thread at a time, alternating slots, each slot with a gof_blobs
 * of its own; gof_pipeline_wait_slot may be called from another thread.  Replaces nothing in the
 * reference (its detect() is one synchronous frame, pygofpid/pipeline.py:434-493). */
int gof_pipeline_submit_host_blobs_ptrs(gof_pipeline *p, const uint8_t *const *frames, int n_frames,
                                        gof_blobs *b, double min_area, int slot);
/* Wait for the batch submitted into the slot: its records are then in its gof_blobs (and
 * gof_pipeline_fetch_masks returns the masks of the batch submitted last). */
int gof_pipeline_wait_slot(gof_pipeline *p, int slot);
int gof_pipeline_fetch_masks(gof_pipeline *p, uint8_t *masks, int n_frames);
/* ... of the batch that went through the given slot last */
int gof_pipeline_fetch_masks_slot(gof_pipeline *p, uint8_t *masks, int n_frames, int slot);

/* Per-kernel timing with HIP events on the launch stream (what bench.py's
 * roofline block is computed from).  enable != 0: every following apply
 * brackets its detector kernel (fused blur + MOG2, or FD) and its morphology
 * kernel(s) with event pairs (at most `capacity` applies are kept).  The query
 * synchronises the recorded events and returns the SUM of elapsed ms and the
 * number of applies measured, then clears the record. */
int gof_pipeline_profile_enable(gof_pipeline *p, int enable, int capacity);
/* Bracket only every stride-th apply (stride >= 1, default 1): a sample over the
 * timed region that costs the stream fewer event records. */
int gof_pipeline_profile_stride(gof_pipeline *p, int stride);
int gof_pipeline_profile_query(gof_pipeline *p, double *detector_ms, double *morph_ms,
                               int *n_applies);
/* Elapsed ms from the first bracketed apply's first event to the last one's last
 * event (with the number of applies: the period of an apply in bracketed mode, which
 * minus the un-bracketed period is what the event markers themselves cost).  Call
 * before profile_query / profile_read, which clear the record. */
int gof_pipeline_profile_span(gof_pipeline *p, double *span_ms);
/* The same record apply by apply: fills detector_ms[i] / morph_ms[i] for the first
 * min(capacity, measured) applies, returns the number measured in *n_applies and
 * clears the record (per-launch series: medians, ramps). */
int gof_pipeline_profile_read(gof_pipeline *p, float *detector_ms, float *morph_ms, int capacity,
                              int *n_applies);

#ifdef __cplusplus
}
#endif
#endif /* GOFPID_HIP_H */
