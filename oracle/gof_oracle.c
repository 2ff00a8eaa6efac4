/*
 * gof_oracle.c -- CPU restatement of PyGOFPID's per-frame foreground path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pygofpid_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline.
 *
 * PARITY UNPINNED.  The reference delegates all arithmetic on this path to the
 * third-party package opencv-python (pinned ">=3.4.0,<4.0.0" in
 * /root/reference/setup.py:14, "^3.4.0" in pyproject.toml:23), which is absent
 * from /root/reference, from this image and from the GPU box.  The reference's
 * own tests assert no numerical result of blur / MOG2 / erode / dilate
 * (SURVEY.md section 4, 8c).  What follows restates OpenCV's published
 * algorithms from recollection of its public sources ("[OPENCV-RECALL]"); the
 * only value pins available are (i) FrameDifferencing's first frame == zeros
 * (reference tests/test_segmentation.py:19), (ii) hand-derived known-answer
 * vectors and (iii) independent scipy.ndimage cross-checks in
 * tests/test_oracle.py.
 *
 * OPENCV VERSION EACH CHOICE ASSUMES (all from recollection of the public sources;
 * nothing below could be run against cv2 here -- this is the list a maintainer with
 * cv2 checks FIRST, most output-changing item first):
 *   1. GaussianBlur, 8-bit, taps for (3,3), sigma 4:  [84, 88, 84]  (sum 256)
 *        = the fixed-point path WITH error-diffused tap rounding
 *        (getGaussianKernelFixedPoint_ED), as in 3.4.9+ / 4.2+.
 *        3.4.1 .. 3.4.8 (fixed point, every tap rounded on its own) give
 *        [84, 87, 84] (sum 255): constant images come out one grey level darker
 *        where 255*c/256 rounds down.  3.4.0 ran the float filter engine
 *        (cvRound of a float convolution): differs in the last bit on ~half the
 *        pixels.  Probe: cv.GaussianBlur(np.full((8,8),200,np.uint8),(3,3),4)
 *        -> 200 everywhere iff error-diffused.  tests/test_oracle.py pins the choice.
 *   2. cvtColor BGR/RGB -> GRAY: shift 14, (B,G,R) = (1868, 9617, 4899), the 3.4 line
 *        (RGB2Gray<uchar> table).  4.x SIMD path: shift 15, (3735, 19235, 9798); the
 *        two differ by one grey level on a few percent of inputs.  Probe: the pixel
 *        (B,G,R) = (1,2,253) and a sweep over one channel.
 *   3. cv.blur (box): rint(sum * (1.0/area)) in double (ColumnSum<int,uchar>).  Later
 *        3.4.x / 4.x use an integer ColumnSum<ushort,uchar> = (sum + area/2) / area:
 *        identical for every ODD area (no ties exist), which is all this library
 *        accepts; even sizes (refused here) are where the two differ.
 *   4. MOG2 weight renormalisation guarded by |totalWeight| > FLT_EPSILON (4.x and late
 *        3.4.x).  Older 3.4 divides unguarded: only differs when every mode of a pixel
 *        was pruned in one frame (totalWeight == 0 -> inf/NaN there), which the default
 *        rates cannot produce (prune = -0.05 alpha, weights sum to ~1).
 *   5. MOG2 shadow test runs on the model AFTER the frame's update, mask 127; learning
 *        rate 1/min(2 n, history) in double: unchanged across 3.4 / 4.x.
 *   6. erode / dilate with BORDER_CONSTANT, borderValue = morphologyDefaultBorderValue
 *        (never wins a min / max): unchanged across versions.
 *
 * Reference call sites restated here:
 *   pipeline.py:462-467   blur['fun'](X, ksize, borderType)
 *                         default cv.GaussianBlur(X,(3,3),4): the 3rd positional
 *                         argument of GaussianBlur is sigmaX, so BORDER_DEFAULT
 *                         (=4) arrives as sigma 4.0 and the border stays
 *                         REFLECT_101.                       -> gofo_gaussian_blur
 *   pipeline.py:232,470   cv.createBackgroundSubtractorMOG2().apply(X)
 *                                                            -> gofo_mog2_apply
 *   pipeline.py:233-234   cv.createBackgroundSubtractorKNN().apply(X)
 *                         (cv2's own output is not reproducible: rand() in a
 *                         row-parallel loop; see the section)  -> gofo_knn_apply
 *   segmentation.py:40-70 FrameDifferencing.apply            -> gofo_fd_apply
 *   pipeline.py:473-475   d['fun'](mask, d['kernel']) with cv.erode / cv.dilate
 *                                                            -> gofo_morph
 *   pipeline.py:461-476   the sequence of the above          -> gofo_pipeline_*
 *
 * Build:  make -C oracle   (gcc, -ffp-contract=off so that a*b+c is never
 * fused: the MOG2 update must round after every operation like OpenCV's scalar
 * C++ built for baseline x86-64).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define GOFO_MAX_MODES 5   /* OpenCV defaultNMixtures2 */
#define GOFO_MAX_CH 4

/* ------------------------------------------------------------------------- */
/* borders                                                                   */
/* ------------------------------------------------------------------------- */

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) [OPENCV-RECALL]:
 * gfedcb|abcdefgh|gfedcba ; len == 1 -> 0. */
static int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    }
    return p;
}

int gofo_reflect101(int p, int len) { return reflect101(p, len); }

/* ------------------------------------------------------------------------- */
/* cv.cvtColor(X, COLOR_BGR2GRAY | COLOR_RGB2GRAY)  [REF pipeline.py:458-459]  */
/* OpenCV 3.4 (the line the reference pins) [OPENCV-RECALL]: yuv_shift = 14,   */
/* R2Y = 4899, G2Y = 9617, B2Y = 1868, CV_DESCALE rounding.                    */
/* ------------------------------------------------------------------------- */
void gofo_cvt_gray(const uint8_t *src, uint8_t *dst, int h, int w, int c, int rgb_order)
{
    size_t n = (size_t)h * w;
    int c0 = rgb_order ? 4899 : 1868, c2 = rgb_order ? 1868 : 4899;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = src + i * c;
        dst[i] = (uint8_t)((p[0] * c0 + p[1] * 9617 + p[2] * c2 + (1 << 13)) >> 14);
    }
}

/* ------------------------------------------------------------------------- */
/* Gaussian blur, 8-bit fixed point   [OPENCV-RECALL]          */
/* ------------------------------------------------------------------------- */

/*
 * 8.8 fixed-point Gaussian taps as OpenCV's 8-bit path builds them
 * (getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED): float taps
 * g_i ~ exp(-x^2/(2 sigma^2)) normalised to 1; side taps rounded to 1/256 with
 * the rounding error carried to the next tap ("error diffusion"); the centre
 * tap is 256 - 2*sum(side).  sigma <= 0: sigma = 0.3*((n-1)*0.5-1)+0.8, except
 * n in {1,3,5,7,9} which use OpenCV's fixed binomial-like tables.
 * Returns 0 on success.  n must be odd, 1 <= n <= 255.
 */
int gofo_gauss_kernel_q8(int n, double sigma, uint16_t *taps)
{
    if (n < 1 || (n & 1) == 0 || n > 255)
        return -1;
    if (sigma <= 0) {
        static const uint16_t t1[] = {256};
        static const uint16_t t3[] = {64, 128, 64};
        static const uint16_t t5[] = {16, 64, 96, 64, 16};
        static const uint16_t t7[] = {8, 28, 56, 72, 56, 28, 8};
        static const uint16_t t9[] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
        const uint16_t *t = NULL;
        switch (n) {
        case 1: t = t1; break;
        case 3: t = t3; break;
        case 5: t = t5; break;
        case 7: t = t7; break;
        case 9: t = t9; break;
        default: break;
        }
        if (t) {
            memcpy(taps, t, (size_t)n * sizeof(uint16_t));
            return 0;
        }
        sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    }
    int half = n / 2;
    double g[256];
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - half;
        g[i] = exp(scale2x * x * x);
        sum += g[i];
    }
    double err = 0;
    long acc = 0;
    for (int i = 0; i < half; i++) {
        double adj = g[i] / sum * 256.0 + err;
        long v = lrint(adj); /* cvRound: round-half-even */
        err = adj - (double)v;
        taps[i] = (uint16_t)v;
        taps[n - 1 - i] = (uint16_t)v;
        acc += 2 * v;
    }
    taps[half] = (uint16_t)(256 - acc);
    return 0;
}

/*
 * Separable 8-bit blur with 8.8 taps: horizontal pass exact in 16 bits,
 * vertical pass exact in 32 bits, ONE rounding:
 *     dst = (sum_j ky[j] * (sum_i kx[i] * src[y+j][x+i]) + 2^15) >> 16
 * Border REFLECT_101.  src/dst are (h, w, c) interleaved, dst != src.
 * Works for any taps with sum == 256 (so the box filter of cv.blur does NOT
 * go through here: it is not a fixed-point filter in OpenCV).
 */
void gofo_sepfilter_q8(const uint8_t *src, uint8_t *dst, int h, int w, int c,
                       const uint16_t *kx, int nkx, const uint16_t *ky, int nky)
{
    int rx = nkx / 2, ry = nky / 2;
    size_t rowlen = (size_t)w * c;
#pragma omp parallel
    {
        uint16_t *hrows = (uint16_t *)malloc(sizeof(uint16_t) * rowlen * (size_t)nky);
#pragma omp for schedule(static)
        for (int y = 0; y < h; y++) {
            for (int j = 0; j < nky; j++) {
                const uint8_t *s = src + (size_t)reflect101(y + j - ry, h) * rowlen;
                uint16_t *hr = hrows + (size_t)j * rowlen;
                for (int x = 0; x < w; x++) {
                    for (int ch = 0; ch < c; ch++) {
                        uint32_t a = 0;
                        for (int i = 0; i < nkx; i++)
                            a += (uint32_t)kx[i] * s[(size_t)reflect101(x + i - rx, w) * c + ch];
                        hr[(size_t)x * c + ch] = (uint16_t)a;
                    }
                }
            }
            uint8_t *d = dst + (size_t)y * rowlen;
            for (size_t e = 0; e < rowlen; e++) {
                uint32_t a = 0;
                for (int j = 0; j < nky; j++)
                    a += (uint32_t)ky[j] * hrows[(size_t)j * rowlen + e];
                uint32_t v = (a + 32768u) >> 16;
                d[e] = (uint8_t)(v > 255 ? 255 : v);
            }
        }
        free(hrows);
    }
}

/* 3x3 fast form used by the timed CPU baseline: same arithmetic, rolling
 * horizontal rows instead of recomputing them per output row. */
static void blur3_rows(const uint8_t *src, uint8_t *dst, int h, int w, int c,
                       const uint16_t *kx, const uint16_t *ky, int y0, int y1)
{
    size_t rowlen = (size_t)w * c;
    uint16_t *hbuf = (uint16_t *)malloc(sizeof(uint16_t) * rowlen * 3);
    uint16_t *hr[3] = {hbuf, hbuf + rowlen, hbuf + 2 * rowlen};
    for (int y = y0 - 1; y <= y1; y++) {
        /* horizontal pass of source row reflect(y) into slot (y+1)%3 */
        const uint8_t *s = src + (size_t)reflect101(y, h) * rowlen;
        uint16_t *o = hr[(y + 1) % 3];
        for (int x = 0; x < w; x++) {
            int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            for (int ch = 0; ch < c; ch++)
                o[(size_t)x * c + ch] = (uint16_t)(kx[0] * s[(size_t)xl * c + ch] +
                                                   kx[1] * s[(size_t)x * c + ch] +
                                                   kx[2] * s[(size_t)xr * c + ch]);
        }
        int yo = y - 1; /* output row whose 3 inputs are now complete */
        if (yo >= y0 && yo < y1) {
            const uint16_t *a = hr[(yo) % 3], *b = hr[(yo + 1) % 3], *cc = hr[(yo + 2) % 3];
            uint8_t *d = dst + (size_t)yo * rowlen;
            for (size_t e = 0; e < rowlen; e++) {
                uint32_t v = ((uint32_t)ky[0] * a[e] + (uint32_t)ky[1] * b[e] +
                              (uint32_t)ky[2] * cc[e] + 32768u) >> 16;
                d[e] = (uint8_t)v;
            }
        }
    }
    free(hbuf);
}

/*
 * cv.GaussianBlur(src, (kw,kh), sigmaX, sigmaY=0) on uint8, border REFLECT_101.
 * sigmaY <= 0 -> sigmaY = sigmaX.  Returns 0, or -1 on bad ksize.
 */
int gofo_gaussian_blur(const uint8_t *src, uint8_t *dst, int h, int w, int c,
                       int kw, int kh, double sigma_x, double sigma_y)
{
    uint16_t kx[256], ky[256];
    if (sigma_y <= 0)
        sigma_y = sigma_x;
    if (gofo_gauss_kernel_q8(kw, sigma_x, kx) || gofo_gauss_kernel_q8(kh, sigma_y, ky))
        return -1;
    gofo_sepfilter_q8(src, dst, h, w, c, kx, kw, ky, kh);
    return 0;
}

/* cv.blur(src, (kw, kh)) on uint8: normalised box filter, REFLECT_101.
 * OpenCV sums the window in int and stores saturate_cast<uchar>(sum * scale)
 * with scale = 1.0 / (kw*kh) in double, i.e. cvRound (round-half-even) of the
 * double product [OPENCV-RECALL: box_filter ColumnSum<int, uchar>]. */
void gofo_box_blur(const uint8_t *src, uint8_t *dst, int h, int w, int c, int kw, int kh)
{
    int rx = kw / 2, ry = kh / 2;
    double scale = 1.0 / ((double)kw * (double)kh);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            for (int ch = 0; ch < c; ch++) {
                int sum = 0;
                for (int j = -ry; j <= ry; j++) {
                    const uint8_t *row = src + (size_t)reflect101(y + j, h) * w * c;
                    for (int i = -rx; i <= rx; i++)
                        sum += row[(size_t)reflect101(x + i, w) * c + ch];
                }
                long v = lrint((double)sum * scale);
                dst[((size_t)y * w + x) * c + ch] = (uint8_t)(v > 255 ? 255 : v);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* MOG2   [OPENCV-RECALL: bgfg_gaussmix2.cpp MOG2Invoker]      */
/* ------------------------------------------------------------------------- */

typedef struct gofo_mog2_params {
    int history;          /* 500  */
    int nmixtures;        /* 5    */
    float var_threshold;  /* Tb  16   : background test                     */
    float background_ratio; /* TB 0.9                                       */
    float var_threshold_gen; /* Tg 9  : fit test                            */
    float var_init;       /* 15 */
    float var_min;        /* 4  */
    float var_max;        /* 75 */
    float ct;             /* 0.05 complexity-reduction threshold            */
    float tau;            /* 0.5  shadow threshold                          */
    int detect_shadows;   /* 1    */
    int shadow_value;     /* 127  */
} gofo_mog2_params;

void gofo_mog2_default_params(gofo_mog2_params *p)
{
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

typedef struct gofo_gmm {
    float weight;
    float variance;
} gofo_gmm;

typedef struct gofo_mog2 {
    gofo_mog2_params p;
    int h, w, c;
    int nframes;
    gofo_gmm *gmm;      /* [h*w][nmixtures]           */
    float *mean;        /* [h*w][nmixtures][c]        */
    uint8_t *modes;     /* [h*w]                      */
} gofo_mog2;

gofo_mog2 *gofo_mog2_create(int h, int w, int c, const gofo_mog2_params *p)
{
    if (h <= 0 || w <= 0 || c <= 0 || c > GOFO_MAX_CH)
        return NULL;
    gofo_mog2 *m = (gofo_mog2 *)calloc(1, sizeof(*m));
    if (p)
        m->p = *p;
    else
        gofo_mog2_default_params(&m->p);
    if (m->p.nmixtures < 1 || m->p.nmixtures > GOFO_MAX_MODES) {
        free(m);
        return NULL;
    }
    m->h = h;
    m->w = w;
    m->c = c;
    size_t n = (size_t)h * w;
    m->gmm = (gofo_gmm *)calloc(n * m->p.nmixtures, sizeof(gofo_gmm));
    m->mean = (float *)calloc(n * m->p.nmixtures * c, sizeof(float));
    m->modes = (uint8_t *)calloc(n, 1);
    return m;
}

void gofo_mog2_destroy(gofo_mog2 *m)
{
    if (!m)
        return;
    free(m->gmm);
    free(m->mean);
    free(m->modes);
    free(m);
}

int gofo_mog2_nframes(const gofo_mog2 *m) { return m->nframes; }

/* Export the model as structure-of-arrays planes, the layout the HIP side
 * keeps in HBM:  weight[k][n], var[k][n], mean[k][ch][n], modes[n]. */
void gofo_mog2_export(const gofo_mog2 *m, float *weight, float *var, float *mean, uint8_t *modes)
{
    size_t n = (size_t)m->h * m->w;
    int K = m->p.nmixtures, C = m->c;
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < K; k++) {
            weight[(size_t)k * n + i] = m->gmm[i * K + k].weight;
            var[(size_t)k * n + i] = m->gmm[i * K + k].variance;
            for (int ch = 0; ch < C; ch++)
                mean[((size_t)k * C + ch) * n + i] = m->mean[(i * K + k) * C + ch];
        }
        modes[i] = m->modes[i];
    }
}

/* The inverse of gofo_mog2_export, plus the frame counter (which sets the learning rate): lets a
 * timing run or a test start from a trained model (bench.py feeds the CPU leg the state the GPU
 * leg runs at).  Planes as gofo_mog2_export writes them. */
void gofo_mog2_import(gofo_mog2 *m, const float *weight, const float *var, const float *mean,
                      const uint8_t *modes, int nframes)
{
    size_t n = (size_t)m->h * m->w;
    int K = m->p.nmixtures, C = m->c;
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < K; k++) {
            m->gmm[i * K + k].weight = weight[(size_t)k * n + i];
            m->gmm[i * K + k].variance = var[(size_t)k * n + i];
            for (int ch = 0; ch < C; ch++)
                m->mean[(i * K + k) * C + ch] = mean[((size_t)k * C + ch) * n + i];
        }
        m->modes[i] = modes[i];
    }
    m->nframes = nframes;
}

/* detectShadowGMM [OPENCV-RECALL]; runs on the model AFTER this frame's update. */
static int shadow_gmm(const float *data, int nch, int nmodes, const gofo_gmm *gmm,
                      const float *mean, float Tb, float TB, float tau)
{
    float tWeight = 0.f;
    for (int mode = 0; mode < nmodes; mode++, mean += nch) {
        gofo_gmm g = gmm[mode];
        float numerator = 0.f, denominator = 0.f;
        for (int c = 0; c < nch; c++) {
            numerator += data[c] * mean[c];
            denominator += mean[c] * mean[c];
        }
        if (denominator == 0)
            return 0;
        if (numerator <= denominator && numerator >= tau * denominator) {
            float a = numerator / denominator;
            float dist2a = 0.f;
            for (int c = 0; c < nch; c++) {
                float dD = a * mean[c] - data[c];
                dist2a += dD * dD;
            }
            if (dist2a < Tb * g.variance * a * a)
                return 1;
        }
        tWeight += g.weight;
        if (tWeight > TB)
            return 0;
    }
    return 0;
}

/* One pixel of MOG2Invoker::operator() -- statement order kept as in OpenCV. */
static inline uint8_t mog2_pixel(const gofo_mog2_params *P, int nch, const float *data,
                                 gofo_gmm *gmm, float *mean, uint8_t *modes_used,
                                 float alphaT, float alpha1, float prune)
{
    const int nmixtures = P->nmixtures;
    const float Tb = P->var_threshold, TB = P->background_ratio, Tg = P->var_threshold_gen;
    float dData[GOFO_MAX_CH];
    int background = 0;
    int fitsPDF = 0;
    int nmodes = *modes_used;
    float totalWeight = 0.f;
    float *mean_m = mean;

    /* NB: the bound is the live nmodes, which shrinks when a mode is pruned. */
    for (int mode = 0; mode < nmodes; mode++, mean_m += nch) {
        float weight = alpha1 * gmm[mode].weight + prune;
        int swap_count = 0;
        if (!fitsPDF) {
            float var = gmm[mode].variance;
            float dist2;
            if (nch == 3) {
                dData[0] = mean_m[0] - data[0];
                dData[1] = mean_m[1] - data[1];
                dData[2] = mean_m[2] - data[2];
                dist2 = dData[0] * dData[0] + dData[1] * dData[1] + dData[2] * dData[2];
            } else {
                dist2 = 0.f;
                for (int c = 0; c < nch; c++) {
                    dData[c] = mean_m[c] - data[c];
                    dist2 += dData[c] * dData[c];
                }
            }
            if (totalWeight < TB && dist2 < Tb * var)
                background = 1;
            if (dist2 < Tg * var) {
                fitsPDF = 1;
                weight += alphaT;
                float k = alphaT / weight;
                for (int c = 0; c < nch; c++)
                    mean_m[c] -= k * dData[c];
                float varnew = var + k * (dist2 - var);
                varnew = varnew > P->var_min ? varnew : P->var_min; /* MAX(varnew, varMin) */
                varnew = varnew < P->var_max ? varnew : P->var_max; /* MIN(varnew, varMax) */
                gmm[mode].variance = varnew;
                for (int i = mode; i > 0; i--) {
                    if (weight < gmm[i - 1].weight)
                        break;
                    swap_count++;
                    gofo_gmm t = gmm[i];
                    gmm[i] = gmm[i - 1];
                    gmm[i - 1] = t;
                    for (int c = 0; c < nch; c++) {
                        float tm = mean[i * nch + c];
                        mean[i * nch + c] = mean[(i - 1) * nch + c];
                        mean[(i - 1) * nch + c] = tm;
                    }
                }
            }
        }
        if (weight < -prune) {
            weight = 0.f;
            nmodes--;
        }
        gmm[mode - swap_count].weight = weight;
        totalWeight += weight;
    }

    float invWeight = 0.f;
    if (fabsf(totalWeight) > FLT_EPSILON)
        invWeight = 1.f / totalWeight;
    for (int mode = 0; mode < nmodes; mode++)
        gmm[mode].weight *= invWeight;

    if (!fitsPDF && alphaT > 0.f) {
        int mode = nmodes == nmixtures ? nmixtures - 1 : nmodes++;
        if (nmodes == 1)
            gmm[mode].weight = 1.f;
        else {
            gmm[mode].weight = alphaT;
            for (int i = 0; i < nmodes - 1; i++)
                gmm[i].weight *= alpha1;
        }
        for (int c = 0; c < nch; c++)
            mean[mode * nch + c] = data[c];
        gmm[mode].variance = P->var_init;
        for (int i = nmodes - 1; i > 0; i--) {
            if (alphaT < gmm[i - 1].weight)
                break;
            gofo_gmm t = gmm[i];
            gmm[i] = gmm[i - 1];
            gmm[i - 1] = t;
            for (int c = 0; c < nch; c++) {
                float tm = mean[i * nch + c];
                mean[i * nch + c] = mean[(i - 1) * nch + c];
                mean[(i - 1) * nch + c] = tm;
            }
        }
    }

    *modes_used = (uint8_t)nmodes;
    if (background)
        return 0;
    if (P->detect_shadows && shadow_gmm(data, nch, nmodes, gmm, mean, Tb, TB, P->tau))
        return (uint8_t)P->shadow_value;
    return 255;
}

/* learning rate of BackgroundSubtractorMOG2Impl::apply with learningRate=-1:
 * ++nframes; alpha = 1/min(2*nframes, history) in double, cast to float;
 * prune = float(-alpha_double * fCT). */
static void mog2_rates(gofo_mog2 *m, float *alphaT, float *alpha1, float *prune)
{
    ++m->nframes;
    int d = 2 * m->nframes < m->p.history ? 2 * m->nframes : m->p.history;
    double lr = 1.0 / (double)d;
    *alphaT = (float)lr;
    *alpha1 = 1.f - *alphaT;
    *prune = (float)(-lr * m->p.ct);
}

static void mog2_rows(gofo_mog2 *m, const uint8_t *frame, uint8_t *mask, int y0, int y1,
                      float alphaT, float alpha1, float prune)
{
    int K = m->p.nmixtures, C = m->c, w = m->w;
    float data[GOFO_MAX_CH];
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            for (int ch = 0; ch < C; ch++)
                data[ch] = (float)frame[i * C + ch];
            mask[i] = mog2_pixel(&m->p, C, data, m->gmm + i * K, m->mean + i * K * C,
                                 m->modes + i, alphaT, alpha1, prune);
        }
    }
}

/* frame: (h,w,c) uint8 interleaved; mask: (h,w) uint8 in {0, shadow_value, 255} */
void gofo_mog2_apply(gofo_mog2 *m, const uint8_t *frame, uint8_t *mask)
{
    float alphaT, alpha1, prune;
    mog2_rates(m, &alphaT, &alpha1, &prune);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < m->h; y++)
        mog2_rows(m, frame, mask, y, y + 1, alphaT, alpha1, prune);
}

/* ------------------------------------------------------------------------- */
/* FrameDifferencing  [REF segmentation.py:40-70]                             */
/* ------------------------------------------------------------------------- */

/*
 * has_prev == 0: F = zeros_like(X)                          (segmentation.py:55-56)
 * else: diff = cv.absdiff(X, prev); F = 255 where diff > floor(thr)
 *       (cv.threshold on uint8 floors a float threshold [OPENCV-RECALL])
 *                                                            (segmentation.py:58-64)
 * c > 1: np.mean(F, axis=-1, dtype=np.uint8): the sum is accumulated in uint8
 *        (wraps mod 256), then true-divided by c and truncated (segmentation.py:67-68)
 * The caller keeps prev = X (segmentation.py:66).
 */
void gofo_fd_apply(const uint8_t *frame, const uint8_t *prev, int has_prev, double thr,
                   int h, int w, int c, uint8_t *mask)
{
    size_t n = (size_t)h * w;
    int ithr = (int)floor(thr);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint8_t acc = 0;
        for (int ch = 0; ch < c; ch++) {
            uint8_t f = 0;
            if (has_prev) {
                int a = frame[i * c + ch], b = prev[i * c + ch];
                int d = a > b ? a - b : b - a;
                f = d > ithr ? 255 : 0;
            }
            acc = (uint8_t)(acc + f);
        }
        mask[i] = c > 1 ? (uint8_t)((double)acc / (double)c) : acc;
    }
}

/* ------------------------------------------------------------------------- */
/* erode / dilate   [OPENCV-RECALL]                            */
/* ------------------------------------------------------------------------- */

/*
 * op 0 = erode (min), 1 = dilate (max) over {(dy,dx): K[dy][dx] != 0},
 * anchor (kh/2, kw/2), one iteration, BORDER_CONSTANT with
 * morphologyDefaultBorderValue: out-of-image samples never win (255 for
 * erode, 0 for dilate).  An all-zero K yields that identity everywhere.
 */
void gofo_morph(int op, const uint8_t *src, uint8_t *dst, int h, int w,
                const uint8_t *k, int kh, int kw)
{
    int ay = kh / 2, ax = kw / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int v = op ? 0 : 255;
            for (int dy = 0; dy < kh; dy++) {
                int yy = y + dy - ay;
                if (yy < 0 || yy >= h)
                    continue;
                for (int dx = 0; dx < kw; dx++) {
                    if (!k[dy * kw + dx])
                        continue;
                    int xx = x + dx - ax;
                    if (xx < 0 || xx >= w)
                        continue;
                    int s = src[(size_t)yy * w + xx];
                    if (op ? s > v : s < v)
                        v = s;
                }
            }
            dst[(size_t)y * w + x] = (uint8_t)v;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* BackgroundSubtractorKNN  [REF pipeline.py:233-234: cv.createBackgroundSubtractorKNN()] */
/* ------------------------------------------------------------------------- */
/* cv2's K-nearest-neighbours subtractor (Zivkovic & van der Heijden 2006) with all defaults
 * [OPENCV-RECALL bgfg_KNN.cpp]: per pixel three sets (short / mid / long term) of nN = 7 samples,
 * a sample = the pixel's channel bytes + one flag byte ("was background when stored"); a pixel is
 * background when at least kNN = 2 FLAGGED samples lie within dist2Threshold = 400 of it, shadow
 * (127) when that many flagged samples pass the chromaticity test (tau 0.5), else foreground.
 * The sets are refreshed at staggered times: a set's counter runs 0 .. nUpdate - 1 (nUpdate from
 * the learning rate 1 / min(2 n, history), history 500), a pixel is refreshed in the frame where
 * the counter meets its own slot, and at the end of a cycle every pixel DRAWS the slot of the next.
 *
 * PARITY WITH cv2: UNPINNABLE BY CONSTRUCTION.  cv2 draws those slots with C rand() inside a loop
 * it runs row-parallel, so its own output depends on thread scheduling: two cv2 runs differ, and
 * no restatement can match "it" bit for bit.  This restatement keeps cv2's model, thresholds,
 * update structure and statement order and replaces the draw by a counter-based generator keyed
 * on (pixel, frame, set) -- gofo_knn_draw below -- so that the C oracle and the HIP kernel are
 * bit-identical to EACH OTHER (masks and every model byte); against cv2 only the statistics match.
 */
typedef struct gofo_knn_params {
    int history;            /* 500 */
    int nsamples;           /* nN: samples per set, 7 */
    float dist2_threshold;  /* 400 */
    int knn;                /* 2 = max(1, cvRound(0.1 * nN * 3 + 0.40)) */
    float tau;              /* 0.5 */
    int detect_shadows;     /* 1 */
    int shadow_value;       /* 127 */
} gofo_knn_params;

void gofo_knn_default_params(gofo_knn_params *p)
{
    p->history = 500;
    p->nsamples = 7;
    p->dist2_threshold = 20.0f * 20.0f;
    p->knn = 2;
    p->tau = 0.5f;
    p->detect_shadows = 1;
    p->shadow_value = 127;
}

#define GOFO_KNN_MAX_SAMPLES 16

typedef struct gofo_knn {
    gofo_knn_params p;
    int h, w, c;
    int nframes;
    int short_counter, mid_counter, long_counter;
    uint8_t *model;       /* [h*w][3*nN][c+1]: short set, mid set, long set */
    uint8_t *next;        /* [h*w][3]: slot of the next short / mid / long refresh */
    uint8_t *index;       /* [h*w][3]: write position in the short / mid / long set */
} gofo_knn;

gofo_knn *gofo_knn_create(int h, int w, int c, const gofo_knn_params *p)
{
    if (h <= 0 || w <= 0 || c <= 0 || c > GOFO_MAX_CH)
        return NULL;
    gofo_knn *m = (gofo_knn *)calloc(1, sizeof(*m));
    if (p)
        m->p = *p;
    else
        gofo_knn_default_params(&m->p);
    if (m->p.nsamples < 1 || m->p.nsamples > GOFO_KNN_MAX_SAMPLES || m->p.knn < 1 || m->p.history < 1) {
        free(m);
        return NULL;
    }
    m->h = h;
    m->w = w;
    m->c = c;
    size_t n = (size_t)h * w;
    m->model = (uint8_t *)calloc(n * 3 * m->p.nsamples * (c + 1), 1);   /* cv2: all zero, flags clear */
    m->next = (uint8_t *)calloc(n * 3, 1);
    m->index = (uint8_t *)calloc(n * 3, 1);
    return m;
}

void gofo_knn_destroy(gofo_knn *m)
{
    if (!m)
        return;
    free(m->model);
    free(m->next);
    free(m->index);
    free(m);
}

int gofo_knn_nframes(const gofo_knn *m) { return m->nframes; }

/* The draw that replaces rand(): 31 bits (glibc's range) from (pixel, frame, set) -- the
 * "lowbias32" integer hash.  pixel = y * w + x, frame = the 1-based frame count, set 0 / 1 / 2. */
uint32_t gofo_knn_draw(uint32_t pixel, uint32_t frame, uint32_t set)
{
    uint32_t x = pixel * 0x9E3779B1u ^ (frame * 0x85EBCA77u + set * 0xC2B2AE3Du);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x >> 1;
}

/* nShortUpdate, nMidUpdate, nLongUpdate of this frame [OPENCV-RECALL apply()] */
static void knn_rates(const gofo_knn *m, int nframes, int *ns, int *nm, int *nl)
{
    double lr = 1.0 / (double)(2 * nframes < m->p.history ? 2 * nframes : m->p.history);
    int nN = m->p.nsamples;
    int Kshort = (int)(log(0.7) / log(1 - lr)) + 1;
    int Kmid = (int)(log(0.4) / log(1 - lr)) - Kshort + 1;
    int Klong = (int)(log(0.1) / log(1 - lr)) - Kshort - Kmid + 1;
    *ns = Kshort / nN + 1;
    *nm = Kmid / nN + 1;
    *nl = Klong / nN + 1;
}

/* _cvCheckPixelBackgroundNP: 1 background, 2 shadow, 0 foreground; *include: store the pixel as
 * a background sample */
static int knn_check(const gofo_knn_params *P, const uint8_t *data, int nch, const uint8_t *model,
                     uint8_t *include)
{
    int Pbf = 0, Pb = 0, nN = P->nsamples, ndata = nch + 1;
    float dData[GOFO_MAX_CH];
    *include = 0;
    for (int n = 0; n < nN * 3; n++) {
        const uint8_t *mean_m = model + n * ndata;
        float dist2;
        if (nch == 3) {
            dData[0] = (float)mean_m[0] - data[0];
            dData[1] = (float)mean_m[1] - data[1];
            dData[2] = (float)mean_m[2] - data[2];
            dist2 = dData[0] * dData[0] + dData[1] * dData[1] + dData[2] * dData[2];
        } else {
            dist2 = 0.f;
            for (int c = 0; c < nch; c++) {
                dData[c] = (float)mean_m[c] - data[c];
                dist2 += dData[c] * dData[c];
            }
        }
        if (dist2 < P->dist2_threshold) {
            Pbf++;
            if (mean_m[nch]) {
                Pb++;
                if (Pb >= P->knn) {
                    *include = 1;
                    return 1;
                }
            }
        }
    }
    if (Pbf >= P->knn)
        *include = 1;
    int Ps = 0;
    if (P->detect_shadows) {
        for (int n = 0; n < nN * 3; n++) {
            const uint8_t *mean_m = model + n * ndata;
            if (mean_m[nch]) {
                float numerator = 0.0f, denominator = 0.0f;
                for (int c = 0; c < nch; c++) {
                    numerator += (float)data[c] * mean_m[c];
                    denominator += (float)mean_m[c] * mean_m[c];
                }
                if (denominator == 0)
                    return 0;
                if (numerator <= denominator && numerator >= P->tau * denominator) {
                    float a = numerator / denominator;
                    float dist2a = 0.0f;
                    for (int c = 0; c < nch; c++) {
                        float dD = a * mean_m[c] - data[c];
                        dist2a += dD * dD;
                    }
                    if (dist2a < P->dist2_threshold * a * a) {
                        Ps++;
                        if (Ps >= P->knn)
                            return 2;
                    }
                }
            }
        }
    }
    return 0;
}

/* frame (h,w,c) uint8 -> mask (h,w) in {0, shadow_value, 255} */
void gofo_knn_apply(gofo_knn *m, const uint8_t *frame, uint8_t *mask)
{
    int h = m->h, w = m->w, nch = m->c, nN = m->p.nsamples, ndata = nch + 1;
    m->nframes++;
    int nShort, nMid, nLong;
    knn_rates(m, m->nframes, &nShort, &nMid, &nLong);
    int LongCounter = m->long_counter, MidCounter = m->mid_counter, ShortCounter = m->short_counter;
    m->short_counter++;
    m->mid_counter++;
    m->long_counter++;
    if (m->short_counter >= nShort)
        m->short_counter = 0;
    if (m->mid_counter >= nMid)
        m->mid_counter = 0;
    if (m->long_counter >= nLong)
        m->long_counter = 0;
    const uint32_t frame_no = (uint32_t)m->nframes;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            const uint8_t *data = frame + i * nch;
            uint8_t *model = m->model + i * 3 * nN * ndata;
            uint8_t *next = m->next + i * 3, *index = m->index + i * 3;
            uint8_t include = 0;
            int result = knn_check(&m->p, data, nch, model, &include);
            /* _cvUpdatePixelBackgroundNP: offsets first, then long <- mid, mid <- short, short <- pixel */
            size_t offsetLong = (size_t)ndata * (index[2] + nN * 2);
            size_t offsetMid = (size_t)ndata * (index[1] + nN * 1);
            size_t offsetShort = (size_t)ndata * (index[0]);
            if (next[2] == LongCounter) {
                memcpy(model + offsetLong, model + offsetMid, ndata);
                index[2] = (index[2] >= nN - 1) ? 0 : (uint8_t)(index[2] + 1);
            }
            if (LongCounter == nLong - 1)
                next[2] = (uint8_t)(gofo_knn_draw((uint32_t)i, frame_no, 2) % (uint32_t)nLong);
            if (next[1] == MidCounter) {
                memcpy(model + offsetMid, model + offsetShort, ndata);
                index[1] = (index[1] >= nN - 1) ? 0 : (uint8_t)(index[1] + 1);
            }
            if (MidCounter == nMid - 1)
                next[1] = (uint8_t)(gofo_knn_draw((uint32_t)i, frame_no, 1) % (uint32_t)nMid);
            if (next[0] == ShortCounter) {
                memcpy(model + offsetShort, data, nch);
                model[offsetShort + nch] = include;
                index[0] = (index[0] >= nN - 1) ? 0 : (uint8_t)(index[0] + 1);
            }
            if (ShortCounter == nShort - 1)
                next[0] = (uint8_t)(gofo_knn_draw((uint32_t)i, frame_no, 0) % (uint32_t)nShort);
            mask[i] = result == 0 ? 255 : (result == 1 ? 0 : (uint8_t)m->p.shadow_value);
        }
    }
}

/* The model as planes, the layout the HIP side exports: samples[3*nN][c+1][n] (set-major: short
 * 0..nN-1, mid, long; last byte of a sample = its flag), state[6][n] = next short / mid / long,
 * index short / mid / long. */
void gofo_knn_export(const gofo_knn *m, uint8_t *samples, uint8_t *state)
{
    size_t n = (size_t)m->h * m->w;
    int nN = m->p.nsamples, ndata = m->c + 1;
    for (size_t i = 0; i < n; i++) {
        for (int s = 0; s < 3 * nN; s++)
            for (int b = 0; b < ndata; b++)
                samples[((size_t)s * ndata + b) * n + i] = m->model[(i * 3 * nN + s) * ndata + b];
        for (int k = 0; k < 3; k++) {
            state[(size_t)k * n + i] = m->next[i * 3 + k];
            state[(size_t)(3 + k) * n + i] = m->index[i * 3 + k];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Whole path, steps 1-3 of GOFPID.detect  [REF pipeline.py:461-476]          */
/* Used for end-to-end parity and as the timed CPU baseline ("port").         */
/* ------------------------------------------------------------------------- */

typedef struct gofo_pipeline {
    int h, w, c;
    int blur_k;          /* 0 = no blur, else odd ksize (square); < 0: box of -blur_k */
    double blur_sigma;
    int detector;        /* 0 = MOG2, 1 = FD, 2 = KNN                         */
    double fd_thr;
    int n_morph;
    int morph_op[8];
    int morph_k[8];      /* rect size                                         */
    gofo_mog2 *mog2;
    gofo_knn *knn;
    uint8_t *blurred, *prev, *m0, *m1, *tmp;
    int has_prev;
    int threads;
} gofo_pipeline;

gofo_pipeline *gofo_pipeline_create(int h, int w, int c, int blur_k, double blur_sigma,
                                    int detector, double fd_thr, int n_morph,
                                    const int *morph_op, const int *morph_k, int threads)
{
    if (n_morph > 8)
        return NULL;
    gofo_pipeline *p = (gofo_pipeline *)calloc(1, sizeof(*p));
    p->h = h;
    p->w = w;
    p->c = c;
    p->blur_k = blur_k;
    p->blur_sigma = blur_sigma;
    p->detector = detector;
    p->fd_thr = fd_thr;
    p->n_morph = n_morph;
    for (int i = 0; i < n_morph; i++) {
        p->morph_op[i] = morph_op[i];
        p->morph_k[i] = morph_k[i];
    }
    size_t n = (size_t)h * w;
    p->blurred = (uint8_t *)malloc(n * c);
    p->prev = (uint8_t *)malloc(n * c);
    p->m0 = (uint8_t *)malloc(n);
    p->m1 = (uint8_t *)malloc(n);
    p->tmp = (uint8_t *)malloc(n);
    p->mog2 = detector == 0 ? gofo_mog2_create(h, w, c, NULL) : NULL;
    p->knn = detector == 2 ? gofo_knn_create(h, w, c, NULL) : NULL;
    p->threads = threads;
    return p;
}

void gofo_pipeline_destroy(gofo_pipeline *p)
{
    if (!p)
        return;
    gofo_mog2_destroy(p->mog2);
    gofo_knn_destroy(p->knn);
    free(p->blurred);
    free(p->prev);
    free(p->m0);
    free(p->m1);
    free(p->tmp);
    free(p);
}

int gofo_pipeline_threads(const gofo_pipeline *p)
{
#ifdef _OPENMP
    return p->threads > 0 ? p->threads : omp_get_max_threads();
#else
    (void)p;
    return 1;
#endif
}

/* frame (h,w,c) -> mask (h,w).  Row bands over `threads` OpenMP threads, the
 * same decomposition OpenCV's parallel_for_ uses inside each call. */
void gofo_pipeline_apply(gofo_pipeline *p, const uint8_t *frame, uint8_t *mask)
{
    int h = p->h, w = p->w, c = p->c;
    int nt = gofo_pipeline_threads(p);
    const uint8_t *x = frame;
    uint16_t kx[256];

    if (p->blur_k < 0) {
        gofo_box_blur(frame, p->blurred, h, w, c, -p->blur_k, -p->blur_k);
        x = p->blurred;
    } else if (p->blur_k) {
        gofo_gauss_kernel_q8(p->blur_k, p->blur_sigma, kx);
        if (p->blur_k == 3) {
#pragma omp parallel for schedule(static) num_threads(nt)
            for (int b = 0; b < nt; b++) {
                int y0 = (int)((long)h * b / nt), y1 = (int)((long)h * (b + 1) / nt);
                if (y1 > y0)
                    blur3_rows(frame, p->blurred, h, w, c, kx, kx, y0, y1);
            }
        } else {
            gofo_sepfilter_q8(frame, p->blurred, h, w, c, kx, p->blur_k, kx, p->blur_k);
        }
        x = p->blurred;
    }

    uint8_t *cur = p->n_morph ? p->m0 : mask;
    if (p->detector == 0) {
        float alphaT, alpha1, prune;
        mog2_rates(p->mog2, &alphaT, &alpha1, &prune);
#pragma omp parallel for schedule(static) num_threads(nt)
        for (int b = 0; b < nt; b++) {
            int y0 = (int)((long)h * b / nt), y1 = (int)((long)h * (b + 1) / nt);
            mog2_rows(p->mog2, x, cur, y0, y1, alphaT, alpha1, prune);
        }
    } else if (p->detector == 2) {
        gofo_knn_apply(p->knn, x, cur);
    } else {
        gofo_fd_apply(x, p->prev, p->has_prev, p->fd_thr, h, w, c, cur);
        memcpy(p->prev, x, (size_t)h * w * c);
        p->has_prev = 1;
    }

    for (int i = 0; i < p->n_morph; i++) {
        uint8_t *out = (i == p->n_morph - 1) ? mask : (cur == p->m0 ? p->m1 : p->m0);
        int k = p->morph_k[i], op = p->morph_op[i];
        /* the row pass must be complete for all bands before the column pass
         * reads neighbours' rows: two parallel regions. */
#pragma omp parallel for schedule(static) num_threads(nt)
        for (int b = 0; b < nt; b++) {
            int y0 = (int)((long)h * b / nt), y1 = (int)((long)h * (b + 1) / nt);
            int ax = k / 2;
            for (int y = y0; y < y1; y++) {
                const uint8_t *s = cur + (size_t)y * w;
                uint8_t *t = p->tmp + (size_t)y * w;
                for (int xx = 0; xx < w; xx++) {
                    int lo = xx - ax < 0 ? 0 : xx - ax;
                    int hi = xx + (k - 1 - ax) >= w ? w - 1 : xx + (k - 1 - ax);
                    int v = op ? 0 : 255;
                    for (int q = lo; q <= hi; q++) {
                        int sv = s[q];
                        v = op ? (sv > v ? sv : v) : (sv < v ? sv : v);
                    }
                    t[xx] = (uint8_t)v;
                }
            }
        }
#pragma omp parallel for schedule(static) num_threads(nt)
        for (int b = 0; b < nt; b++) {
            int y0 = (int)((long)h * b / nt), y1 = (int)((long)h * (b + 1) / nt);
            int ay = k / 2;
            for (int y = y0; y < y1; y++) {
                int lo = y - ay < 0 ? 0 : y - ay;
                int hi = y + (k - 1 - ay) >= h ? h - 1 : y + (k - 1 - ay);
                uint8_t *d = out + (size_t)y * w;
                memcpy(d, p->tmp + (size_t)lo * w, (size_t)w);
                for (int yy = lo + 1; yy <= hi; yy++) {
                    const uint8_t *t = p->tmp + (size_t)yy * w;
                    if (op)
                        for (int xx = 0; xx < w; xx++)
                            d[xx] = t[xx] > d[xx] ? t[xx] : d[xx];
                    else
                        for (int xx = 0; xx < w; xx++)
                            d[xx] = t[xx] < d[xx] ? t[xx] : d[xx];
                }
            }
        }
        cur = out;
    }
}

/* mean number of Gaussian modes in use per pixel (the m-bar of the roofline
 * byte count). */
double gofo_pipeline_mean_modes(const gofo_pipeline *p)
{
    if (!p->mog2)
        return 0.0;
    size_t n = (size_t)p->h * p->w;
    double s = 0;
    for (size_t i = 0; i < n; i++)
        s += p->mog2->modes[i];
    return s / (double)n;
}

gofo_mog2 *gofo_pipeline_mog2(gofo_pipeline *p) { return p->mog2; }
gofo_knn *gofo_pipeline_knn(gofo_pipeline *p) { return p->knn; }
