"""The slice of the cv2 namespace that GOFPID's foreground path calls, on MI355X.

The reference does `import cv2 as cv` and calls, on steps 1-3 of
GOFPID.detect (pygofpid/pipeline.py:461-476):

    cv.GaussianBlur(X, ksize, borderType)           pipeline.py:463-467
    cv.createBackgroundSubtractorMOG2().apply(X)    pipeline.py:232, 470
    cv.createBackgroundSubtractorKNN().apply(X)     pipeline.py:233-234, 470
    cv.erode(mask, kernel) / cv.dilate(mask, kernel) pipeline.py:475
    cv.getStructuringElement(cv.MORPH_RECT, (5, 5))  pipeline.py:180, 184, 247

`from pygofpid_amd import cv` gives the same names with the same positional
signatures, each one a ctypes call into libgofpid_hip.so (HIP kernels, no
CPU fallback).  Only uint8 input is supported, like every call site above.
"""
import ctypes as C

import numpy as np

from . import _lib

# constants with cv2's numeric values
BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_REFLECT = 2
BORDER_REFLECT_101 = 4
BORDER_REFLECT101 = 4
BORDER_DEFAULT = 4
MORPH_RECT = 0
MORPH_CROSS = 1
MORPH_ELLIPSE = 2
COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7


def getStructuringElement(shape, ksize, anchor=(-1, -1)):
    """cv.getStructuringElement: (rows=ksize[1], cols=ksize[0]) uint8 of 0/1.

    MORPH_RECT is what the reference uses (pipeline.py:180); CROSS and ELLIPSE
    follow cv2's construction [OPENCV-RECALL] and are host-side glue only.
    """
    kw, kh = int(ksize[0]), int(ksize[1])
    if kw < 1 or kh < 1:
        raise ValueError("ksize must be positive")
    ax = kw // 2 if anchor[0] < 0 else int(anchor[0])
    ay = kh // 2 if anchor[1] < 0 else int(anchor[1])
    if shape == MORPH_RECT or kw == kh == 1:
        return np.ones((kh, kw), np.uint8)
    k = np.zeros((kh, kw), np.uint8)
    if shape == MORPH_CROSS:
        k[ay, :] = 1
        k[:, ax] = 1
        return k
    if shape == MORPH_ELLIPSE:
        r, c = kh // 2, kw // 2
        inv_r2 = 1.0 / (r * r) if r else 0.0
        for i in range(kh):
            dy = i - r
            if abs(dy) <= r:
                dx = int(round(c * np.sqrt((r * r - dy * dy) * inv_r2)))
                k[i, max(c - dx, 0):min(c + dx + 1, kw)] = 1
        return k
    raise ValueError("unknown structuring element shape %r" % (shape,))


def cvtColor(src, code):
    """cv.cvtColor for the two codes the reference's tests use
    (COLOR_BGR2GRAY, COLOR_RGB2GRAY; pipeline.py:458-459): 8-bit fixed-point luma
    of the OpenCV 3.4 line, Y = (1868 B + 9617 G + 4899 R + 2^13) >> 14."""
    if code not in (COLOR_BGR2GRAY, COLOR_RGB2GRAY):
        raise ValueError("only COLOR_BGR2GRAY / COLOR_RGB2GRAY are built")
    x = _lib.as_u8(src, "src")
    x, h, w, c = _lib.hwc(x)
    out = np.empty((h, w), np.uint8)
    ctx = _lib.default_context()
    _lib.check(_lib.load().gof_cvt_gray_host(ctx.handle, _lib.p8(x), _lib.p8(out), h, w, c,
                                             int(code == COLOR_RGB2GRAY)))
    return out


def GaussianBlur(src, ksize, sigmaX, dst=None, sigmaY=0, borderType=BORDER_DEFAULT):
    """cv.GaussianBlur on uint8: 8.8 fixed-point separable filter, REFLECT_101.

    Positional order is cv2's, so the reference's call
    `blur['fun'](X, blur['ksize'], blur['borderType'])` (pipeline.py:463-467)
    binds BORDER_DEFAULT (= 4) to sigmaX exactly as it does with cv2.
    (H, W, 1) input returns (H, W), as cv2 does.
    """
    if borderType not in (BORDER_DEFAULT, BORDER_REFLECT_101):
        raise ValueError("only BORDER_REFLECT_101 (the reference's effective border) is built")
    x = _lib.as_u8(src, "src")
    x, h, w, c = _lib.hwc(x)
    kw, kh = int(ksize[0]), int(ksize[1])
    sx, sy = float(sigmaX), float(sigmaY)
    if kw <= 0 and sx > 0:      # cv2: ksize from sigma, 8-bit uses 3 sigma
        kw = int(round(sx * 3 * 2 + 1)) | 1
    if kh <= 0 and (sy if sy > 0 else sx) > 0:
        kh = int(round((sy if sy > 0 else sx) * 3 * 2 + 1)) | 1
    out = np.empty((h, w) if c == 1 else (h, w, c), np.uint8)
    ctx = _lib.default_context()
    _lib.check(_lib.load().gof_gaussian_blur_host(ctx.handle, _lib.p8(x), _lib.p8(out), h, w, c,
                                                  kw, kh, sx, sy))
    return out


def blur(src, ksize, dst=None, anchor=(-1, -1), borderType=BORDER_DEFAULT):
    """cv.blur: normalised box filter on uint8, REFLECT_101, round-half-even.

    Positional order is cv2's.  The reference calls
    `blur['fun'](X, blur['ksize'], blur['borderType'])` (pipeline.py:463-467) and its
    tests pass cv.blur there (tests/test_pipeline.py:68-75): cv2 binds that third
    argument to `dst`, so the border it names is ignored -- and so it is here.
    """
    if borderType not in (BORDER_DEFAULT, BORDER_REFLECT_101):
        raise ValueError("only BORDER_REFLECT_101 (the reference's effective border) is built")
    if anchor not in ((-1, -1), None):
        raise ValueError("only the centre anchor is built")
    x = _lib.as_u8(src, "src")
    x, h, w, c = _lib.hwc(x)
    out = np.empty((h, w) if c == 1 else (h, w, c), np.uint8)
    ctx = _lib.default_context()
    _lib.check(_lib.load().gof_box_blur_host(ctx.handle, _lib.p8(x), _lib.p8(out), h, w, c,
                                             int(ksize[0]), int(ksize[1])))
    return out


def _morph(op, src, kernel):
    x = _lib.as_u8(src, "src")
    if x.ndim == 3 and x.shape[2] == 1:
        x = np.ascontiguousarray(x[:, :, 0])
    if x.ndim != 2:
        raise ValueError("morphology is built for single-channel masks, got shape %r" % (x.shape,))
    k = np.ascontiguousarray(np.asarray(kernel) != 0, dtype=np.uint8)
    if k.ndim != 2:
        raise ValueError("kernel must be 2-D")
    out = np.empty_like(x)
    ctx = _lib.default_context()
    _lib.check(_lib.load().gof_morph_host(ctx.handle, op, _lib.p8(x), _lib.p8(out), x.shape[0],
                                          x.shape[1], _lib.p8(k), k.shape[0], k.shape[1]))
    return out


def erode(src, kernel):
    """cv.erode(mask, kernel): min over the kernel's non-zero taps, centre anchor,
    one iteration, constant border that never wins (pipeline.py:475)."""
    return _morph(_lib.ERODE, src, kernel)


def dilate(src, kernel):
    """cv.dilate(mask, kernel): max over the kernel's non-zero taps (pipeline.py:475)."""
    return _morph(_lib.DILATE, src, kernel)


class BackgroundSubtractorMOG2:
    """cv.createBackgroundSubtractorMOG2() object: `.apply(X) -> (H, W) uint8` in
    {0, 127, 255}; the model lives in HBM as SoA planes and is sized lazily on
    the first frame (no frame size is known at GOFPID.initialize(),
    pipeline.py:231-232)."""

    def __init__(self, history=500, varThreshold=16, detectShadows=True, device=None):
        p = _lib.Mog2Params()
        _lib.load().gof_mog2_default_params(C.byref(p))
        p.history = int(history)
        p.var_threshold = float(varThreshold)
        p.detect_shadows = int(bool(detectShadows))
        self._params = p
        self._device = device
        self._h = C.c_void_p()
        self._shape = None
        self._ctx = None

    def _ensure(self, h, w, c):
        if self._shape == (h, w, c):
            return
        # cv2 re-initialises the model when the frame size or type changes
        self._free()
        self._ctx = _lib.default_context(self._device)
        _lib.check(_lib.load().gof_mog2_create(self._ctx.handle, 1, h, w, c,
                                               C.byref(self._params), C.byref(self._h)))
        self._shape = (h, w, c)

    def apply(self, image):
        x = _lib.as_u8(image, "image")
        x, h, w, c = _lib.hwc(x)
        self._ensure(h, w, c)
        out = np.empty((h, w), np.uint8)
        _lib.check(_lib.load().gof_mog2_apply_host(self._h, _lib.p8(x), _lib.p8(out)))
        return out

    def export_state(self):
        """SoA planes {weight (K,N), var (K,N), mean (K,C,N), modes (N,)}, nframes."""
        if not self._h:
            raise RuntimeError("no frame seen yet")
        h, w, c = self._shape
        n, K = h * w, self._params.nmixtures
        wt, var = np.empty((K, n), np.float32), np.empty((K, n), np.float32)
        mean, modes = np.empty((K, c, n), np.float32), np.empty(n, np.uint8)
        f32 = C.POINTER(C.c_float)
        _lib.check(_lib.load().gof_mog2_export(self._h, wt.ctypes.data_as(f32),
                                               var.ctypes.data_as(f32), mean.ctypes.data_as(f32),
                                               _lib.p8(modes)))
        return {"weight": wt, "var": var, "mean": mean, "modes": modes,
                "nframes": _lib.load().gof_mog2_nframes(self._h)}

    def import_state(self, state, shape):
        h, w, c = shape
        self._ensure(h, w, c)
        f32 = C.POINTER(C.c_float)
        wt, var, mean, modes = _lib.mog2_state_arrays(state, self._params.nmixtures, c, h * w)
        _lib.check(_lib.load().gof_mog2_import(self._h, wt.ctypes.data_as(f32),
                                               var.ctypes.data_as(f32), mean.ctypes.data_as(f32),
                                               _lib.p8(modes), int(state["nframes"])))

    def _free(self):
        if getattr(self, "_h", None):
            _lib.load().gof_mog2_destroy(self._h)
            self._h = C.c_void_p()
            self._shape = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def createBackgroundSubtractorMOG2(history=500, varThreshold=16, detectShadows=True):
    return BackgroundSubtractorMOG2(history, varThreshold, detectShadows)


class BackgroundSubtractorKNN:
    """cv.createBackgroundSubtractorKNN() object [REF pygofpid/pipeline.py:233-234]: `.apply(X) ->
    (H, W) uint8` in {0, 127, 255}.  cv2's model (three sets of 7 samples per pixel), rule and
    refresh structure with cv2's defaults; the per-pixel refresh slots, which cv2 draws with rand()
    in a row-parallel loop (its own masks differ run to run), come from a hash of (pixel, frame,
    set): deterministic, and bit-identical to this repository's CPU restatement.  Parity with cv2
    itself is unpinnable by construction."""

    def __init__(self, history=500, dist2Threshold=400.0, detectShadows=True, device=None):
        p = _lib.KnnParams()
        _lib.load().gof_knn_default_params(C.byref(p))
        p.history = int(history)
        p.dist2_threshold = float(dist2Threshold)
        p.detect_shadows = int(bool(detectShadows))
        self._params = p
        self._device = device
        self._h = C.c_void_p()
        self._shape = None
        self._ctx = None

    def _ensure(self, h, w, c):
        if self._shape == (h, w, c):
            return
        self._free()     # cv2 re-initialises the model when the frame size or type changes
        self._ctx = _lib.default_context(self._device)
        _lib.check(_lib.load().gof_knn_create(self._ctx.handle, 1, h, w, c, C.byref(self._params),
                                              C.byref(self._h)))
        self._shape = (h, w, c)

    def apply(self, image):
        x = _lib.as_u8(image, "image")
        x, h, w, c = _lib.hwc(x)
        self._ensure(h, w, c)
        out = np.empty((h, w), np.uint8)
        _lib.check(_lib.load().gof_knn_apply_host(self._h, _lib.p8(x), _lib.p8(out)))
        return out

    def export_state(self):
        """{"samples": (3 nsamples, c + 1, N) uint8 -- sets short, mid, long; last byte of a sample
        = its flag --, "state": (6, N) uint8 = next short / mid / long slot, index short / mid / long}"""
        if not self._h:
            raise RuntimeError("no frame seen yet")
        h, w, c = self._shape
        return knn_export(self._h, h * w, c, self._params.nsamples)

    def _free(self):
        if getattr(self, "_h", None):
            _lib.load().gof_knn_destroy(self._h)
            self._h = C.c_void_p()
            self._shape = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def knn_export(handle, n, c, nsamples=7):
    samples = np.empty((3 * nsamples, c + 1, n), np.uint8)
    state = np.empty((6, n), np.uint8)
    _lib.check(_lib.load().gof_knn_export(handle, _lib.p8(samples), _lib.p8(state)))
    return {"samples": samples, "state": state, "nframes": _lib.load().gof_knn_nframes(handle)}


def createBackgroundSubtractorKNN(history=500, dist2Threshold=400.0, detectShadows=True):
    return BackgroundSubtractorKNN(history, dist2Threshold, detectShadows)
