"""ctypes front-end of the C oracle (oracle/gof_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gof_oracle.c.  Imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  PARITY
UNPINNED: the arithmetic restated here lives in opencv-python, which is absent
(SURVEY.md 8c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgof_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "gof_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO




def cpu_budget():
    """CPUs this process may actually use: min(affinity mask, cgroup CPU quota).
    A GPU box exposes all host cores but grants a share; OpenMP would otherwise
    start one spinning thread per visible core."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


# libgomp reads these once, when the library is loaded
os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

_lib = None

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_f32p = C.POINTER(C.c_float)


class Mog2Params(C.Structure):
    _fields_ = [
        ("history", C.c_int),
        ("nmixtures", C.c_int),
        ("var_threshold", C.c_float),
        ("background_ratio", C.c_float),
        ("var_threshold_gen", C.c_float),
        ("var_init", C.c_float),
        ("var_min", C.c_float),
        ("var_max", C.c_float),
        ("ct", C.c_float),
        ("tau", C.c_float),
        ("detect_shadows", C.c_int),
        ("shadow_value", C.c_int),
    ]


class KnnParams(C.Structure):
    _fields_ = [("history", C.c_int), ("nsamples", C.c_int), ("dist2_threshold", C.c_float),
                ("knn", C.c_int), ("tau", C.c_float), ("detect_shadows", C.c_int),
                ("shadow_value", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.gofo_reflect101.restype = C.c_int
        L.gofo_gauss_kernel_q8.argtypes = [C.c_int, C.c_double, _u16p]
        L.gofo_gauss_kernel_q8.restype = C.c_int
        L.gofo_sepfilter_q8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int,
                                        _u16p, C.c_int, _u16p, C.c_int]
        L.gofo_sepfilter_q8.restype = None
        L.gofo_gaussian_blur.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_double, C.c_double]
        L.gofo_gaussian_blur.restype = C.c_int
        L.gofo_cvt_gray.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.gofo_cvt_gray.restype = None
        L.gofo_box_blur.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.gofo_box_blur.restype = None
        L.gofo_mog2_default_params.argtypes = [C.POINTER(Mog2Params)]
        L.gofo_mog2_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Mog2Params)]
        L.gofo_mog2_create.restype = C.c_void_p
        L.gofo_mog2_destroy.argtypes = [C.c_void_p]
        L.gofo_mog2_nframes.argtypes = [C.c_void_p]
        L.gofo_mog2_nframes.restype = C.c_int
        L.gofo_mog2_export.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _u8p]
        L.gofo_mog2_import.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _u8p, C.c_int]
        L.gofo_mog2_import.restype = None
        L.gofo_mog2_apply.argtypes = [C.c_void_p, _u8p, _u8p]
        L.gofo_fd_apply.argtypes = [_u8p, _u8p, C.c_int, C.c_double, C.c_int, C.c_int,
                                    C.c_int, _u8p]
        L.gofo_morph.argtypes = [C.c_int, _u8p, _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.gofo_pipeline_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                           C.c_int, C.c_double, C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        L.gofo_pipeline_create.restype = C.c_void_p
        L.gofo_pipeline_destroy.argtypes = [C.c_void_p]
        L.gofo_pipeline_threads.argtypes = [C.c_void_p]
        L.gofo_pipeline_threads.restype = C.c_int
        L.gofo_pipeline_apply.argtypes = [C.c_void_p, _u8p, _u8p]
        L.gofo_pipeline_mean_modes.argtypes = [C.c_void_p]
        L.gofo_pipeline_mean_modes.restype = C.c_double
        L.gofo_pipeline_mog2.argtypes = [C.c_void_p]
        L.gofo_pipeline_mog2.restype = C.c_void_p
        L.gofo_pipeline_knn.argtypes = [C.c_void_p]
        L.gofo_pipeline_knn.restype = C.c_void_p
        L.gofo_knn_default_params.argtypes = [C.POINTER(KnnParams)]
        L.gofo_knn_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(KnnParams)]
        L.gofo_knn_create.restype = C.c_void_p
        L.gofo_knn_destroy.argtypes = [C.c_void_p]
        L.gofo_knn_nframes.argtypes = [C.c_void_p]
        L.gofo_knn_nframes.restype = C.c_int
        L.gofo_knn_apply.argtypes = [C.c_void_p, _u8p, _u8p]
        L.gofo_knn_export.argtypes = [C.c_void_p, _u8p, _u8p]
        L.gofo_knn_draw.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.gofo_knn_draw.restype = C.c_uint32
        _lib = L
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _as_hwc(x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    if x.ndim == 2:
        return x, x.shape[0], x.shape[1], 1
    return x, x.shape[0], x.shape[1], x.shape[2]


def gauss_kernel_q8(n, sigma):
    taps = np.zeros(n, dtype=np.uint16)
    if lib().gofo_gauss_kernel_q8(n, float(sigma), taps.ctypes.data_as(_u16p)):
        raise ValueError("bad Gaussian ksize %r" % (n,))
    return taps


def gaussian_blur(x, ksize=(3, 3), sigma_x=4.0, sigma_y=0.0):
    """cv.GaussianBlur(x, ksize, sigma_x, sigmaY=sigma_y), REFLECT_101, uint8."""
    x, h, w, c = _as_hwc(x)
    out = np.empty((h, w) if c == 1 else (h, w, c), dtype=np.uint8)
    if lib().gofo_gaussian_blur(_p8(x), _p8(out), h, w, c, int(ksize[0]), int(ksize[1]),
                                float(sigma_x), float(sigma_y)):
        raise ValueError("bad Gaussian ksize %r" % (ksize,))
    return out


def box_blur(x, ksize=(3, 3)):
    """cv.blur(x, ksize): normalised box filter, REFLECT_101, round-half-even."""
    x, h, w, c = _as_hwc(x)
    out = np.empty((h, w) if c == 1 else (h, w, c), dtype=np.uint8)
    lib().gofo_box_blur(_p8(x), _p8(out), h, w, c, int(ksize[0]), int(ksize[1]))
    return out


def cvt_gray(x, rgb_order=False):
    """cv.cvtColor(x, COLOR_BGR2GRAY) (COLOR_RGB2GRAY when rgb_order)."""
    x, h, w, c = _as_hwc(x)
    out = np.empty((h, w), dtype=np.uint8)
    lib().gofo_cvt_gray(_p8(x), _p8(out), h, w, c, int(bool(rgb_order)))
    return out


def sepfilter_q8(x, kx, ky):
    x, h, w, c = _as_hwc(x)
    kx = np.ascontiguousarray(kx, dtype=np.uint16)
    ky = np.ascontiguousarray(ky, dtype=np.uint16)
    out = np.empty((h, w) if c == 1 else (h, w, c), dtype=np.uint8)
    lib().gofo_sepfilter_q8(_p8(x), _p8(out), h, w, c, kx.ctypes.data_as(_u16p), len(kx),
                            ky.ctypes.data_as(_u16p), len(ky))
    return out


def morph(op, mask, kernel):
    """op 'erode' | 'dilate'; mask (h,w) uint8; kernel (kh,kw), non-zero = member."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    kernel = np.ascontiguousarray(kernel, dtype=np.uint8)
    out = np.empty_like(mask)
    lib().gofo_morph(0 if op == "erode" else 1, _p8(mask), _p8(out), mask.shape[0],
                     mask.shape[1], _p8(kernel), kernel.shape[0], kernel.shape[1])
    return out


def erode(mask, kernel):
    return morph("erode", mask, kernel)


def dilate(mask, kernel):
    return morph("dilate", mask, kernel)


class FrameDifferencing:
    """[REF segmentation.py:21-70]"""

    def __init__(self, threshold=5):
        self.threshold = threshold
        self._model = None

    def apply(self, x):
        x, h, w, c = _as_hwc(x)
        out = np.empty((h, w), dtype=np.uint8)
        prev = self._model if self._model is not None else x
        lib().gofo_fd_apply(_p8(x), _p8(prev), int(self._model is not None),
                            float(self.threshold), h, w, c, _p8(out))
        self._model = x.copy()
        return out


class MOG2:
    """cv.createBackgroundSubtractorMOG2() with its defaults [REF pipeline.py:232]."""

    def __init__(self, **overrides):
        self._params = Mog2Params()
        lib().gofo_mog2_default_params(C.byref(self._params))
        for k, v in overrides.items():
            setattr(self._params, k, v)
        self._h = None
        self._shape = None

    def apply(self, x):
        x, h, w, c = _as_hwc(x)
        if self._h is None or self._shape != (h, w, c):
            if self._h is not None:
                lib().gofo_mog2_destroy(self._h)
            self._h = lib().gofo_mog2_create(h, w, c, C.byref(self._params))
            if not self._h:
                raise ValueError("bad MOG2 geometry")
            self._shape = (h, w, c)
        out = np.empty((h, w), dtype=np.uint8)
        lib().gofo_mog2_apply(self._h, _p8(x), _p8(out))
        return out

    def export(self):
        """-> dict of SoA planes: weight (K,N), var (K,N), mean (K,C,N), modes (N,)"""
        h, w, c = self._shape
        n, K = h * w, self._params.nmixtures
        wt = np.empty((K, n), np.float32)
        var = np.empty((K, n), np.float32)
        mean = np.empty((K, c, n), np.float32)
        modes = np.empty(n, np.uint8)
        lib().gofo_mog2_export(self._h, wt.ctypes.data_as(_f32p), var.ctypes.data_as(_f32p),
                               mean.ctypes.data_as(_f32p), _p8(modes))
        return {"weight": wt, "var": var, "mean": mean, "modes": modes}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gofo_mog2_destroy(self._h)
            self._h = None


class KNN:
    """cv.createBackgroundSubtractorKNN() restated (defaults unless overridden); its update draws
    come from gofo_knn_draw (see gof_oracle.c: parity with cv2 itself is unpinnable by construction)."""

    def __init__(self, **overrides):
        self._params = KnnParams()
        lib().gofo_knn_default_params(C.byref(self._params))
        for k, v in overrides.items():
            setattr(self._params, k, v)
        self._h = None
        self._shape = None

    def apply(self, x):
        x, h, w, c = _as_hwc(x)
        if self._h is None:
            self._h = lib().gofo_knn_create(h, w, c, C.byref(self._params))
            if not self._h:
                raise ValueError("bad KNN parameters")
            self._shape = (h, w, c)
        assert (h, w, c) == self._shape
        out = np.empty((h, w), dtype=np.uint8)
        lib().gofo_knn_apply(self._h, _p8(x), _p8(out))
        return out

    def export(self):
        """-> {"samples": (3 nN, c + 1, N) uint8 (last byte of a sample = its flag),
        "state": (6, N) uint8 = next short / mid / long, index short / mid / long}"""
        return _knn_export(self._h, self._shape, self._params.nsamples)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gofo_knn_destroy(self._h)
            self._h = None


def _knn_export(handle, shape, nsamples):
    h, w, c = shape
    n = h * w
    samples = np.empty((3 * nsamples, c + 1, n), np.uint8)
    state = np.empty((6, n), np.uint8)
    lib().gofo_knn_export(handle, _p8(samples), _p8(state))
    return {"samples": samples, "state": state}


class Pipeline:
    """Steps 1-3 of GOFPID.detect [REF pipeline.py:461-476], threaded over rows.

    blur_k: 0 or odd square ksize (negative: box filter of that size); sigma is what the reference passes
    positionally (BORDER_DEFAULT=4 -> sigmaX=4.0).  morph: [('erode', 5), ...]
    with rectangular kernels.
    """

    def __init__(self, h, w, c, blur_k=3, blur_sigma=4.0, detector="MOG2", fd_thr=5.0,
                 morph=(("erode", 5), ("dilate", 5)), threads=0):
        ops = (C.c_int * max(1, len(morph)))(*[0 if m[0] == "erode" else 1 for m in morph])
        ks = (C.c_int * max(1, len(morph)))(*[int(m[1]) for m in morph])
        self.shape = (h, w, c)
        self._h = lib().gofo_pipeline_create(h, w, c, blur_k, float(blur_sigma),
                                             {"MOG2": 0, "FD": 1, "KNN": 2}[detector], float(fd_thr),
                                             len(morph), ops, ks, threads)
        if not self._h:
            raise ValueError("bad pipeline configuration")

    @property
    def threads(self):
        return lib().gofo_pipeline_threads(self._h)

    def apply(self, x):
        x, h, w, c = _as_hwc(x)
        assert (h, w, c) == self.shape
        out = np.empty((h, w), dtype=np.uint8)
        lib().gofo_pipeline_apply(self._h, _p8(x), _p8(out))
        return out

    def mean_modes(self):
        return lib().gofo_pipeline_mean_modes(self._h)

    def export(self):
        """SoA planes of the MOG2 model (as MOG2.export), the KNN model (as KNN.export), or None for FD."""
        k = lib().gofo_pipeline_knn(self._h)
        if k:
            return _knn_export(k, self.shape, 7)
        m = lib().gofo_pipeline_mog2(self._h)
        if not m:
            return None
        h, w, c = self.shape
        n, K = h * w, 5
        wt = np.empty((K, n), np.float32)
        var = np.empty((K, n), np.float32)
        mean = np.empty((K, c, n), np.float32)
        modes = np.empty(n, np.uint8)
        lib().gofo_mog2_export(m, wt.ctypes.data_as(_f32p), var.ctypes.data_as(_f32p),
                               mean.ctypes.data_as(_f32p), _p8(modes))
        return {"weight": wt, "var": var, "mean": mean, "modes": modes}

    def import_state(self, state, nframes):
        """Start from a trained model: SoA planes as export() returns them (weight (K,N), var (K,N),
        mean (K,C,N), modes (N,)) and the number of frames seen (it sets the learning rate)."""
        m = lib().gofo_pipeline_mog2(self._h)
        if not m:
            raise ValueError("not a MOG2 pipeline")
        h, w, c = self.shape
        n, K = h * w, 5
        wt = np.ascontiguousarray(state["weight"], np.float32)
        var = np.ascontiguousarray(state["var"], np.float32)
        mean = np.ascontiguousarray(state["mean"], np.float32)
        modes = np.ascontiguousarray(state["modes"], np.uint8)
        if wt.shape != (K, n) or var.shape != (K, n) or mean.shape != (K, c, n) or modes.shape != (n,):
            raise ValueError("state planes do not match this pipeline's geometry")
        lib().gofo_mog2_import(m, wt.ctypes.data_as(_f32p), var.ctypes.data_as(_f32p),
                               mean.ctypes.data_as(_f32p), _p8(modes), int(nframes))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gofo_pipeline_destroy(self._h)
            self._h = None
