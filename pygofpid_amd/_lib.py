"""ctypes binding of libgofpid_hip.so (C ABI: include/gofpid_hip.h).

There is no CPU fallback anywhere in this package: if the HIP library is not
built, or no gfx950 device is visible, every entry point raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GOFPID_LIB: A/B builds of the kernels (profiles/ ablations); default is the in-tree build
LIB_PATH = os.environ.get("GOFPID_LIB") or os.path.join(_HERE, "csrc", "libgofpid_hip.so")

OK, EINVAL, EHIP, ENOMEM, ENODEV = 0, 1, 2, 3, 4
ERODE, DILATE = 0, 1
DET_MOG2, DET_FD, DET_KNN = 0, 1, 2
BLUR_NONE, BLUR_GAUSS, BLUR_BOX = 0, 1, 2
MAX_MODES, MAX_MORPH, MAX_KSIZE = 5, 8, 31

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_u16p = C.POINTER(C.c_uint16)


class Mog2Params(C.Structure):
    _fields_ = [
        ("history", C.c_int), ("nmixtures", C.c_int),
        ("var_threshold", C.c_float), ("background_ratio", C.c_float),
        ("var_threshold_gen", C.c_float), ("var_init", C.c_float),
        ("var_min", C.c_float), ("var_max", C.c_float),
        ("ct", C.c_float), ("tau", C.c_float),
        ("detect_shadows", C.c_int), ("shadow_value", C.c_int),
    ]


class KnnParams(C.Structure):
    _fields_ = [("history", C.c_int), ("nsamples", C.c_int), ("dist2_threshold", C.c_float),
                ("knn", C.c_int), ("tau", C.c_float), ("detect_shadows", C.c_int),
                ("shadow_value", C.c_int)]


class MorphOp(C.Structure):
    _fields_ = [("op", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("kernel", _u8p)]


class BlobRecord(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("xmin", C.c_int32), ("ymin", C.c_int32),
                ("xmax", C.c_int32), ("ymax", C.c_int32), ("n_points", C.c_int32),
                ("point_offset", C.c_int32), ("a00", C.c_int64), ("a10", C.c_int64),
                ("a01", C.c_int64)]


class PipelineDesc(C.Structure):
    _fields_ = [
        ("streams", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int),
        ("blur", C.c_int), ("blur_kw", C.c_int), ("blur_kh", C.c_int),
        ("blur_sigma_x", C.c_double), ("blur_sigma_y", C.c_double),
        ("detector", C.c_int), ("mog2", Mog2Params), ("fd_threshold", C.c_double),
        ("n_morph", C.c_int), ("morph", MorphOp * MAX_MORPH),
    ]


# name -> (restype, argtypes); also the list tests check against the header
SIGNATURES = {
    "gof_abi_version": (C.c_int, []),
    "gof_strerror": (C.c_char_p, [C.c_int]),
    "gof_last_error": (C.c_char_p, []),
    "gof_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gof_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gof_destroy": (None, [C.c_void_p]),
    "gof_synchronize": (C.c_int, [C.c_void_p]),
    "gof_stream": (C.c_void_p, [C.c_void_p]),
    "gof_cvt_gray_host": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gof_cvt_gray_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p]),
    "gof_gaussian_blur_host": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_double, C.c_double]),
    "gof_gaussian_blur_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_void_p]),
    "gof_box_blur_host": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int]),
    "gof_box_blur_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gof_gauss_kernel_q8": (C.c_int, [C.c_int, C.c_double, _u16p]),
    "gof_mog2_default_params": (None, [C.POINTER(Mog2Params)]),
    "gof_mog2_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(Mog2Params), C.POINTER(C.c_void_p)]),
    "gof_mog2_destroy": (None, [C.c_void_p]),
    "gof_mog2_apply_host": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "gof_mog2_apply_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gof_mog2_nframes": (C.c_int, [C.c_void_p]),
    "gof_mog2_set_nframes": (C.c_int, [C.c_void_p, C.c_int]),
    "gof_mog2_export": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _u8p]),
    "gof_mog2_import": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _u8p, C.c_int]),
    "gof_mog2_device_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_size_t)]),
    "gof_mog2_state_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "gof_fd_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                C.POINTER(C.c_void_p)]),
    "gof_fd_destroy": (None, [C.c_void_p]),
    "gof_fd_apply_host": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "gof_fd_apply_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gof_morph_host": (C.c_int, [C.c_void_p, C.c_int, _u8p, _u8p, C.c_int, C.c_int, _u8p,
                                 C.c_int, C.c_int]),
    "gof_morph_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, _u8p, C.c_int, C.c_int, C.c_void_p]),
    "gof_pipeline_create": (C.c_int, [C.c_void_p, C.POINTER(PipelineDesc),
                                      C.POINTER(C.c_void_p)]),
    "gof_pipeline_destroy": (None, [C.c_void_p]),
    "gof_pipeline_apply_host": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "gof_pipeline_apply_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gof_pipeline_mog2": (C.c_void_p, [C.c_void_p]),
    "gof_pipeline_knn": (C.c_void_p, [C.c_void_p]),
    "gof_knn_default_params": (None, [C.POINTER(KnnParams)]),
    "gof_knn_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(KnnParams),
                                 C.POINTER(C.c_void_p)]),
    "gof_knn_destroy": (None, [C.c_void_p]),
    "gof_knn_apply_host": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "gof_knn_apply_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gof_knn_nframes": (C.c_int, [C.c_void_p]),
    "gof_knn_export": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "gof_pipeline_set_convert": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gof_pipeline_mean_modes": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "gof_pipeline_apply_host_many": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int]),
    "gof_pipeline_apply_dev_many": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p]),
    "gof_pipeline_algorithmic_bytes": (C.c_int, [C.c_void_p, C.c_double, C.c_int,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gof_find_external_contours": (C.c_int, [_u8p, C.c_int, C.c_int,
                                             C.POINTER(C.POINTER(C.c_int32)),
                                             C.POINTER(C.POINTER(C.c_int32)),
                                             C.POINTER(C.c_int)]),
    "gof_free": (None, [C.c_void_p]),
    "gof_blobs_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_void_p)]),
    "gof_blobs_destroy": (None, [C.c_void_p]),
    "gof_blobs_extract_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "gof_blobs_extract_host": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_double]),
    "gof_blobs_counts": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "gof_blobs_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "gof_pipeline_apply_host_blobs": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_void_p, C.c_double]),
    "gof_pipeline_apply_host_blobs_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                                                     C.c_double]),
    "gof_pipeline_fetch_masks": (C.c_int, [C.c_void_p, _u8p, C.c_int]),
    "gof_pipeline_fetch_masks_slot": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int]),
    "gof_pipeline_submit_host_blobs_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                                                      C.c_double, C.c_int]),
    "gof_pipeline_wait_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "gof_pipeline_profile_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gof_pipeline_profile_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "gof_pipeline_profile_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "gof_pipeline_profile_span": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "gof_pipeline_profile_read": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.POINTER(C.c_int)]),
}

_lib = None
_lock = threading.Lock()


def _adopt_torch_hip_runtime():
    """One process, one HIP runtime.  A PyTorch-ROCm wheel carries its own
    libamdhip64; if this library pulled in the system copy first, a later
    `import torch` would meet a second runtime and find no GPU.  So when torch is
    installed (whether or not it has been imported yet) its copy is loaded first
    and libgofpid_hip.so binds to it.  Without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the HIP library and type every entry point; raises if it is missing."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    "pygofpid_amd: %s is not built (run `python -c 'import __graft_entry__ as g; "
                    "g.build()'` or `make -C pygofpid_amd/csrc`); there is no CPU fallback"
                    % LIB_PATH)
            _adopt_torch_hip_runtime()
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc):
    """GOF_EINVAL -> ValueError (what the reference raises for bad shapes /
    unknown methods, tests/test_pipeline.py:31-41,112-118); else RuntimeError."""
    if rc == OK:
        return
    lib = load()
    msg = lib.gof_last_error().decode() or lib.gof_strerror(rc).decode()
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError("%s (gof error %d)" % (msg, rc))


def p8(a):
    return a.ctypes.data_as(_u8p)


def as_u8(x, name="X"):
    x = np.asarray(x)
    if x.dtype != np.uint8:
        raise ValueError("%s must be uint8, got %s" % (name, x.dtype))
    return np.ascontiguousarray(x)


def mog2_state_arrays(state, K, c, n):
    """The four arrays of an exported MOG2 model as C-contiguous arrays of exactly the
    shapes gof_mog2_import reads -- weight (K, n), var (K, n), mean (K, c, n) float32,
    modes (n,) uint8 -- or ValueError (a checkpoint written with another nmixtures or
    geometry must not reach the C side, which trusts the sizes)."""
    out = []
    for key, shape, dt in (("weight", (K, n), np.float32), ("var", (K, n), np.float32),
                           ("mean", (K, c, n), np.float32), ("modes", (n,), np.uint8)):
        a = np.asarray(state[key])
        if a.shape != shape:
            raise ValueError("MOG2 state %r has shape %r, this model needs %r (nmixtures=%d)"
                             % (key, a.shape, shape, K))
        out.append(np.ascontiguousarray(a, dt))
    return out


def hwc(x):
    """(H,W) | (H,W,1) | (H,W,C) -> (array, h, w, c)"""
    if x.ndim == 2:
        return x, x.shape[0], x.shape[1], 1
    if x.ndim == 3:
        return x, x.shape[0], x.shape[1], x.shape[2]
    raise ValueError("expected a (H,W) or (H,W,C) image, got shape %r" % (x.shape,))


class Context:
    """One per (process, device): stream + scratch (gof_create)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self.device = device
        check(load().gof_create(int(device), C.byref(self._h)))

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("context destroyed")
        return self._h

    @property
    def stream(self):
        return load().gof_stream(self.handle)

    def synchronize(self):
        check(load().gof_synchronize(self.handle))

    def close(self):
        if getattr(self, "_h", None):
            load().gof_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    """Lazily created per-device context used by the cv2-style free functions."""
    if device is None:
        device = int(os.environ.get("GOFPID_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = C.c_int(0)
        load().gof_device_count(C.byref(n))
        if n.value > 0:
            device %= n.value
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx
