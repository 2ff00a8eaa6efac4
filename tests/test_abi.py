"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/gofpid_hip.h declares (no compute calls: there is no GPU here),
and the Python host layer mirrors the reference's names and error behaviour."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gofpid_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gof_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pygofpid_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libgofpid_hip.so does not export %s" % n
    # ... and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_strerror_without_gpu():
    from pygofpid_amd import _lib
    lib = _lib.load()
    assert lib.gof_abi_version() == 3
    assert lib.gof_strerror(_lib.EINVAL) == b"invalid argument"
    p = _lib.Mog2Params()
    lib.gof_mog2_default_params(C.byref(p))
    assert (p.history, p.nmixtures, p.detect_shadows, p.shadow_value) == (500, 5, 1, 127)
    assert (p.var_threshold, p.var_threshold_gen, p.var_init, p.var_min, p.var_max) == \
        (16.0, 9.0, 15.0, 4.0, 75.0)
    assert abs(p.background_ratio - 0.9) < 1e-7 and abs(p.ct - 0.05) < 1e-8 and p.tau == 0.5


def test_gauss_taps_host_side_match_oracle(oracle):
    from pygofpid_amd import _lib
    lib = _lib.load()
    for n, s in [(3, 4.0), (3, 0.0), (5, 4.0), (7, 1.5), (9, 0.0), (11, 0.0), (31, 6.0), (1, 1.0)]:
        t = np.zeros(n, np.uint16)
        assert lib.gof_gauss_kernel_q8(n, s, t.ctypes.data_as(C.POINTER(C.c_uint16))) == 0
        assert t.tolist() == oracle.gauss_kernel_q8(n, s).tolist()
    t = np.zeros(4, np.uint16)
    assert lib.gof_gauss_kernel_q8(4, 1.0, t.ctypes.data_as(C.POINTER(C.c_uint16))) == _lib.EINVAL


def test_struct_layout_matches_header():
    from pygofpid_amd import _lib
    assert C.sizeof(_lib.Mog2Params) == 48
    assert C.sizeof(_lib.MorphOp) == 24
    # streams,h,w,c,blur,kw,kh (7 ints, padded to 32) + 2 doubles + detector(+pad) ...
    assert C.sizeof(_lib.PipelineDesc) == 32 + 16 + 4 + 48 + 4 + 8 + 8 + 8 * 24


def test_no_gpu_fails_loudly():
    """Without a device the product path must raise, never fall back to CPU."""
    from pygofpid_amd import _lib
    n = C.c_int(0)
    _lib.load().gof_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible")
    from pygofpid_amd import cv
    with pytest.raises(RuntimeError):
        cv.GaussianBlur(np.zeros((8, 8), np.uint8), (3, 3), 4)
    with pytest.raises(RuntimeError):
        cv.erode(np.zeros((8, 8), np.uint8), np.ones((5, 5), np.uint8))
    with pytest.raises(RuntimeError):
        cv.createBackgroundSubtractorMOG2().apply(np.zeros((8, 8, 3), np.uint8))


def test_structuring_element_glue():
    from pygofpid_amd import cv
    assert np.array_equal(cv.getStructuringElement(cv.MORPH_RECT, (5, 5)), np.ones((5, 5), np.uint8))
    assert cv.getStructuringElement(cv.MORPH_RECT, (3, 7)).shape == (7, 3)
    assert cv.getStructuringElement(cv.MORPH_CROSS, (3, 3)).tolist() == [[0, 1, 0], [1, 1, 1],
                                                                        [0, 1, 0]]
    assert cv.BORDER_DEFAULT == 4 and cv.MORPH_RECT == 0
