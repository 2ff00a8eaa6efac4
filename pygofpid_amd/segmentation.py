"""Foreground estimators with the reference's plug-in interface.

Mirrors pygofpid/segmentation.py:11-70: `ForegroundEstimator` is the ABC the
pipeline calls as `self._frg_detect_mth.apply(X)` (pipeline.py:470);
`FrameDifferencing` keeps the reference's constructor and semantics but runs on
the GPU.  ViBe (segmentation.py:73-181) is out of scope: its sequential
RandomState draw order makes it serial by construction.
"""
import ctypes as C
from abc import ABCMeta, abstractmethod

import numpy as np

from . import _lib


class ForegroundEstimator(metaclass=ABCMeta):

    BACKGROUND = 0
    FOREGROUND = 255

    @abstractmethod
    def apply(self, X):
        pass


class FrameDifferencing(ForegroundEstimator):
    """Foreground detection by frame differencing, F = |X_t - X_{t-1}| > threshold.

    Same contract as the reference class (segmentation.py:21-70): the first
    frame gives an all-zero mask; a float threshold is floored (cv.threshold on
    uint8); a colour frame's per-channel 0/255 masks are averaged in uint8, so
    colour masks hold {0, 85, 84} (the reference's np.mean(..., dtype=np.uint8)
    wraps the channel sum).  The previous frame is kept as a device copy.
    """

    def __init__(self, threshold=5, device=None):
        self.threshold = threshold
        self._device = device
        self._h = C.c_void_p()
        self._shape = None

    def apply(self, X):
        x = _lib.as_u8(X, "X")
        x, h, w, c = _lib.hwc(x)
        if self._shape != (h, w, c):
            if self._shape is not None:
                # the reference would fail inside cv.absdiff on a size change
                raise ValueError("Input shape has changed.")
            ctx = _lib.default_context(self._device)
            _lib.check(_lib.load().gof_fd_create(ctx.handle, 1, h, w, c, float(self.threshold),
                                                 C.byref(self._h)))
            self._shape = (h, w, c)
        out = np.empty((h, w), np.uint8)
        _lib.check(_lib.load().gof_fd_apply_host(self._h, _lib.p8(x), _lib.p8(out)))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().gof_fd_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass
