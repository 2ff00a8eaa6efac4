"""Geometry helpers used by GOFPID steps 4-7 (host side, off the GPU path).

Same public names and results as the functions of pygofpid/helpers.py that
the pipeline itself calls (find_contours :125-150, get_bottom :153-159,
get_center :162-168, normalize_coords :171-177, unnormalize_coords :180-186,
SimpleLinearRegression :208-285), on top of pygofpid_amd.hostcv instead of cv2.
"""
import numpy as np

from . import hostcv


def find_contours(mask):
    """External contours of the non-zero regions of `mask` (a tuple, like cv2)."""
    return tuple(hostcv.findContours(mask))


def _centroid_x(contour):
    m = hostcv.moments(contour)
    return m, int(m["m10"] / m["m00"])


def get_bottom(contour, dtype=np.int16):
    """(centroid x, lowest y) -- the image y axis points down, so lowest = max."""
    _, x = _centroid_x(contour)
    return np.array([x, np.max(np.asarray(contour)[..., 1])], dtype=dtype)


def get_center(contour, dtype=np.int16):
    m, x = _centroid_x(contour)
    return np.array([x, int(m["m01"] / m["m00"])], dtype=dtype)


def get_bottom_and_center(contour, dtype=np.int16):
    """(get_bottom(contour), get_center(contour)) from one evaluation of the moments
    (the reference computes them once per call, pipeline.py:575-576: same values)."""
    m, x = _centroid_x(contour)
    bottom = np.array([x, np.max(np.asarray(contour)[..., 1])], dtype=dtype)
    center = np.array([x, int(m["m01"] / m["m00"])], dtype=dtype)
    return bottom, center


def normalize_coords(coords, shape, dtype=np.float64):
    """Pixel (x, y) -> fractions of (width, height)."""
    out = np.array(coords, dtype=dtype)
    out[..., 0] /= shape[1]
    out[..., 1] /= shape[0]
    return out


def unnormalize_coords(ncoords, shape, dtype=np.uint16):
    """Fractions of (width, height) -> pixel (x, y), truncated to `dtype`.
    Unlike the reference this never scales the caller's array in place."""
    out = np.array(ncoords, dtype=np.float64)
    out[..., 0] *= shape[1]
    out[..., 1] *= shape[0]
    return out.astype(dtype)


class SimpleLinearRegression:
    """y = coeff * x + intercept through exactly two samples."""

    def __init__(self):
        self.coeff = None
        self.intercept = None

    def fit(self, x, y):
        if len(x) != 2:
            raise ValueError("Fit only two samples.")
        if len(x) != len(y):
            raise ValueError("Inputs must have the same size.")
        self.coeff = (y[1] - y[0]) / (x[1] - x[0])
        self.intercept = y[0] - self.coeff * x[0]
        return self

    def predict(self, x):
        return self.coeff * np.asarray(x) + self.intercept

    def predict_clip(self, x, a_min=None, a_max=None):
        return np.clip(self.predict(x), a_min=a_min, a_max=a_max)
