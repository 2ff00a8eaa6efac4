"""CPU check of the theory behind the GPU blob extraction (csrc/gof_blobs.hip): the cycles of
the crack successor permutation, with runs of one pixel merged, ARE the borders the
Suzuki-Abe follower (gof_contours.cpp, the checker) traces -- same contours, same order,
same points, same moments, same RETR_EXTERNAL selection."""
import numpy as np
import pytest
from scipy import ndimage

from crack_model import check


def _masks(seed, n):
    rng = np.random.default_rng(seed)
    for it in range(n):
        h, w = int(rng.integers(1, 36)), int(rng.integers(1, 44))
        m = (rng.random((h, w)) < rng.choice([0.2, 0.4, 0.5, 0.6, 0.8, 0.95])).astype(np.uint8) * 255
        if it % 4 == 1 and h > 4 and w > 4:
            m = ndimage.binary_closing(m, np.ones((3, 3))).astype(np.uint8) * 200
        elif it % 4 == 2 and h > 6 and w > 6:
            m = ndimage.binary_opening(ndimage.binary_dilation(m, np.ones((3, 3))), np.ones((3, 3))).astype(np.uint8) * 127
        yield m, int(rng.choice([0, 0, 2, 20, 200]))


def test_crack_cycles_equal_border_following():
    total = sum(check(m, a2) for m, a2 in _masks(0, 160))
    assert total > 200


def test_crack_cycles_nested_and_degenerate():
    m = np.zeros((40, 40), np.uint8)
    m[2:38, 2:38] = 255
    m[5:35, 5:35] = 0
    m[10:30, 10:30] = 255
    m[14:26, 14:26] = 0
    m[17:23, 17:23] = 255
    assert check(m, 0) == 1                       # only the outermost ring is external
    m[20, 2:5] = 0                                # open the ring: the inner square becomes external
    m[20, 0:2] = 0
    assert check(m, 0) >= 1
    for shape in [(1, 1), (1, 7), (7, 1), (2, 2), (3, 3)]:
        assert check(np.full(shape, 255, np.uint8), 0) == 1
    x = np.zeros((9, 9), np.uint8)
    x[np.arange(9), np.arange(9)] = 1             # a diagonal: one 8-connected component
    x[np.arange(9), 8 - np.arange(9)] = 1
    assert check(x, 0) == 1
    chk = (np.indices((8, 8)).sum(0) % 2).astype(np.uint8)
    assert check(chk, 0) == 1                     # checkerboard: one component, many holes
