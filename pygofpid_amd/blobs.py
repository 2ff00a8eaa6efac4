"""Step 4's front end on the GPU: external contours and blob statistics of the final
foreground mask (gof_blobs in the C ABI, kernels in csrc/gof_blobs.hip).

Replaces, for masks that are already on the device, the reference's
`helpers.find_contours(mask)` (cv.findContours RETR_EXTERNAL / CHAIN_APPROX_NONE,
pygofpid/helpers.py:125-150), the `cv.contourArea(contour) >= area_min` filter of
`GOFPID._find_blob` (pipeline.py:555-572) and the `cv.moments` / `np.max` evaluations of
`get_bottom` / `get_center` (helpers.py:153-168).  Same contours, same order, same points
as the host path (`hostcv.findContours` + the same filter), which stays the checker and
the fall-back for masks too intricate for the device scratch.
"""
import ctypes as C

import numpy as np

from . import _lib

RECORD_DTYPE = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("xmin", "<i4"), ("ymin", "<i4"),
                         ("xmax", "<i4"), ("ymax", "<i4"), ("n_points", "<i4"),
                         ("point_offset", "<i4"), ("a00", "<i8"), ("a10", "<i8"), ("a01", "<i8")])
assert RECORD_DTYPE.itemsize == C.sizeof(_lib.BlobRecord)


class BlobExtractor:
    """Scratch + result buffers for `n_images` masks of (h, w) on one device."""

    def __init__(self, n_images, h, w, ctx=None, device=None, crack_capacity=0):
        self.n_images, self.h, self.w = int(n_images), int(h), int(w)
        self._ctx = ctx if ctx is not None else _lib.default_context(device)
        self._h = C.c_void_p()
        _lib.check(_lib.load().gof_blobs_create(self._ctx.handle, self.n_images, self.h, self.w,
                                                int(crack_capacity), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def extract_device(self, masks_ptr, n_images=None, min_area=100.0, stream=None):
        """Enqueue the extraction on device masks (an int pointer); does not synchronise."""
        _lib.check(_lib.load().gof_blobs_extract_dev(
            self._h, C.c_void_p(masks_ptr), int(n_images or self.n_images), float(min_area),
            C.c_void_p(stream) if stream else None))

    def extract(self, masks, min_area=100.0):
        """Host masks (h, w) or (n, h, w) uint8 -> list of results (see result())."""
        m = _lib.as_u8(masks, "masks")
        if m.ndim == 2:
            m = m[None]
        if m.ndim != 3 or m.shape[1:] != (self.h, self.w) or m.shape[0] > self.n_images:
            raise ValueError("masks must be (n <= %d, %d, %d)" % (self.n_images, self.h, self.w))
        _lib.check(_lib.load().gof_blobs_extract_host(self._h, _lib.p8(m), m.shape[0], float(min_area)))
        self.flags = []
        out = []
        for i in range(m.shape[0]):
            out.append(self.result(i))
            self.flags.append(self.last_flags)
        return out

    def result(self, image):
        """(records, points) of one image of the last extraction -- records: structured
        array (RECORD_DTYPE) in hostcv.CONTOUR_ORDER (cv2's order of the contour list), points: (n_points, 2) int32 (x, y) --
        or None when the device flagged an overflow (fall back to the mask)."""
        nb, npts, fl = C.c_int(), C.c_int(), C.c_int()
        lib = _lib.load()
        _lib.check(lib.gof_blobs_counts(self._h, int(image), C.byref(nb), C.byref(npts), C.byref(fl)))
        self.last_flags = fl.value
        if fl.value & 8:
            # GOF_BLOBS_INTERNAL: a bounded device loop ran out -- a bug, not an overflow.  The
            # caller's fall-back (the host follower on the mask) still gives the right contours,
            # but this must not pass silently as a slow frame.
            import warnings
            self.internal_faults = getattr(self, "internal_faults", 0) + 1
            warnings.warn("gof_blobs: internal fault flag on image %d (flags %#x); falling back to the host "
                          "contour path -- please report" % (image, fl.value), RuntimeWarning, stacklevel=2)
        if fl.value:
            return None
        recs = np.zeros(nb.value, RECORD_DTYPE)
        pts = np.zeros((npts.value, 2), np.int32)
        _lib.check(lib.gof_blobs_fetch(self._h, int(image), recs.ctypes.data_as(C.c_void_p), nb.value,
                                       pts.ctypes.data_as(C.c_void_p), npts.value))
        from . import hostcv
        if hostcv.CONTOUR_ORDER == "cv2" and nb.value > 1:
            # the device reports raster order: records and their point runs into cv2's list order
            recs = np.ascontiguousarray(recs[::-1])
            pts = np.concatenate([pts[o:o + n] for o, n in zip(recs["point_offset"], recs["n_points"])])
            recs["point_offset"] = np.cumsum(recs["n_points"]) - recs["n_points"]
        return recs, pts

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().gof_blobs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def contours_of(records, points):
    """The contours as cv2 returns them: a list of (n, 1, 2) int32 arrays."""
    return [points[o:o + n].reshape(-1, 1, 2)
            for o, n in zip(records["point_offset"].tolist(), records["n_points"].tolist())]


def moments_of(rec):
    """cv.moments(contour) entries m00, m10, m01 from the exact shoelace sums
    (same expressions as hostcv.moments)."""
    s = -1.0 if rec["a00"] < 0 else 1.0
    return {"m00": s * float(rec["a00"]) / 2.0, "m10": s * float(rec["a10"]) / 6.0,
            "m01": s * float(rec["a01"]) / 6.0}


def bottom_and_center(rec, dtype=np.int16):
    """helpers.get_bottom / get_center of the record's contour [REF helpers.py:153-168]."""
    m = moments_of(rec)
    x = int(m["m10"] / m["m00"])
    return (np.array([x, int(rec["ymax"])], dtype=dtype),
            np.array([x, int(m["m01"] / m["m00"])], dtype=dtype))


def _wrapped(rows, dtype):
    # (int64 -> dtype as astype does it: modulo 2^bits, like the per-record np.array(..., dtype) for in-range values)
    return np.array(rows, dtype=np.int64).reshape(-1, 2).astype(dtype)


def bottoms_and_centers(records, dtype=np.int16):
    """bottom_and_center for all records at once: (n, 2) arrays of (x, y); the same float64
    quotients truncated toward zero, element for element.  (Python floats ARE float64 and
    float(int) rounds as the C conversion does: with ~10 records per frame a loop over column
    lists costs a fifth of the equivalent numpy expression's fixed overheads.)"""
    if len(records) == 0:
        z = np.zeros((0, 2), dtype)
        return z, z
    bottoms, centers = [], []
    for a00, a10, a01, ymax in zip(records["a00"].tolist(), records["a10"].tolist(),
                                   records["a01"].tolist(), records["ymax"].tolist()):
        if a00 == 0:
            return _bottoms_and_centers_np(records, dtype)     # (inf / nan quotients: numpy's conversion rules)
        s = -1.0 if a00 < 0 else 1.0
        m00 = s * float(a00) / 2.0
        x = int(s * float(a10) / 6.0 / m00)
        bottoms.append((x, ymax))
        centers.append((x, int(s * float(a01) / 6.0 / m00)))
    return _wrapped(bottoms, dtype), _wrapped(centers, dtype)


def _bottoms_and_centers_np(records, dtype):
    a00 = records["a00"].astype(np.float64)
    s = np.where(records["a00"] < 0, -1.0, 1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        m00, m10, m01 = s * a00 / 2.0, s * records["a10"].astype(np.float64) / 6.0, s * records["a01"].astype(np.float64) / 6.0
        x = (m10 / m00).astype(np.int64)
        cy = (m01 / m00).astype(np.int64)
    return (np.stack([x, records["ymax"].astype(np.int64)], 1).astype(dtype),
            np.stack([x, cy], 1).astype(dtype))


def host_result(mask, min_area=100.0):
    """The same (records, points) from the host contour follower: the checker of the
    device path, and its fall-back."""
    from . import hostcv
    recs, pts, off = [], [], 0
    for c in hostcv.findContours(mask):
        if hostcv.contourArea(c) < min_area:
            continue
        p = c.reshape(-1, 2).astype(np.int64)
        x0, y0 = np.roll(p[:, 0], 1), np.roll(p[:, 1], 1)
        cr = x0 * p[:, 1] - p[:, 0] * y0
        recs.append((p[0, 0], p[0, 1], p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max(),
                     len(p), off, cr.sum(), (cr * (x0 + p[:, 0])).sum(), (cr * (y0 + p[:, 1])).sum()))
        pts.append(p.astype(np.int32))
        off += len(p)
    return (np.array(recs, RECORD_DTYPE) if recs else np.zeros(0, RECORD_DTYPE),
            np.concatenate(pts) if pts else np.zeros((0, 2), np.int32))
