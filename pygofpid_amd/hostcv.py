"""Host-side stand-ins for the cv2 calls of GOFPID.detect steps 4-7 and display().

None of this is on the GPU hot path (out of
scope, scalar per-blob work).  It exists so that `GOFPID.detect()` runs end to
end where cv2 is not installed.  Semantics follow cv2's documented behaviour
[OPENCV-RECALL]; results the reference consumes (areas, centroids, bounding
boxes, inside/outside tests) are exact integer or double arithmetic.
"""
import ctypes as C

import numpy as np

from . import _lib


# Order of a returned contour list.  cv2's findContours meets the borders in raster order of their
# first pixel but hands the list back REVERSED -- the border found last comes first [OPENCV-RECALL;
# unpinned: neither the reference's tests nor its files hold a contour list].  GOFPID._track_blob
# is greedy over blobs_ in that order (reference pipeline.py:614-622), so the order can decide
# which track a blob takes over.  "cv2": reversed discovery order (the default since round 3);
# "raster": order of discovery (what rounds 1-2 returned).  The C ABI (gof_find_external_contours,
# gof_blobs) always reports raster order; this switch is applied where its results become lists.
CONTOUR_ORDER = "cv2"


def in_contour_order(items):
    """`items` (a sequence in raster order of the borders' first pixels) in CONTOUR_ORDER."""
    return items[::-1] if CONTOUR_ORDER == "cv2" else items


def findContours(mask):
    """cv.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE)[contours]:
    a list of (n, 1, 2) int32 arrays of (x, y) border points, in CONTOUR_ORDER
    (reference call: helpers.py:139-143)."""
    m = _lib.as_u8(mask, "mask")
    if m.ndim == 3 and m.shape[2] == 1:
        m = np.ascontiguousarray(m[:, :, 0])
    if m.ndim != 2:
        raise ValueError("findContours needs a single-channel mask")
    pts = C.POINTER(C.c_int32)()
    offs = C.POINTER(C.c_int32)()
    n = C.c_int()
    lib = _lib.load()
    _lib.check(lib.gof_find_external_contours(_lib.p8(m), m.shape[0], m.shape[1], C.byref(pts),
                                              C.byref(offs), C.byref(n)))
    try:
        o = np.ctypeslib.as_array(offs, shape=(n.value + 1,)).copy()
        total = int(o[-1])
        p = (np.ctypeslib.as_array(pts, shape=(max(total, 1) * 2,))[:total * 2].copy()
             .reshape(total, 1, 2))
    finally:
        lib.gof_free(pts)
        lib.gof_free(offs)
    return in_contour_order([p[o[k]:o[k + 1]] for k in range(n.value)])


def _xy(contour):
    c = np.asarray(contour)
    return c.reshape(-1, 2).astype(np.float64)


def _next(v):
    """np.roll(v, -1) without its generic machinery (these run per blob and frame)."""
    out = np.empty_like(v)
    out[:-1] = v[1:]
    out[-1] = v[0]
    return out


def _prev(v):
    """np.roll(v, 1)"""
    out = np.empty_like(v)
    out[1:] = v[:-1]
    out[0] = v[-1]
    return out


def contourArea(contour):
    """|shoelace| / 2 (cv.contourArea, oriented=False)."""
    p = _xy(contour)
    if len(p) < 3:
        return 0.0
    x, y = p[:, 0], p[:, 1]
    return abs(float(np.dot(x, _next(y)) - np.dot(_next(x), y))) * 0.5


def moments(contour):
    """Spatial moments m00, m10, m01 of a polygon by Green's theorem
    (cv.moments on a contour); sign-normalised so that m00 >= 0."""
    p = _xy(contour)
    if len(p) == 0:
        return {"m00": 0.0, "m10": 0.0, "m01": 0.0}
    x0, y0 = _prev(p[:, 0]), _prev(p[:, 1])   # previous vertex
    x1, y1 = p[:, 0], p[:, 1]
    cross = x0 * y1 - x1 * y0
    a00 = cross.sum()
    a10 = (cross * (x0 + x1)).sum()
    a01 = (cross * (y0 + y1)).sum()
    s = -1.0 if a00 < 0 else 1.0
    return {"m00": s * a00 / 2.0, "m10": s * a10 / 6.0, "m01": s * a01 / 6.0}


def boundingRect(contour):
    p = np.asarray(contour).reshape(-1, 2)
    x0, y0 = int(p[:, 0].min()), int(p[:, 1].min())
    return x0, y0, int(p[:, 0].max()) - x0 + 1, int(p[:, 1].max()) - y0 + 1


def polygon_points(contour):
    """The polygon's vertices as a list of [x, y] python floats (float64 values): what
    point_in_polygon walks; python float arithmetic is the float64 arithmetic numpy would do,
    without a numpy scalar per operation."""
    return _xy(contour).tolist()


def point_in_polygon(p, px, py):
    """pointPolygonTest(., ., False) on polygon_points(): +1 inside, 0 on an edge, -1 outside."""
    n = len(p)
    if n == 0:
        return -1.0
    inside = False
    x0, y0 = p[-1]
    for x1, y1 in p:
        # on the segment?
        cross = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0)
        if cross == 0 and min(x0, x1) <= px <= max(x0, x1) and min(y0, y1) <= py <= max(y0, y1):
            return 0.0
        if (y0 > py) != (y1 > py):
            xi = x0 + (py - y0) * (x1 - x0) / (y1 - y0)
            if px < xi:
                inside = not inside
        x0, y0 = x1, y1
    return 1.0 if inside else -1.0


def pointPolygonTest(contour, pt, measureDist=False):
    """+1 inside, 0 on an edge, -1 outside (measureDist=False only; that is the
    only form the reference uses, pipeline.py:739-743, helpers.py:63-74)."""
    if measureDist:
        raise NotImplementedError("measureDist=True is not used by the reference")
    return point_in_polygon(polygon_points(contour), float(pt[0]), float(pt[1]))


def norm(v):
    return float(np.sqrt(np.sum(np.asarray(v, dtype=np.float64) ** 2)))


# ---- drawing (display() only; visualisation, in place like cv2) ------------

def _put(img, x, y, color):
    if 0 <= y < img.shape[0] and 0 <= x < img.shape[1]:
        img[y, x] = color if img.ndim == 3 else color[0]


def _color(img, color):
    c = tuple(int(v) for v in (color if np.ndim(color) else (color,)))
    if img.ndim == 3:
        return (c + (0,) * img.shape[2])[:img.shape[2]]
    return c


def line(img, p0, p1, color, thickness=1):
    color = _color(img, color)
    x0, y0, x1, y1 = int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])
    dx, dy = abs(x1 - x0), -abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err = dx + dy
    r = max(0, int(thickness) // 2)
    while True:
        for oy in range(-r, r + 1):
            for ox in range(-r, r + 1):
                _put(img, x0 + ox, y0 + oy, color)
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy


def drawContours(img, contours, idx, color, thickness=1):
    for k, c in enumerate(contours):
        if idx >= 0 and k != idx:
            continue
        p = np.asarray(c).reshape(-1, 2)
        for i in range(len(p)):
            line(img, p[i - 1], p[i], color, thickness)


def circle(img, center, radius, color, thickness=1):
    color = _color(img, color)
    cx, cy, r = int(center[0]), int(center[1]), int(radius)
    for y in range(cy - r, cy + r + 1):
        for x in range(cx - r, cx + r + 1):
            d2 = (x - cx) ** 2 + (y - cy) ** 2
            if (thickness < 0 and d2 <= r * r) or (thickness >= 0 and abs(np.sqrt(d2) - r) < 0.5 + thickness / 2):
                _put(img, x, y, color)


def rectangle(img, p0, p1, color, thickness=1):
    x0, y0, x1, y1 = int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])
    if thickness < 0:
        c = _color(img, color)
        for y in range(min(y0, y1), max(y0, y1) + 1):
            for x in range(min(x0, x1), max(x0, x1) + 1):
                _put(img, x, y, c)
        return
    line(img, (x0, y0), (x1, y0), color, thickness)
    line(img, (x1, y0), (x1, y1), color, thickness)
    line(img, (x1, y1), (x0, y1), color, thickness)
    line(img, (x0, y1), (x0, y0), color, thickness)


def ellipse(img, center, axes, angle, start, end, color, thickness=1):
    a, b = float(axes[0]), float(axes[1])
    if a <= 0 or b <= 0:
        return
    t = np.deg2rad(np.arange(start, end + 1, 2.0))
    ca, sa = np.cos(np.deg2rad(angle)), np.sin(np.deg2rad(angle))
    xs = center[0] + a * np.cos(t) * ca - b * np.sin(t) * sa
    ys = center[1] + a * np.cos(t) * sa + b * np.sin(t) * ca
    p = np.stack([np.rint(xs), np.rint(ys)], axis=1).astype(int)
    for i in range(1, len(p)):
        line(img, p[i - 1], p[i], color, thickness)
