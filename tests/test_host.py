"""CPU tests of the host logic around the GPU path: contour glue, geometry
helpers, and GOFPID.initialize() error behaviour (mirrors the reference's
tests/test_pipeline.py error cases and tests/test_helpers.py values)."""
import numpy as np
import pytest

from pygofpid_amd import cv, hostcv
from pygofpid_amd.helpers import (SimpleLinearRegression, find_contours, get_bottom, get_center,
                                  normalize_coords, unnormalize_coords)
from pygofpid_amd.pipeline import GOFPID

perimeter = np.array([[0, 0], [0, 1], [1, 1], [1, 0]])
perspective = np.array([[0.6, 0.5], [0.5, 0.8], [0.7, 0.2], [0.7, 0.3]])
post_filter = {'perimeter': perimeter, 'anchor': 'center', 'perspective': perspective,
               'perspective_coeff': 0.5, 'presence_min': 3, 'distance_min': 0.25}


# ---- contours ---------------------------------------------------------------

def test_find_contours_rectangles():
    m = np.zeros((40, 60), np.uint8)
    m[5:15, 10:30] = 255          # 10 x 20 block
    m[20:35, 40:50] = 127         # grey levels count as foreground
    cs = find_contours(m)
    assert len(cs) == 2
    cs = sorted(cs, key=lambda c: c[:, 0, 1].min())
    a, b = cs
    assert a.shape[1:] == (1, 2) and a.dtype == np.int32
    assert hostcv.boundingRect(a) == (10, 5, 20, 10)
    assert hostcv.boundingRect(b) == (40, 20, 10, 15)
    assert hostcv.contourArea(a) == 19 * 9          # polygon through pixel centres
    assert len(a) == 2 * (20 + 10) - 4              # every border pixel once
    assert a[0, 0].tolist() == [10, 5]              # starts at the raster-first pixel
    # square from the reference's tests/test_helpers.py: centre and bottom
    sq = np.array([[[10, 10]], [[10, 20]], [[20, 20]], [[20, 10]]], np.int32)
    assert get_center(sq).tolist() == [15, 15]
    assert get_bottom(sq).tolist() == [15, 20]


def test_find_contours_external_only_and_shapes():
    m = np.zeros((30, 30), np.uint8)
    m[2:28, 2:28] = 255
    m[6:24, 6:24] = 0             # ring
    m[10:20, 10:20] = 255         # island inside the hole: not external
    cs = find_contours(m)
    assert len(cs) == 1
    assert hostcv.boundingRect(cs[0]) == (2, 2, 26, 26)
    # single pixel, a 1-px line, diagonal 8-connectivity, touching the image border
    m = np.zeros((12, 12), np.uint8)
    m[1, 1] = 1
    m[5, 2:9] = 1
    m[8, 8] = m[9, 9] = m[10, 10] = 1
    m[0, 6:12] = 1
    cs = find_contours(m)
    assert len(cs) == 4
    by_box = {hostcv.boundingRect(c): c for c in cs}
    assert (1, 1, 1, 1) in by_box and len(by_box[(1, 1, 1, 1)]) == 1
    assert (2, 5, 7, 1) in by_box and len(by_box[(2, 5, 7, 1)]) == 12   # there and back
    assert (8, 8, 3, 3) in by_box
    assert (6, 0, 6, 1) in by_box
    assert find_contours(np.zeros((5, 5), np.uint8)) == ()
    assert len(find_contours(np.full((5, 7), 255, np.uint8))) == 1


def test_find_contours_random_matches_scipy_components():
    """One external contour per 8-connected component that touches the outside."""
    from scipy import ndimage
    rng = np.random.default_rng(3)
    for _ in range(10):
        m = (rng.random((40, 50)) > 0.62).astype(np.uint8) * 255
        lab, n = ndimage.label(m, structure=np.ones((3, 3)))
        # outside background: 4-connected zero region reachable from a 1-px frame
        pad = np.pad(m == 0, 1, constant_values=True)
        bl, _ = ndimage.label(pad)
        outside = bl == bl[0, 0]
        touch = ndimage.binary_dilation(outside, structure=ndimage.generate_binary_structure(2, 1))[1:-1, 1:-1]
        external = {lab[y, x] for y, x in zip(*np.nonzero(touch & (m != 0)))}
        cs = find_contours(m)
        assert len(cs) == len(external)
        for c in cs:
            ids = {lab[y, x] for x, y in c[:, 0]}
            assert len(ids) == 1 and ids.pop() in external
            x, y, w, h = hostcv.boundingRect(c)
            comp = np.nonzero(lab == lab[c[0, 0, 1], c[0, 0, 0]])
            assert (x, y, w, h) == (comp[1].min(), comp[0].min(), np.ptp(comp[1]) + 1, np.ptp(comp[0]) + 1)


@pytest.mark.parametrize("shape", [(300, 1003), (97, 131), (64, 33), (1, 70), (70, 1), (9, 8), (33, 64)])
def test_find_contours_sparse_masks(shape):
    """Mostly-empty masks of widths around the 8- and 32-byte strides the scan skips
    zeros with: small blobs at random places (edges and stride boundaries included),
    alternating with other geometries in between (the per-thread label buffer must
    come back clean).  One external contour per 8-connected component, same bounding
    box, and the same answer when asked again."""
    from scipy import ndimage
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    for rep in range(6):
        m = np.zeros((h, w), np.uint8)
        for _ in range(int(rng.integers(1, 9))):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            bh, bw = int(rng.integers(1, 8)), int(rng.integers(1, 12))
            m[y:y + bh, x:x + bw] = rng.choice([1, 127, 255])
        m[rng.random((h, w)) > 0.995] = 255           # isolated pixels
        lab, n = ndimage.label(m, structure=np.ones((3, 3)))
        cs = find_contours(m)
        find_contours(np.full((7, 5 + rep), 255, np.uint8))        # another geometry in between
        assert len(cs) == n        # solid blobs and dots: every component is external
        again = find_contours(m)
        assert len(again) == len(cs) and all(np.array_equal(a, b) for a, b in zip(again, cs))
        seen = set()
        for c in cs:
            ids = {lab[y, x] for x, y in c[:, 0]}
            assert len(ids) == 1
            cid = ids.pop()
            seen.add(cid)
            x, y, bw, bh = hostcv.boundingRect(c)
            comp = np.nonzero(lab == cid)
            assert (x, y, bw, bh) == (comp[1].min(), comp[0].min(), np.ptp(comp[1]) + 1, np.ptp(comp[0]) + 1)
        assert seen == set(range(1, n + 1))


def test_polygon_tests_and_moments():
    sq = np.array([[0, 0], [0, 10], [10, 10], [10, 0]])
    assert hostcv.pointPolygonTest(sq, (5, 5)) == 1
    assert hostcv.pointPolygonTest(sq, (0, 5)) == 0
    assert hostcv.pointPolygonTest(sq, (10, 10)) == 0
    assert hostcv.pointPolygonTest(sq, (11, 5)) == -1
    assert hostcv.pointPolygonTest(sq, np.array([5, -1], np.int16)) == -1
    m = hostcv.moments(sq.reshape(-1, 1, 2))
    assert m["m00"] == 100 and m["m10"] / m["m00"] == 5 and m["m01"] / m["m00"] == 5
    assert hostcv.moments(sq[::-1].reshape(-1, 1, 2))["m00"] == 100      # orientation-free
    assert hostcv.contourArea(sq.reshape(-1, 1, 2)) == 100
    assert hostcv.norm(np.array([3, 4], np.int16)) == 5


def test_coords_and_regression():
    c = unnormalize_coords(np.array([[0.5, 0.5], [1.0, 0.25]]), (100, 200), dtype=np.int32)
    assert c.tolist() == [[100, 50], [200, 25]]
    n = normalize_coords([[100, 50]], (100, 200))
    assert n.tolist() == [[0.5, 0.5]]
    src = np.array([[0.5, 0.5]])
    unnormalize_coords(src, (10, 10))
    assert src.tolist() == [[0.5, 0.5]]                    # caller's array untouched
    slr = SimpleLinearRegression().fit([0, 10], [5, 25])
    assert (slr.coeff, slr.intercept) == (2, 5)
    assert slr.predict_clip([-10, 0, 20], 0, 30).tolist() == [0, 5, 30]
    with pytest.raises(ValueError):
        slr.fit([1, 2, 3], [1, 2, 3])


# ---- GOFPID.initialize(): the reference's error contract --------------------

def test_gofpid_defaults_are_not_shared():
    a, b = GOFPID(), GOFPID()
    assert a.blur is not b.blur and a.mat_morph is not b.mat_morph
    assert a.post_filter is not b.post_filter
    assert a.blur['fun'] is cv.GaussianBlur and a.blur['borderType'] == cv.BORDER_DEFAULT
    assert [d['fun'] for d in a.mat_morph] == [cv.erode, cv.dilate]
    assert a.mat_morph[0]['kernel'].shape == (5, 5)
    g = a.initialize()
    assert g is a and g.tracked_blobs_ == [] and g.input_shape_ is None


def test_gofpid_blur_and_morph_errors():
    with pytest.raises(KeyError):           # reference tests/test_pipeline.py:89-95
        GOFPID(blur={'ksize': (3, 3)}, post_filter=post_filter.copy()).initialize()
    with pytest.raises(KeyError):           # :149-158
        GOFPID(mat_morph=[{'kernel': cv.getStructuringElement(cv.MORPH_RECT, (5, 5))}],
               post_filter=post_filter.copy()).initialize()
    with pytest.raises(ValueError):         # :112-118
        GOFPID(frg_detect='blabla', post_filter=post_filter.copy()).initialize()
    with pytest.raises(KeyError):           # :161-166
        GOFPID(tracking={'blabla': 42}, post_filter=post_filter.copy()).initialize()
    g = GOFPID(blur={'fun': cv.GaussianBlur}, mat_morph=[{'fun': cv.erode}],
               post_filter=post_filter.copy()).initialize()
    assert g.blur['ksize'] == (3, 3) and g.blur['borderType'] == cv.BORDER_DEFAULT
    assert g.mat_morph[0]['kernel'].shape == (5, 5)
    with pytest.raises(NotImplementedError):
        GOFPID(frg_detect='ViBe', post_filter=post_filter.copy()).initialize()
    GOFPID(frg_detect='KNN', post_filter=post_filter.copy()).initialize()   # built since round 3


@pytest.mark.parametrize("drop_or_set", [
    ('drop', 'perimeter'), ('set', 'perimeter', np.array([[1, 5], [3, 9]])), ('drop', 'anchor'),
    ('set', 'anchor', 'blabla'), ('drop', 'perspective'),
    ('set', 'perspective', np.array([[1, 5, 2], [3, 9, 4]])), ('drop', 'perspective_coeff'),
    ('drop', 'presence_min'), ('drop', 'distance_min'), ('set', 'perimeter', [[0, 0], [1, 1], [1, 0]]),
])
def test_gofpid_postfilter_errors(drop_or_set):
    """reference tests/test_pipeline.py:188-264"""
    pf = post_filter.copy()
    if drop_or_set[0] == 'drop':
        del pf[drop_or_set[1]]
    else:
        pf[drop_or_set[1]] = drop_or_set[2]
    with pytest.raises((KeyError, ValueError)):
        GOFPID(post_filter=pf).initialize()


def test_gofpid_alarmdef_errors_and_headless():
    with pytest.raises(KeyError):
        GOFPID(alarm_def={'keep_alarm': True}).initialize()
    with pytest.raises(NotImplementedError):
        GOFPID(post_filter=dict(post_filter, display_config=True)).initialize()


@pytest.mark.parametrize("dtype", [np.int16, np.uint16, np.int32, np.uint32])
def test_blob_records_reproduce_reference_pins(dtype):
    """The reference's own value pins for step 4's geometry (tests/test_helpers.py:65-80: the square
    contour [[0,0],[0,10],[10,10],[10,0]] has bottom (5, 10) and center (5, 5)), through the record
    arithmetic the GPU path uses (blobs.moments_of / bottom_and_center on exact shoelace sums)."""
    from pygofpid_amd import blobs
    from pygofpid_amd.helpers import get_bottom, get_center
    contour = np.array([[0, 0], [0, 10], [10, 10], [10, 0]], dtype=np.int32)
    assert get_bottom(contour, dtype=dtype).tolist() == [5, 10] and get_center(contour, dtype=dtype).tolist() == [5, 5]
    mask = np.zeros((16, 20), np.uint8)
    mask[0:11, 0:11] = 255                       # its border pixels trace that square (40 points)
    recs, pts = blobs.host_result(mask, 100.0)
    assert len(recs) == 1 and recs[0]["n_points"] == 40 and abs(recs[0]["a00"]) == 200
    bottom, center = blobs.bottom_and_center(recs[0], dtype=dtype)
    assert bottom.dtype == dtype and bottom.tolist() == [5, 10] and center.tolist() == [5, 5]
    assert blobs.moments_of(recs[0]) == hostcv.moments(pts.reshape(-1, 1, 2))


def test_tracking_distances_all_pairs_equal_the_scalar_rule():
    """GOFPID._track_blob evaluates all (blob, tracked blob) gate distances at once; element for
    element that is _compute_tracking_distance (reference pipeline.py:630-647): int16 centre
    differences (wrapping squares included), int32 radii, float64 quotients, 0 for an empty gate."""
    import warnings
    from pygofpid_amd.pipeline import GOFPID
    rng = np.random.default_rng(3)
    g = GOFPID().initialize()
    g.input_shape_ = (480, 640, 3)
    g._calib_first_frame()
    mk = lambda: {'center': rng.integers(-300, 900, 2).astype(np.int16),
                  'bottom': rng.integers(-50, 600, 2).astype(np.int16),
                  'contour': np.zeros((4, 1, 2), np.int32), 'anchor': np.zeros(2, np.int16)}
    g.blobs_ = [mk() for _ in range(9)]
    tracked = [dict(mk(), center_first=np.zeros(2, np.int16), presence=1, absence=0,
                    filtered=set(), already_in_alarm=False) for _ in range(7)]
    g.tracked_blobs_ = [dict(t) for t in tracked]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.array([[g._compute_tracking_distance(j, i) for j in range(7)] for i in range(9)], dtype=np.float64)
        got = g._tracking_distances()
        rows = g._tracking_distance_rows()
    assert got.shape == (9, 7) and got.dtype == np.float64
    assert np.array_equal(got, want)
    # ... and the python-int form _track_blob reads (wrapping done by hand)
    assert all(type(v) is float for r in rows for v in r) and np.array_equal(np.array(rows), want)
    # what the python-int form does not cover goes through the array form: a radius whose square wraps to 0
    g.tracking['factor'] = 65536.0 / g._predict_perspective('height', 300)
    g.tracked_blobs_[0]['bottom'] = np.array([0, 300], np.int16)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(np.array(g._tracking_distance_rows()), g._tracking_distances(), equal_nan=True)


def _rect_record(x, y, w, h, off):
    xs, ys = list(range(x, x + w)), list(range(y, y + h))
    p = np.array([(xx, y) for xx in xs] + [(x + w - 1, yy) for yy in ys[1:]] + [(xx, y + h - 1) for xx in xs[-2::-1]]
                 + [(x, yy) for yy in ys[-2:0:-1]], np.int64)
    x0, y0 = np.roll(p[:, 0], 1), np.roll(p[:, 1], 1)
    cr = x0 * p[:, 1] - p[:, 0] * y0
    return (p[0, 0], p[0, 1], p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max(), len(p), off,
            cr.sum(), (cr * (x0 + p[:, 0])).sum(), (cr * (y0 + p[:, 1])).sum()), p.astype(np.int32)


def test_host_steps_python_scalar_forms_equal_the_numpy_forms():
    """Steps 5-7 (tracking, post-filter, alarm) on python ints / floats vs the numpy forms they
    restate (`_slow_host_steps`): the same tracked blobs, flags and alarms, frame after frame, on
    device-style blob records of objects that appear, cross, leave the perimeter and vanish."""
    from pygofpid_amd import blobs as gb
    from pygofpid_amd.pipeline import GOFPID
    rng = np.random.default_rng(5)
    objs = [(int(rng.integers(0, 560)), int(rng.integers(0, 380)), int(rng.integers(-4, 5)), int(rng.integers(-2, 3)),
             int(rng.integers(12, 60)), int(rng.integers(12, 90)), int(rng.integers(0, 60)), int(rng.integers(80, 300)))
            for _ in range(14)]
    frames = []
    for t in range(300):
        recs, pts, off = [], [], 0
        for (x0, y0, vx, vy, w, h, t0, t1) in objs:
            x, y = x0 + vx * t, y0 + vy * t
            if t0 <= t < t1 and t % 11 != 3 and 0 <= x and x + w <= 640 and 0 <= y and y + h <= 480:
                r, p = _rect_record(x, y, w, h, off)
                recs.append(r), pts.append(p)
                off += len(p)
        frames.append((np.array(recs, gb.RECORD_DTYPE) if recs else np.zeros(0, gb.RECORD_DTYPE),
                       np.concatenate(pts) if pts else np.zeros((0, 2), np.int32)))
    runs = []
    for slow in (True, False):
        g = GOFPID(alarm_def={'keep_alarm': True, 'duration_min': 5}).initialize()
        g._slow_host_steps = slow
        g.input_shape_ = (480, 640, 3)
        g._calib_first_frame()
        log = []
        for res in frames:
            g._gpu_blobs = res
            g._find_blob(), g._track_blob(), g._post_filter()
            alarm = g._maintain_alarm(g._detect_blob())
            log.append((alarm, [(t['center'].tolist(), t['bottom'].tolist(), t['anchor'].tolist(), t['center_first'].tolist(),
                                 t['presence'], t['absence'], sorted(t['filtered']), t['already_in_alarm'], t['bbox'])
                                for t in g.tracked_blobs_]))
        runs.append(log)
    assert runs[0] == runs[1]
    assert any(a for a, _ in runs[0]) and not all(a for a, _ in runs[0]) and max(len(t) for _, t in runs[0]) >= 5


def test_bottoms_and_centers_vectorised_equals_per_record():
    from pygofpid_amd import blobs as gb
    rng = np.random.default_rng(11)
    mask = np.zeros((200, 300), np.uint8)
    for _ in range(12):
        x, y = int(rng.integers(0, 260)), int(rng.integers(0, 170))
        mask[y:y + int(rng.integers(8, 30)), x:x + int(rng.integers(8, 40))] = 255
    recs, pts = gb.host_result(mask, 20.0)
    assert len(recs) > 3
    bottoms, centers = gb.bottoms_and_centers(recs)
    for i, r in enumerate(recs):
        b, c = gb.bottom_and_center(r)
        assert np.array_equal(bottoms[i], b) and np.array_equal(centers[i], c) and bottoms.dtype == b.dtype
        x0, y0, w, h = hostcv.boundingRect(gb.contours_of(recs, pts)[i])
        assert (x0, y0, w, h) == (r['xmin'], r['ymin'], r['xmax'] - r['xmin'] + 1, r['ymax'] - r['ymin'] + 1)
