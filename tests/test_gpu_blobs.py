"""GPU tests of step 4's front end (csrc/gof_blobs.hip): external contours, area filter and
blob statistics computed on the device, against the host border follower
(gof_find_external_contours through hostcv -- the checker).  Bar: identical contours in
identical order with identical points, sums, bounding boxes; GOFPID.blobs_ unchanged."""
import numpy as np
import pytest
from scipy import ndimage

from pygofpid_amd.synth import Clip

pytestmark = pytest.mark.gpu


def _same(got, want):
    assert got is not None, "device flagged an overflow / internal error"
    gr, gp = got
    wr, wp = want
    assert len(gr) == len(wr), (len(gr), len(wr))
    for name in gr.dtype.names:
        assert np.array_equal(gr[name], wr[name]), (name, gr[name][:6], wr[name][:6])
    assert gp.shape == wp.shape and np.array_equal(gp, wp)


def _random_mask(rng, h, w, kind):
    m = (rng.random((h, w)) < rng.choice([0.05, 0.2, 0.4, 0.5, 0.6, 0.8, 0.95])).astype(np.uint8) * 255
    if kind == 1 and h > 4 and w > 4:
        m = ndimage.binary_closing(m, np.ones((3, 3))).astype(np.uint8) * 200
    elif kind == 2 and h > 6 and w > 6:
        m = ndimage.binary_opening(ndimage.binary_dilation(m, np.ones((5, 5))), np.ones((3, 3))).astype(np.uint8) * 127
    elif kind == 3 and h > 10 and w > 10:
        m = ndimage.binary_opening(m, np.ones((2, 2))).astype(np.uint8)
    return m


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (9, 1), (2, 2), (7, 5), (33, 65), (64, 64), (67, 131),
                                   (40, 200), (130, 63), (96, 256)])
def test_extract_random_masks(shape):
    from pygofpid_amd import blobs
    h, w = shape
    # raw noise has up to 2 h w border edges: the worst-case scratch (the default, h w / 2,
    # is sized for masks that went through the morphology)
    ex = blobs.BlobExtractor(1, h, w, crack_capacity=2 * h * w + 16)
    rng = np.random.default_rng(h * 1000 + w)
    for it in range(16):
        m = _random_mask(rng, h, w, it % 4)
        for min_area in (0.0, 1.5, 100.0):
            _same(ex.extract(m, min_area)[0], blobs.host_result(m, min_area))
    for m in (np.zeros(shape, np.uint8), np.full(shape, 255, np.uint8)):
        _same(ex.extract(m, 0.0)[0], blobs.host_result(m, 0.0))


def test_extract_structures():
    """nested rings (RETR_EXTERNAL keeps the outermost only), an opened ring, diagonals,
    a checkerboard, blobs touching the frame, one-pixel-wide lines"""
    from pygofpid_amd import blobs
    ex = blobs.BlobExtractor(1, 40, 40, crack_capacity=2 * 40 * 40 + 16)
    m = np.zeros((40, 40), np.uint8)
    m[2:38, 2:38] = 255
    m[5:35, 5:35] = 0
    m[10:30, 10:30] = 255
    m[14:26, 14:26] = 0
    m[17:23, 17:23] = 255
    for a in (0.0, 100.0):
        got = ex.extract(m, a)[0]
        _same(got, blobs.host_result(m, a))
        assert len(got[0]) == 1
    m[20, 0:5] = 0
    _same(ex.extract(m, 0.0)[0], blobs.host_result(m, 0.0))
    x = np.zeros((40, 40), np.uint8)
    x[np.arange(40), np.arange(40)] = 1
    x[np.arange(40), 39 - np.arange(40)] = 1
    x[0, :] = 7
    x[:, 39] = 9
    _same(ex.extract(x, 0.0)[0], blobs.host_result(x, 0.0))
    chk = (np.indices((40, 40)).sum(0) % 2).astype(np.uint8)
    _same(ex.extract(chk, 0.0)[0], blobs.host_result(chk, 0.0))
    iso = np.zeros((40, 40), np.uint8)
    iso[::2, ::2] = 255                                   # 400 isolated pixels
    got = ex.extract(iso, 0.0)[0]
    _same(got, blobs.host_result(iso, 0.0))
    assert len(got[0]) == 400 and (got[0]["n_points"] == 1).all()


def test_extract_batch_and_1080p():
    """several masks per call; 1080p masks with large blobs: borders of > 8192 edges go
    through several list-ranking passes and results larger than the 32-KB head travel on
    demand"""
    from pygofpid_amd import blobs
    h, w = 1080, 1920
    rng = np.random.default_rng(7)
    masks = np.zeros((3, h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    masks[0][(yy - 500) ** 2 / 400.0 ** 2 + (xx - 900) ** 2 / 800.0 ** 2 < 1] = 255     # a huge ellipse
    masks[0][(yy - 500) ** 2 + (xx - 900) ** 2 < 150 ** 2] = 0                         # with a hole
    masks[0][(yy - 500) ** 2 + (xx - 900) ** 2 < 60 ** 2] = 255                        # and an island in it
    noise = ndimage.binary_opening(rng.random((h, w)) < 0.55, np.ones((3, 3)))
    masks[1] = noise.astype(np.uint8) * 255                                            # a ragged tangle
    masks[1, 20:24, 100:1700] = 255                                                    # and a comb: one border
    for k in range(40):                                                                # of ~80 000 points
        masks[1, 24:1040, 100 + 40 * k:103 + 40 * k] = 255
    for k in range(40):                                                                # people-sized boxes
        x0, y0 = int(rng.integers(0, w - 60)), int(rng.integers(0, h - 160))
        masks[2, y0:y0 + int(rng.integers(20, 160)), x0:x0 + int(rng.integers(8, 60))] = 255
    ex = blobs.BlobExtractor(3, h, w, crack_capacity=h * w)
    for a in (100.0, 0.0):
        res = ex.extract(masks, a)
        for i in range(3):
            want = blobs.host_result(masks[i], a)
            if len(want[0]) > 4096:      # more borders than one image's record block takes
                assert res[i] is None
                continue
            assert res[i] is not None, (a, i, len(want[0]), len(want[1]))
            _same(res[i], want)
    assert res[0][0]["n_points"].max() > 3000
    big = ex.extract(masks[1:2], 100.0)[0]
    assert 4 + 14 * len(big[0]) + len(big[1]) > 8192      # beyond the 32-KB head: the on-demand copy ran
    # a sub-batch of a larger scratch
    res = ex.extract(masks[1:], 100.0)
    _same(res[0], blobs.host_result(masks[1], 100.0))
    _same(res[1], blobs.host_result(masks[2], 100.0))


def test_extract_overflow_falls_back():
    from pygofpid_amd import blobs
    h, w = 96, 128
    m = (np.indices((h, w)).sum(0) % 2).astype(np.uint8) * 255     # 2 * h * w border edges
    ex = blobs.BlobExtractor(1, h, w, crack_capacity=4096)
    assert ex.extract(m, 0.0)[0] is None
    small = np.zeros((h, w), np.uint8)
    small[10:30, 10:40] = 255
    _same(ex.extract(small, 100.0)[0], blobs.host_result(small, 100.0))   # the scratch is reusable
    with pytest.raises(ValueError):
        ex.extract(np.zeros((h, w + 1), np.uint8))


def test_pipeline_apply_blobs_matches_host(oracle):
    """steps 1-3 + extraction in one call: masks equal the oracle's, blobs equal the host
    follower's on those masks; one stream and several, one frame and several"""
    from pygofpid_amd import blobs
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, c = 2, 120, 200, 3
    clips = [Clip(h, w, c, seed=40 + s, n_objects=3) for s in range(S)]
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    pipe = ForegroundPipeline((h, w, c), streams=S)
    for t0 in range(0, 24, 4):
        T = 4 if t0 % 8 == 0 else 1
        fr = np.stack([np.stack([cl.frame(t0 + i) for cl in clips]) for i in range(T)])
        res = pipe.apply_blobs(fr, min_area=20.0)
        masks = pipe.last_masks()
        assert masks.shape == (T, S, h, w)
        for i in range(T):
            for s in range(S):
                assert np.array_equal(masks[i, s], refs[s].apply(fr[i, s]))
                _same(res[i * S + s], blobs.host_result(masks[i, s], 20.0))
    assert sum(len(r[0]) for r in res) > 0


def _host_blobs(g):
    from pygofpid_amd import hostcv
    from pygofpid_amd.helpers import find_contours
    return [g._create_blob(c) for c in find_contours(g.foreground_mask_) if hostcv.contourArea(c) >= 100]


@pytest.mark.parametrize("anchor", ["bottom", "center"])
def test_gofpid_blobs_equal_host_path(anchor):
    """GOFPID.blobs_ (count, order, contour, bottom, center, anchor) and everything built on
    them (tracking, alarm) are what the host contour path gives on the same masks"""
    from pygofpid_amd import hostcv
    from pygofpid_amd.pipeline import GOFPID
    pf = {'perimeter': None, 'anchor': anchor, 'perspective': None, 'perspective_coeff': 0.5,
          'presence_min': 3, 'distance_min': 0.25}
    h, w = 240, 320
    clip = Clip(h, w, 3, seed=23, n_objects=4)
    g = GOFPID(post_filter=dict(pf)).initialize()
    ref = GOFPID(post_filter=dict(pf)).initialize()
    n_blobs = 0
    for t in range(40):
        f = clip.frame(t)
        out = g.detect(f)
        assert g._gpu_blobs is not None                 # the device path ran
        want = _host_blobs(g)
        assert len(g.blobs_) == len(want)
        for a, b in zip(g.blobs_, want):
            assert np.array_equal(a['contour'], b['contour'])
            for key in ('bottom', 'center', 'anchor'):
                assert a[key].dtype == b[key].dtype and np.array_equal(a[key], b[key])
            assert hostcv.boundingRect(a['contour']) == hostcv.boundingRect(b['contour'])
        n_blobs += len(want)
        # the same frames through a GOFPID forced onto the host contour path
        ref_out = _detect_on_host_path(ref, f)
        assert out == ref_out
        assert len(g.tracked_blobs_) == len(ref.tracked_blobs_)
    assert n_blobs > 20


def _detect_on_host_path(g, f):
    """GOFPID.detect (pipeline.py:434-493) with the device blob result discarded before
    step 4, so that contours / filter / moments come from the host follower"""
    if not g.input_shape_:
        g.input_shape_ = f.shape
        g._calib_first_frame()
    g._foreground(f)
    g._gpu_blobs = None
    g._find_blob()
    g._track_blob()
    g._post_filter()
    return g._maintain_alarm(g._detect_blob())


def test_apply_blobs_frame_differencing_masks(oracle):
    """FD pipelines leave {0, 84, 85} masks: any non-zero byte is foreground to findContours"""
    from pygofpid_amd import blobs
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 96, 160, 3
    clip = Clip(h, w, c, seed=77, n_objects=3, speed=5)
    pipe = ForegroundPipeline((h, w, c), detector="FD", fd_threshold=5)
    ref = oracle.Pipeline(h, w, c, detector="FD", fd_thr=5.0)
    seen = set()
    for t in range(10):
        f = clip.frame(t)
        res = pipe.apply_blobs(f, min_area=4.0)[0]
        mask = pipe.last_masks()[0, 0]
        assert np.array_equal(mask, ref.apply(f))
        seen |= set(np.unique(mask).tolist())
        _same(res, blobs.host_result(mask, 4.0))
    assert seen - {0}, "the clip produced no foreground at all"


@pytest.mark.parametrize("shape", [(270, 481), (2160, 3840)])
def test_extract_batches_odd_and_4k(shape):
    """batches of smoothed masks at an odd width (byte-wise loads everywhere) and at 4K"""
    from pygofpid_amd import blobs
    h, w = shape
    n = 5 if h < 1000 else 2
    rng = np.random.default_rng(h + w)
    masks = np.zeros((n, h, w), np.uint8)
    for i in range(n):
        m = rng.random((h // 6 + 1, w // 6 + 1)) < (0.25 + 0.1 * i)
        m = np.kron(m, np.ones((6, 6), bool))[:h, :w]                 # blocky blobs, touching diagonally
        masks[i] = ndimage.binary_opening(m, np.ones((3, 3))).astype(np.uint8) * (255 if i % 2 else 127)
    ex = blobs.BlobExtractor(n, h, w)
    res = ex.extract(masks, 100.0)
    for i in range(n):
        want = blobs.host_result(masks[i], 100.0)
        if len(want[0]) > 4096:
            assert res[i] is None
        else:
            _same(res[i], want)
    assert any(r is not None and len(r[0]) > 0 for r in res)


def test_reference_pins_through_the_device_path():
    """reference tests/test_helpers.py:59-80 on the device: find_contours on its random 0/1 mask runs
    (and equals the host follower); the square's bottom / center are (5, 10) / (5, 5)"""
    from pygofpid_amd import blobs
    np.random.seed(17)
    mask = np.random.randint(low=0, high=2, size=(64, 64), dtype=np.uint8)
    ex = blobs.BlobExtractor(1, 64, 64, crack_capacity=2 * 64 * 64 + 16)
    _same(ex.extract(mask, 0.0)[0], blobs.host_result(mask, 0.0))
    sq = np.zeros((64, 64), np.uint8)
    sq[0:11, 0:11] = 1
    recs, pts = ex.extract(sq, 100.0)[0]
    assert len(recs) == 1 and recs[0]["n_points"] == 40
    for dtype in (np.int16, np.uint16, np.int32, np.uint32):
        bottom, center = blobs.bottom_and_center(recs[0], dtype=dtype)
        assert bottom.tolist() == [5, 10] and center.tolist() == [5, 5]


def test_device_extraction_against_the_independent_crack_model():
    """The HIP path meets a checker that shares no code with the product library: the plain-loop
    crack-cycle model of tests/crack_model.py (pure Python: cracks, successor permutation, cycle
    sums, nesting depth) -- contours, order, points, shoelace sums and boxes, on random and
    structured small masks, for several area thresholds."""
    import crack_model
    from pygofpid_amd import blobs, hostcv
    rng = np.random.default_rng(2024)
    n_checked = 0
    for it in range(60):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        m = _random_mask(rng, h, w, it % 4)
        ex = blobs.BlobExtractor(1, h, w, crack_capacity=2 * h * w + 16)
        for min_a2 in (0, 3, 40):
            got = ex.extract(m, min_a2 / 2.0)[0]
            assert got is not None
            recs, pts = got
            want = hostcv.in_contour_order(crack_model.extract(m, min_a2))
            assert len(recs) == len(want), (it, min_a2, len(recs), len(want))
            for r, g in zip(recs, want):
                c = pts[r["point_offset"]:r["point_offset"] + r["n_points"]].reshape(-1, 1, 2)
                assert c.shape == g["pts"].shape and np.array_equal(c, g["pts"])
                assert (int(r["a00"]), int(r["a10"]), int(r["a01"])) == (g["a00"], g["a10"], g["a01"])
                assert (r["xmin"], r["ymin"], r["xmax"], r["ymax"]) == g["bbox"]
                n_checked += 1
    assert n_checked > 150
