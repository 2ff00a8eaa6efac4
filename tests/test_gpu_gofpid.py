"""GPU tests of the drop-in GOFPID class: the reference's own smoke tests
(tests/test_pipeline.py) re-run against pygofpid_amd, plus value checks of
`foreground_mask_` against the CPU oracle and an end-to-end intrusion scenario."""
import numpy as np
import pytest

from pygofpid_amd.synth import Clip

pytestmark = pytest.mark.gpu

n_reps = 5
perimeter = np.array([[0, 0], [0, 1], [1, 1], [1, 0]])
perspective = np.array([[0.6, 0.5], [0.5, 0.8], [0.7, 0.2], [0.7, 0.3]])
post_filter = {'perimeter': perimeter, 'anchor': 'center', 'perspective': perspective,
               'perspective_coeff': 0.5, 'presence_min': 3, 'distance_min': 0.25}


@pytest.fixture(scope="module")
def G():
    from pygofpid_amd.pipeline import GOFPID
    return GOFPID


@pytest.fixture(scope="module")
def cv():
    from pygofpid_amd import cv as _cv
    return _cv


def test_gofpid_errors(G):
    """reference tests/test_pipeline.py:31-41"""
    np.random.seed(17)
    gofpid = G(post_filter=post_filter.copy()).initialize()
    gofpid.detect(np.random.randint(0, high=255, size=(64, 64, 3), dtype=np.uint8))
    with pytest.raises(ValueError):
        gofpid.detect(np.random.randint(0, high=255, size=(32, 32, 3), dtype=np.uint8))


@pytest.mark.parametrize("code", [None, "BGR2GRAY", "RGB2GRAY"])
def test_gofpid_convert(G, cv, oracle, code):
    """reference :44-55, plus the value of the converted + processed mask"""
    convert = None if code is None else getattr(cv, "COLOR_" + code)
    gofpid = G(convert=convert, post_filter=post_filter.copy()).initialize()
    clip = Clip(64, 64, 3, seed=5, n_objects=2)
    ref = oracle.Pipeline(64, 64, 3 if code is None else 1)
    for t in range(6):
        img = clip.frame(t)
        assert gofpid.detect(img) in (True, False)
        x = img if code is None else oracle.cvt_gray(img, rgb_order=(code == "RGB2GRAY"))
        assert np.array_equal(gofpid.foreground_mask_, ref.apply(x))


def test_cvtcolor_values(cv, oracle):
    x = np.random.default_rng(0).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(cv.cvtColor(x, cv.COLOR_BGR2GRAY), oracle.cvt_gray(x))
    assert np.array_equal(cv.cvtColor(x, cv.COLOR_RGB2GRAY), oracle.cvt_gray(x, True))
    w = np.full((2, 2, 3), 255, np.uint8)
    assert (cv.cvtColor(w, cv.COLOR_BGR2GRAY) == 255).all()       # coefficients sum to 2^14
    px = np.array([[[255, 0, 0]]], np.uint8)
    assert cv.cvtColor(px, cv.COLOR_BGR2GRAY)[0, 0] == 29         # round(255 * 1868 / 16384)
    assert cv.cvtColor(px, cv.COLOR_RGB2GRAY)[0, 0] == 76


@pytest.mark.parametrize("size", [(64, 50), (64, 32, 1), (32, 20, 3)])
@pytest.mark.parametrize("blur", ["none", "gauss", "gauss_default_keys", "gauss5", "box_replicate",
                                  "box"])
def test_gofpid_blur(G, cv, oracle, size, blur):
    """reference :58-86, all four of its blur configurations plus two more"""
    cfg = {"none": None,
           "gauss": {'fun': cv.GaussianBlur, 'ksize': (3, 3), 'borderType': cv.BORDER_DEFAULT},
           "gauss_default_keys": {'fun': cv.GaussianBlur},
           "gauss5": {'fun': cv.GaussianBlur, 'ksize': (5, 5), 'borderType': cv.BORDER_DEFAULT},
           "box_replicate": {'fun': cv.blur, 'borderType': cv.BORDER_REPLICATE},
           "box": {'fun': cv.blur, 'ksize': (3, 3)}}[blur]
    gofpid = G(blur=cfg, post_filter=post_filter.copy()).initialize()
    np.random.seed(17)
    h, w = size[:2]
    c = 1 if len(size) == 2 else size[2]
    ref = oracle.Pipeline(h, w, c, blur_sigma=4.0,
                          blur_k={"none": 0, "gauss5": 5, "box": -3, "box_replicate": -3}.get(blur, 3))
    for _ in range(3):
        img = np.random.randint(0, high=255, size=size, dtype=np.uint8)
        gofpid.detect(img)
        assert gofpid.foreground_mask_.shape == size[:2]
        assert np.array_equal(gofpid.foreground_mask_, ref.apply(img))


@pytest.mark.parametrize("size", [(32, 20), (20, 32, 1), (32, 20, 3)])
@pytest.mark.parametrize("frg_detect", ['MOG2', 'KNN', 'FD'])
def test_gofpid_frgdetect(G, oracle, size, frg_detect):
    """reference :98-109 for the three detectors built (ViBe raises NotImplementedError)"""
    gofpid = G(frg_detect=frg_detect, post_filter=post_filter.copy()).initialize()
    np.random.seed(17)
    h, w = size[:2]
    c = 1 if len(size) == 2 else size[2]
    ref = oracle.Pipeline(h, w, c, detector=frg_detect, fd_thr=5)
    for _ in range(n_reps):
        img = np.random.randint(0, high=255, size=size, dtype=np.uint8)
        gofpid.detect(img)
        assert np.array_equal(gofpid.foreground_mask_, ref.apply(img))


@pytest.mark.parametrize("size", [(64, 50), (20, 32, 1), (32, 20, 3)])
@pytest.mark.parametrize("with_morph", [False, True])
def test_gofpid_matmorph(G, cv, oracle, size, with_morph):
    """reference :121-146"""
    mm = None
    if with_morph:
        mm = [{'fun': cv.erode, 'kernel': cv.getStructuringElement(cv.MORPH_RECT, (5, 5))},
              {'fun': cv.dilate, 'kernel': cv.getStructuringElement(cv.MORPH_RECT, (5, 5))}]
    gofpid = G(mat_morph=mm, post_filter=post_filter.copy()).initialize()
    np.random.seed(17)
    h, w = size[:2]
    c = 1 if len(size) == 2 else size[2]
    ref = oracle.Pipeline(h, w, c, morph=(("erode", 5), ("dilate", 5)) if with_morph else ())
    img = np.random.randint(0, high=255, size=size, dtype=np.uint8)
    gofpid.detect(img)
    assert np.array_equal(gofpid.foreground_mask_, ref.apply(img))


def test_gofpid_generic_callables_match_fused(G, cv):
    """A user callable in any slot switches detect() to the reference's own
    call sequence; the result must equal the fused path."""
    calls = []

    def my_blur(X, ksize, borderType):
        calls.append((ksize, borderType))
        return cv.GaussianBlur(X, ksize, borderType)

    a = G(post_filter=post_filter.copy()).initialize()
    b = G(blur={'fun': my_blur}, post_filter=post_filter.copy()).initialize()
    clip = Clip(48, 64, 3, seed=9)
    for t in range(8):
        f = clip.frame(t)
        a.detect(f)
        b.detect(f)
        assert np.array_equal(a.foreground_mask_, b.foreground_mask_)
    assert a._fused is not None and b._fused is None
    assert calls[0] == ((3, 3), cv.BORDER_DEFAULT)      # positional args as the reference passes them


@pytest.mark.parametrize("anchor", ['center', 'bottom'])
def test_gofpid_postfilter(G, anchor):
    """reference :169-185"""
    gofpid = G(post_filter=dict(post_filter, anchor=anchor)).initialize()
    np.random.seed(17)
    for _ in range(n_reps):
        gofpid.detect(np.random.randint(0, high=255, size=(64, 50), dtype=np.uint8))


@pytest.mark.parametrize("keep_alarm", [True, False])
def test_gofpid_alarmdef(G, keep_alarm):
    """reference :267-279 (default-constructed post_filter, twice: per-instance copies)"""
    for _ in range(2):
        gofpid = G(alarm_def={'keep_alarm': keep_alarm, 'duration_min': 3}).initialize()
        np.random.seed(17)
        for _ in range(n_reps):
            gofpid.detect(np.random.randint(0, high=255, size=(64, 50), dtype=np.uint8))


def test_gofpid_intrusion_scenario(G):
    """A person-sized bright box walks across a static, noisy background: no alarm
    while the scene is empty, an alarm once it has been tracked long enough, and
    the alarm is held for duration_min frames after it leaves."""
    h, w = 240, 320                       # BASELINE config 1 geometry
    rng = np.random.default_rng(17)
    bg = np.clip(90 + 30 * np.sin(np.arange(w) / 25.0)[None, :] + rng.integers(-6, 7, (h, w)), 0, 255)
    gofpid = G(alarm_def={'keep_alarm': True, 'duration_min': 5}).initialize()
    alarms = []
    for t in range(110):
        f = bg + rng.normal(0, 1.5, (h, w))
        if 40 <= t < 90:
            x = 10 + 5 * (t - 40)
            f[120:200, x:x + 36] += 90    # 80 x 36 px "person"
        frame = np.repeat(np.clip(np.rint(f), 0, 255).astype(np.uint8)[:, :, None], 3, axis=2)
        alarms.append(bool(gofpid.detect(frame)))
        assert gofpid.foreground_mask_.shape == (h, w)
        if 45 <= t < 85:
            assert len(gofpid.blobs_) >= 1
    assert not any(alarms[5:40])          # empty scene after the model has settled
    assert any(alarms[40:60])             # raised within 20 frames of the entry
    assert all(alarms[60:90])
    assert alarms[90]                     # held after the object left
    canvas = frame.copy()
    gofpid.display(canvas, {'tracking': True, 'perspective': True})
    assert canvas.shape == frame.shape


def test_ilids_view_configs_masks_bit_exact(oracle):
    """BASELINE.json config 5: the reference's i-LIDS view-1 / view-2 configurations
    (examples/dataset_ilids.py:52-83) on 720x576 synthetic stand-ins, streams
    sharded round-robin as over 8 GPUs; every foreground mask equals the CPU oracle's."""
    import importlib.util
    import os
    from pygofpid_amd import distributed as gd
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                        "ilids_synthetic.py")
    spec = importlib.util.spec_from_file_location("ilids_synthetic", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert len(mod.VIDEO_NAMES) == 33
    shards = [gd.shard_streams(33, 8, r) for r in range(8)]
    assert sorted(sum(shards, [])) == list(range(33))
    for s in shards[3][:2] + shards[7][:1]:            # a few streams of two ranks
        name = mod.VIDEO_NAMES[s]
        gofpid = mod.get_gofpid(name[5])
        ref = oracle.Pipeline(576, 720, 3)
        clip = Clip(576, 720, 3, seed=100 + s, n_objects=2)
        for t in range(6):
            f = clip.frame(t)
            assert gofpid.detect(f) in (True, False)
            assert np.array_equal(gofpid.foreground_mask_, ref.apply(f))
        assert gofpid.post_filter['perimeter'].dtype == np.int32      # calibrated to pixels


def test_gofpid_detect_many_equals_detect(G):
    """detect_many (one upload, multi-frame kernels, one download of all blob records) gives
    what frame-by-frame detect() gives: return values, blobs, tracks, masks"""
    h, w = 180, 320
    clip = Clip(h, w, 3, seed=29, n_objects=3)
    a = G(post_filter=post_filter.copy()).initialize()
    b = G(post_filter=post_filter.copy()).initialize()
    want = [a.detect(clip.frame(t)) for t in range(27)]
    got = []
    for t0, n in ((0, 1), (1, 8), (9, 8), (17, 3), (20, 7)):
        got += b.detect_many([clip.frame(t) for t in range(t0, t0 + n)])
    assert got == want
    assert len(a.tracked_blobs_) == len(b.tracked_blobs_) and len(a.blobs_) == len(b.blobs_)
    for x, y in zip(a.blobs_, b.blobs_):
        assert np.array_equal(x['contour'], y['contour']) and np.array_equal(x['center'], y['center'])
    for x, y in zip(a.tracked_blobs_, b.tracked_blobs_):
        assert x['presence'] == y['presence'] and x['filtered'] == y['filtered']
    assert np.array_equal(a.foreground_mask_, b.foreground_mask_)
    with pytest.raises(ValueError):
        b.detect_many([clip.frame(0)[:-1]])


def test_gofpid_detect_many_long_list_is_chunked_and_pipelined(G):
    """A list far longer than one device batch (8 frames): detect_many works through it in
    pipelined batches -- bounded memory, the upload of batch k + 1 beside the kernels of batch k
    and the host steps of batch k -- with the results of frame-by-frame detect(); a frame of
    another shape anywhere in the list is refused BEFORE any frame is consumed."""
    h, w = 120, 200
    clip = Clip(h, w, 3, seed=31, n_objects=3)
    frames = [clip.frame(t) for t in range(61)]
    a = G(post_filter=post_filter.copy()).initialize()
    b = G(post_filter=post_filter.copy()).initialize()
    want = [a.detect(f) for f in frames]
    bad = frames[:20] + [frames[20][:, :-4]] + frames[21:]
    with pytest.raises(ValueError):
        b.detect_many(bad)
    assert not b.input_shape_ and b._fused is None          # nothing was consumed
    assert b.detect_many(frames[:37]) == want[:37]          # first call: 1 + 36 frames = 5 batches
    assert b._fused._slot_blobs[0].n_images <= 8 and b._fused._slot_blobs[1].n_images <= 8
    assert b.detect_many(frames[37:]) == want[37:]
    assert np.array_equal(a.foreground_mask_, b.foreground_mask_)
    for x, y in zip(a.tracked_blobs_, b.tracked_blobs_):
        assert x['presence'] == y['presence'] and x['filtered'] == y['filtered']
        assert np.array_equal(x['contour'], y['contour'])
    assert b.detect(frames[5]) == a.detect(frames[5])       # and back to single frames


def test_gofpid_detect_many_survives_a_failure_mid_list(G):
    """An exception in the calling thread's host steps (or in the upload thread) comes out of
    detect_many, the upload thread is released -- no hang, no leaked batch in flight that a later
    call trips over -- and the object keeps working: later frames give what a twin fed frame by
    frame gives from the same model state onwards (MOG2 has taken the batches that were submitted)."""
    h, w = 96, 160
    clip = Clip(h, w, 3, seed=5, n_objects=2)
    frames = [clip.frame(t) for t in range(80)]
    g = G(post_filter=post_filter.copy()).initialize()
    assert len(g.detect_many(frames[:9])) == 9
    calls = {"n": 0}
    real = g._post_filter

    def failing():
        calls["n"] += 1
        if calls["n"] == 11:
            raise RuntimeError("host step failed")
        return real()
    g._post_filter = failing
    with pytest.raises(RuntimeError, match="host step failed"):
        g.detect_many(frames[9:60])
    g._post_filter = real
    # a bad frame reaches the upload thread only through the device call: shapes are checked up front,
    # so force the failure there
    sub = g._fused.submit_blobs_frames
    g._fused.submit_blobs_frames = lambda fr, a, s: (_ for _ in ()).throw(MemoryError("upload failed")) if s == 1 else sub(fr, a, s)
    with pytest.raises(MemoryError, match="upload failed"):
        g.detect_many(frames[60:78])
    g._fused.submit_blobs_frames = sub
    out = g.detect_many(frames[60:80])                      # works again, batches and all
    assert len(out) == 20 and all(isinstance(v, (bool, np.bool_)) for v in out)
    assert isinstance(g.detect(frames[3]), (bool, np.bool_)) and g.foreground_mask_.shape == (h, w)
