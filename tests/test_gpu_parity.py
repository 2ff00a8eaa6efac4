"""GPU parity tests (run on a real MI355X through gpurun): every HIP entry point,
called through the C ABI, against the CPU oracle on the same seeded inputs.

Bar: bit-exact.  Blur / FD / morphology are integer; MOG2 is float32 and the
north star allows 1 ULP on the variance update, but the kernels are built with
FP contraction off and IEEE division, so masks AND model state are compared
for exact equality (tolerance 0).  PARITY UNPINNED vs cv2 itself: see
tests/test_oracle.py.
"""
import os

import numpy as np
import pytest

from pygofpid_amd.synth import Clip, uniform_noise_frames

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cv():
    from pygofpid_amd import cv as _cv
    return _cv


def _rng(seed=17):
    return np.random.default_rng(seed)


# ---- step 1: Gaussian blur ------------------------------------------------

BLUR_SHAPES = [(64, 50), (64, 32, 1), (32, 20, 3), (1, 1), (1, 9, 3), (9, 1), (2, 2, 3), (5, 3, 4),
               (37, 129, 3), (70, 200, 3), (33, 65)]


@pytest.mark.parametrize("shape", BLUR_SHAPES)
@pytest.mark.parametrize("ks", [((3, 3), 4.0), ((5, 5), 4.0), ((7, 3), 1.5), ((3, 3), 0.0),
                                ((1, 1), 2.0), ((9, 9), 0.0), ((31, 31), 6.0)])
def test_gaussian_blur(cv, oracle, shape, ks):
    x = _rng().integers(0, 255, shape, dtype=np.uint8)
    (kw, kh), sigma = ks
    got = cv.GaussianBlur(x, (kw, kh), sigma)
    want = oracle.gaussian_blur(x, (kw, kh), sigma)
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want)


def test_gaussian_blur_reference_call_convention(cv, oracle):
    """blur['fun'](X, ksize, borderType): BORDER_DEFAULT lands in sigmaX
    (pipeline.py:463-467)."""
    x = _rng(1).integers(0, 255, (48, 64, 3), dtype=np.uint8)
    keep = x.copy()
    got = cv.GaussianBlur(x, (3, 3), cv.BORDER_DEFAULT)
    assert np.array_equal(got, oracle.gaussian_blur(x, (3, 3), 4.0))
    assert np.array_equal(x, keep)                       # input not mutated
    assert cv.GaussianBlur(x[:, :, :1], (3, 3), 4).shape == (48, 64)   # (H,W,1) -> (H,W)
    with pytest.raises(ValueError):
        cv.GaussianBlur(x, (4, 4), 1.0)
    with pytest.raises(ValueError):
        cv.GaussianBlur(x.astype(np.float32), (3, 3), 1.0)


def test_gaussian_blur_full_hd(cv, oracle):
    x = _rng(2).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    assert np.array_equal(cv.GaussianBlur(x, (3, 3), 4), oracle.gaussian_blur(x, (3, 3), 4.0))
    # size-independent property: a constant image is a fixed point
    c = np.full((1080, 1920, 3), 201, np.uint8)
    assert (cv.GaussianBlur(c, (5, 5), 4) == 201).all()


@pytest.mark.parametrize("shape", [(64, 50), (64, 32, 1), (32, 20, 3), (1, 1), (3, 70, 3), (130, 200, 3)])
@pytest.mark.parametrize("ks", [(3, 3), (5, 5), (3, 7), (1, 1), (31, 31)])
def test_box_blur(cv, oracle, shape, ks):
    """cv.blur (reference tests/test_pipeline.py:68-75)."""
    x = _rng(4).integers(0, 256, shape, dtype=np.uint8)
    got = cv.blur(x, ks)
    want = oracle.box_blur(x, ks)
    assert got.shape == want.shape and np.array_equal(got, want)
    # the reference's call convention: a border constant in the 3rd slot is ignored
    assert np.array_equal(cv.blur(x, ks, cv.BORDER_REPLICATE), want)


# ---- step 3: morphology ---------------------------------------------------

def _kernels(rng):
    return [np.ones((5, 5), np.uint8), np.ones((7, 7), np.uint8), np.ones((3, 7), np.uint8),
            np.ones((1, 1), np.uint8), np.ones((4, 4), np.uint8), np.ones((2, 5), np.uint8),
            (rng.random((5, 5)) > 0.5).astype(np.uint8), (rng.random((4, 6)) > 0.4).astype(np.uint8),
            np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8), np.ones((31, 31), np.uint8),
            np.ones((15, 3), np.uint8)]


@pytest.mark.parametrize("shape", [(64, 50), (20, 32), (1, 1), (3, 70), (70, 3), (100, 150),
                                   (33, 129), (64, 64)])
def test_erode_dilate(cv, oracle, shape):
    rng = _rng(5)
    m = rng.choice([0, 127, 255], size=shape).astype(np.uint8)
    g = rng.integers(0, 256, size=shape).astype(np.uint8)   # arbitrary grey levels too
    for k in _kernels(rng):
        for src in (m, g):
            assert np.array_equal(cv.erode(src, k), oracle.erode(src, k)), k.shape
            assert np.array_equal(cv.dilate(src, k), oracle.dilate(src, k)), k.shape


def test_morph_known_answers_and_properties(cv):
    k5 = cv.getStructuringElement(cv.MORPH_RECT, (5, 5))
    full = np.full((1080, 1920), 255, np.uint8)
    assert (cv.erode(full, k5) == 255).all()                    # border never wins
    one = np.zeros((1080, 1920), np.uint8)
    one[500, 900] = 255
    d = cv.dilate(one, k5)
    assert int(d.sum()) == 25 * 255 and (d[498:503, 898:903] == 255).all()
    assert np.array_equal(cv.erode(d, k5), one)
    # idempotence of opening at full size
    m = (_rng(9).random((1080, 1920)) > 0.7).astype(np.uint8) * 255
    o1 = cv.dilate(cv.erode(m, k5), k5)
    o2 = cv.dilate(cv.erode(o1, k5), k5)
    assert np.array_equal(o1, o2)
    assert (o1 <= m).all()                                      # opening is anti-extensive
    with pytest.raises(ValueError):
        cv.erode(np.zeros((4, 4, 3), np.uint8), k5)
    with pytest.raises(ValueError):
        cv.erode(m, np.ones((33, 33), np.uint8))


# ---- step 2b: frame differencing ------------------------------------------

@pytest.mark.parametrize("size", [(64, 50), (50, 64, 1), (32, 28, 3), (7, 5, 4), (300, 500, 3)])
def test_framedifferencing(oracle, size):
    """Reference test (tests/test_segmentation.py:12-24) + value parity."""
    from pygofpid_amd.segmentation import FrameDifferencing, ForegroundEstimator
    np.random.seed(17)
    fd = FrameDifferencing(threshold=7.5)
    ref = oracle.FrameDifferencing(threshold=7.5)
    assert isinstance(fd, ForegroundEstimator) and fd.FOREGROUND == 255 and fd.BACKGROUND == 0
    img = np.random.randint(0, high=255, size=size, dtype=np.uint8)
    out = fd.apply(img)
    ref.apply(img)
    assert np.array_equal(out, np.zeros(size[:2]))
    for i in range(4):
        if i == 2:   # small differences straddling the threshold
            img = (img.astype(np.int16) + np.random.randint(-9, 10, size=size)).clip(0, 255).astype(np.uint8)
        else:
            img = np.random.randint(0, high=255, size=size, dtype=np.uint8)
        out = fd.apply(img)
        assert out.shape == size[:2]
        assert np.array_equal(out, ref.apply(img))
    with pytest.raises(ValueError):
        fd.apply(np.zeros((3, 3), np.uint8))


def test_fd_golden():
    from pygofpid_amd.segmentation import FrameDifferencing
    g = np.load(os.path.join(GOLDEN, "fd_bgr.npz"))
    fd = FrameDifferencing(7.5)
    for f, want in zip(g["frames"], g["masks"]):
        assert np.array_equal(fd.apply(f), want)


# ---- step 2a: MOG2 ---------------------------------------------------------

def _assert_state_equal(got, want):
    assert got["weight"].shape == want["weight"].shape   # (nmixtures, n) on both sides
    live = np.arange(want["weight"].shape[0])[:, None] < want["modes"][None, :]
    assert np.array_equal(got["modes"], want["modes"])
    assert np.array_equal(got["weight"][live], want["weight"][live])
    assert np.array_equal(got["var"][live], want["var"][live])
    for ch in range(want["mean"].shape[1]):
        assert np.array_equal(got["mean"][:, ch][live], want["mean"][:, ch][live])


@pytest.mark.parametrize("shape", [(32, 20), (20, 32, 1), (32, 20, 3), (6, 5, 4), (1, 1, 3),
                                   (67, 131, 3)])
def test_mog2_uniform_noise(cv, oracle, shape):
    """Reference-test style input (tests/test_pipeline.py:98-109): worst case, no
    mode ever fits, 5 modes per pixel."""
    gpu, ref = cv.createBackgroundSubtractorMOG2(), oracle.MOG2()
    for f in uniform_noise_frames(shape, 12, seed=3):
        got, want = gpu.apply(f), ref.apply(f)
        assert got.shape == tuple(shape[:2])
        assert np.array_equal(got, want)
    _assert_state_equal(gpu.export_state(), ref.export())


@pytest.mark.parametrize("c", [1, 3])
def test_mog2_clip_masks_and_state(cv, oracle, c):
    """Static background + noise + moving objects + shadows: fit, prune, sort,
    new-mode and shadow branches; state compared bit for bit after 120 frames."""
    clip = Clip(72, 100, c, seed=17, n_objects=4)
    gpu, ref = cv.createBackgroundSubtractorMOG2(), oracle.MOG2()
    seen = set()
    for t in range(120):
        f = oracle.gaussian_blur(clip.frame(t))
        got, want = gpu.apply(f), ref.apply(f)
        assert np.array_equal(got, want), "frame %d: %d px differ" % (t, (got != want).sum())
        seen |= set(np.unique(got).tolist())
    assert seen == {0, 127, 255}
    st = ref.export()
    assert len(np.unique(st["modes"])) >= 3
    _assert_state_equal(gpu.export_state(), st)


def test_mog2_known_answers(cv):
    x = np.array([[[10, 20, 30], [0, 0, 0], [255, 1, 0]]], np.uint8)
    m = cv.createBackgroundSubtractorMOG2()
    assert m.apply(x).tolist() == [[127, 255, 127]]
    x2 = x.copy()
    x2[0, 2] = [0, 200, 200]
    assert m.apply(x2)[0, 0] == 0
    st = m.export_state()
    assert st["modes"].tolist() == [1, 1, 2] and st["nframes"] == 2
    assert np.allclose(st["weight"][:2, 2], [0.75, 0.25])


def test_mog2_golden_gray():
    from pygofpid_amd import cv
    g = np.load(os.path.join(GOLDEN, "mog2_gray_noise.npz"))
    m = cv.createBackgroundSubtractorMOG2()
    for f, want in zip(g["frames"], g["masks"]):
        assert np.array_equal(m.apply(f), want)


def test_mog2_state_roundtrip(cv, oracle):
    """export -> import into a fresh model -> identical continuation
    (checkpoint / warm start path)."""
    clip = Clip(40, 48, 3, seed=2)
    a = cv.createBackgroundSubtractorMOG2()
    for t in range(20):
        a.apply(clip.frame(t))
    st = a.export_state()
    b = cv.createBackgroundSubtractorMOG2()
    b.import_state(st, (40, 48, 3))
    for t in range(20, 30):
        f = clip.frame(t)
        assert np.array_equal(a.apply(f), b.apply(f))


# ---- steps 1-3 fused -------------------------------------------------------

def _run_pair(gpu, ref, frames):
    for t, f in enumerate(frames):
        got, want = gpu.apply(f), ref.apply(f)
        assert np.array_equal(got, want), "frame %d: %d px differ" % (t, (got != want).sum())


@pytest.mark.parametrize("shape", [(40, 56, 3), (64, 50, 1), (33, 67, 3), (130, 200, 3), (5, 4, 3)])
def test_pipeline_default(oracle, shape):
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = shape
    gpu = ForegroundPipeline(shape)
    ref = oracle.Pipeline(h, w, c)
    _run_pair(gpu, ref, Clip(h, w, c, seed=17, n_objects=3).frames(40))
    assert abs(gpu.mean_modes() - ref.mean_modes()) < 1e-12


def test_pipeline_golden():
    from pygofpid_amd.foreground import ForegroundPipeline
    g = np.load(os.path.join(GOLDEN, "pipeline_mog2_bgr.npz"))
    gpu = ForegroundPipeline(g["frames"].shape[1:])
    for f, want in zip(g["frames"], g["final_mask"]):
        assert np.array_equal(gpu.apply(f), want)
    st = gpu.mog2_state()
    _assert_state_equal(st, {k: g[k] for k in ("weight", "var", "mean", "modes")})


def test_pipeline_open_close_7x7(oracle):
    """BASELINE config 3's operator list: [erode, dilate, dilate, erode] 7x7."""
    from pygofpid_amd.foreground import ForegroundPipeline
    k7 = np.ones((7, 7), np.uint8)
    morph = [("erode", k7), ("dilate", k7), ("dilate", k7), ("erode", k7)]
    gpu = ForegroundPipeline((90, 120, 3), morph=morph)
    ref = oracle.Pipeline(90, 120, 3, morph=(("erode", 7), ("dilate", 7), ("dilate", 7), ("erode", 7)))
    _run_pair(gpu, ref, Clip(90, 120, 3, seed=5).frames(25))


@pytest.mark.parametrize("blur", [None, (5, 5, 4.0), (3, 3, 0.0), (3, 3, 0.05), (3, 3, 1.0)])
def test_pipeline_other_blurs(oracle, blur):
    from pygofpid_amd.foreground import ForegroundPipeline
    gpu = ForegroundPipeline((50, 70, 3), blur=blur)
    ref = oracle.Pipeline(50, 70, 3, blur_k=0 if blur is None else blur[0],
                          blur_sigma=0.0 if blur is None else blur[2])
    _run_pair(gpu, ref, Clip(50, 70, 3, seed=6).frames(20))


def test_pipeline_nonrect_morph(oracle, cv):
    from pygofpid_amd.foreground import ForegroundPipeline
    cross = cv.getStructuringElement(cv.MORPH_CROSS, (5, 5))
    gpu = ForegroundPipeline((50, 70, 3), morph=[("erode", cross), ("dilate", np.ones((4, 4), np.uint8))])
    mog2 = oracle.MOG2()
    for f in Clip(50, 70, 3, seed=7).frames(15):
        want = oracle.dilate(oracle.erode(mog2.apply(oracle.gaussian_blur(f)), cross),
                             np.ones((4, 4), np.uint8))
        assert np.array_equal(gpu.apply(f), want)


def test_pipeline_fd(oracle):
    from pygofpid_amd.foreground import ForegroundPipeline
    gpu = ForegroundPipeline((60, 80, 3), detector="FD", fd_threshold=7.5)
    ref = oracle.Pipeline(60, 80, 3, detector="FD", fd_thr=7.5)
    _run_pair(gpu, ref, Clip(60, 80, 3, seed=8).frames(10))


def test_pipeline_streams_are_independent(oracle):
    """A batch of S streams == S single-stream pipelines (the sharding unit)."""
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, c = 3, 44, 60, 3
    clips = [Clip(h, w, c, seed=20 + s, n_objects=2 + s) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    for t in range(25):
        frames = np.stack([cl.frame(t) for cl in clips])
        got = gpu.apply(frames)
        assert got.shape == (S, h, w)
        for s in range(S):
            assert np.array_equal(got[s], refs[s].apply(frames[s]))


@pytest.mark.parametrize("T", [1, 2, 3, 8, 11])
def test_pipeline_temporal_blocking(oracle, T):
    """apply_many(T frames) == T x apply(): the model stays in registers across
    the frames of one launch (8 per launch; 11 = one full + one partial launch),
    masks AND final model state bit-identical to the frame-by-frame oracle."""
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, c = 2, 46, 70, 3
    clips = [Clip(h, w, c, seed=30 + s, n_objects=3) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    t = 0
    for _ in range(4):
        frames = np.stack([np.stack([cl.frame(t + i) for cl in clips]) for i in range(T)])
        got = gpu.apply_many(frames)
        assert got.shape == (T, S, h, w)
        for i in range(T):
            for s in range(S):
                want = refs[s].apply(frames[i, s])
                assert np.array_equal(got[i, s], want), (t + i, s)
        t += T
    assert abs(gpu.mean_modes() - np.mean([r.mean_modes() for r in refs])) < 1e-12
    st = gpu.mog2_state()
    assert st["nframes"] == t
    n = h * w
    for s in range(S):     # per-stream model state, bit for bit
        got = {"modes": st["modes"][s * n:(s + 1) * n], "weight": st["weight"][:, s * n:(s + 1) * n],
               "var": st["var"][:, s * n:(s + 1) * n], "mean": st["mean"][:, :, s * n:(s + 1) * n]}
        _assert_state_equal(got, refs[s].export())


@pytest.mark.parametrize("shape", [(46, 76, 3), (37, 128, 3), (30, 68, 1), (22, 66, 4), (5, 4, 3)])
def test_multi_frame_kernel_geometries(oracle, shape):
    """The dedicated 8-frame kernel (aligned rows: w*c % 4 == 0) on ragged tiles:
    widths that are not a multiple of the 64-px tile, heights that are not a
    multiple of 4 (lanes outside the image shadow a pixel inside it), 1 / 3 / 4
    channels, full and partial launches.  Masks of every frame and the final model
    state against the frame-by-frame oracle; the scene is quiet enough that whole
    waves sit in the settled state (one mode of weight exactly 1.0f), which is the
    path that takes the host-evaluated weight arithmetic."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = shape
    S = 2
    clips = [Clip(h, w, c, seed=300 + s, n_objects=1) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    t = 0
    for T in (8, 8, 5, 8, 2, 8):
        frames = np.stack([np.stack([cl.frame(t + i) for cl in clips]) for i in range(T)])
        got = gpu.apply_many(frames).reshape(T, S, h, w)
        for i in range(T):
            for s in range(S):
                want = refs[s].apply(frames[i, s])
                assert np.array_equal(got[i, s], want), (shape, t + i, s)
        t += T
    st = gpu.mog2_state()
    n = h * w
    for s in range(S):
        got = {"modes": st["modes"][s * n:(s + 1) * n], "weight": st["weight"][:, s * n:(s + 1) * n],
               "var": st["var"][:, s * n:(s + 1) * n], "mean": st["mean"][:, :, s * n:(s + 1) * n]}
        _assert_state_equal(got, refs[s].export())
    settled = (st["modes"] == 1) & (st["weight"][0] == np.float32(1.0))
    if h * w > 1000:      # (the 5 x 4 image is mostly covered by its moving object)
        assert settled.mean() > 0.3, settled.mean()


@pytest.mark.parametrize("over", [dict(history=6, ct=8.0), dict(history=40, ct=0.0),
                                  dict(history=3, background_ratio=0.0, var_threshold_gen=30.0),
                                  dict(nmixtures=1, history=10), dict(nmixtures=2, detect_shadows=0)])
def test_multi_frame_kernel_other_rates(oracle, over):
    """Learning-rate terms away from the defaults, through 8-frame launches: rates at
    which a lone mode of weight 1 would be pruned (the shortcut must switch itself
    off for those frames), a zero complexity-reduction term, background ratio 0, a
    single-mode mixture.  Oracle: its own blur -> MOG2 -> erode -> dilate."""
    from pygofpid_amd import _lib
    from pygofpid_amd.foreground import ForegroundPipeline
    import ctypes as C
    h, w, c = 33, 72, 3
    p = _lib.Mog2Params()
    _lib.load().gof_mog2_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    gpu = ForegroundPipeline((h, w, c), mog2_params=p)
    ref = oracle.MOG2(**over)
    k5 = np.ones((5, 5), np.uint8)
    clip = Clip(h, w, c, seed=91, n_objects=2)
    for r in range(5):
        frames = np.stack(clip.frames(8, start=8 * r))
        got = gpu.apply_many(frames)
        for i in range(8):
            want = oracle.dilate(oracle.erode(ref.apply(oracle.gaussian_blur(frames[i])), k5), k5)
            assert np.array_equal(got[i], want), (over, 8 * r + i)
    want = ref.export()
    _assert_state_equal(gpu.mog2_state(), want)


@pytest.mark.parametrize("T", [8, 9, 11])
def test_morphology_skips_only_empty_regions(oracle, T):
    """The multi-frame MOG2 launch tells the morphology which 64-px row segments of
    its masks are all zero, and strips whose whole input window is flagged empty are
    written as zeros without being read.  Wide, mostly empty frames (several strips
    per row, small objects that cross strip and tile boundaries), full launches, a
    partial second launch (T = 11) and a one-frame tail that the per-frame kernel
    handles and that must switch the skipping off for its chunk (T = 9)."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c, S = 100, 520, 3, 2
    clips = [Clip(h, w, c, seed=500 + s, n_objects=2, speed=5) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    t = 0
    empty = 0
    for _ in range(3):
        frames = np.stack([np.stack([cl.frame(t + i) for cl in clips]) for i in range(T)])
        got = gpu.apply_many(frames).reshape(T, S, h, w)
        for i in range(T):
            for s in range(S):
                want = refs[s].apply(frames[i, s])
                assert np.array_equal(got[i, s], want), (T, t + i, s, int((got[i, s] != want).sum()))
                empty += int(not want[:, :248].any()) + int(not want[:, 248:].any())
        t += T
    assert empty > 0   # the scene did contain empty strips-worth of mask


@pytest.mark.parametrize("shape", [(46, 76, 3), (30, 68, 1), (22, 66, 4), (37, 130, 3)])
def test_fd_fused_kernel(oracle, shape):
    """GaussianBlur + FrameDifferencing as one multi-frame kernel (aligned frames: the
    first three shapes) and as separate kernels (130 * 3 is not a multiple of 4):
    one-frame calls, full and partial launches, two streams; the first mask ever is
    all zero [REF tests/test_segmentation.py:19], every later one equals the oracle's
    blur -> FD -> erode -> dilate."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = shape
    S = 2
    clips = [Clip(h, w, c, seed=700 + s, n_objects=2, speed=4) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S, detector="FD", fd_threshold=6)
    refs = [oracle.Pipeline(h, w, c, detector="FD", fd_thr=6) for _ in range(S)]
    t = 0
    for T in (1, 1, 8, 3, 11, 1):
        frames = np.stack([np.stack([cl.frame(t + i) for cl in clips]) for i in range(T)])
        got = gpu.apply_many(frames).reshape(T, S, h, w)
        for i in range(T):
            for s in range(S):
                want = refs[s].apply(frames[i, s])
                if t + i == 0:
                    assert not want.any()
                assert np.array_equal(got[i, s], want), (shape, t + i, s, int((got[i, s] != want).sum()))
        t += T


def test_pipeline_temporal_blocking_other_paths(oracle):
    """apply_many through the unfused blur, the FD detector and a grey stream."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w = 40, 52
    for kind in ("blur5", "fd", "gray"):
        c = 1 if kind == "gray" else 3
        clip = Clip(h, w, c, seed=40, n_objects=2)
        if kind == "blur5":
            gpu = ForegroundPipeline((h, w, c), blur=(5, 5, 4.0))
            ref = oracle.Pipeline(h, w, c, blur_k=5, blur_sigma=4.0)
        elif kind == "fd":
            gpu = ForegroundPipeline((h, w, c), detector="FD", fd_threshold=6)
            ref = oracle.Pipeline(h, w, c, detector="FD", fd_thr=6)
        else:
            gpu = ForegroundPipeline((h, w, c))
            ref = oracle.Pipeline(h, w, c)
        for r in range(3):
            frames = np.stack(clip.frames(5, start=5 * r))
            got = gpu.apply_many(frames)
            for i in range(5):
                assert np.array_equal(got[i], ref.apply(frames[i])), (kind, r, i)


def test_pipeline_long_run_steady_state(oracle):
    """640 frames: past frame 250 the learning rate sits at 1/history, modes
    created by passing objects decay and are pruned (the `nmodes--` path), the
    weight sort keeps re-ordering.  Masks every frame and the final model state
    must still be bit-identical, through 8-frame launches."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 36, 64, 3
    clip = Clip(h, w, c, seed=77, n_objects=3, speed=2)
    gpu = ForegroundPipeline((h, w, c))
    ref = oracle.Pipeline(h, w, c)
    peak = 0
    for t in range(0, 640, 8):
        frames = np.stack(clip.frames(8, start=t))
        got = gpu.apply_many(frames)
        for i in range(8):
            assert np.array_equal(got[i], ref.apply(frames[i])), t + i
        if t % 160 == 0:
            peak = max(peak, int(gpu.mog2_state()["modes"].max()))
    st = gpu.mog2_state()
    assert st["nframes"] == 640 and peak >= 3
    _assert_state_equal(st, ref.export())
    # pruning happened: some pixel holds fewer modes now than the peak
    assert st["modes"].min() < peak


def test_pipeline_checkpoint_restore(tmp_path):
    """save_state -> load_state into a fresh pipeline -> identical continuation
    (fail-over of a long-running stream)."""
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, c = 2, 40, 56, 3
    clips = [Clip(h, w, c, seed=50 + s) for s in range(S)]
    a = ForegroundPipeline((h, w, c), streams=S)
    for t in range(0, 24, 8):
        a.apply_many(np.stack([np.stack([cl.frame(t + i) for cl in clips]) for i in range(8)]))
    path = str(tmp_path / "model.npz")
    a.save_state(path)
    b = ForegroundPipeline((h, w, c), streams=S)
    b.load_state(path)
    for t in range(24, 34):
        f = np.stack([cl.frame(t) for cl in clips])
        assert np.array_equal(a.apply(f), b.apply(f))
    with pytest.raises(ValueError):
        ForegroundPipeline((h, w + 8, c), streams=S).load_state(path)


def test_pipeline_full_hd_bit_exact(oracle):
    """BASELINE config 2 geometry, a few frames against the oracle."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 1080, 1920, 3
    gpu = ForegroundPipeline((h, w, c))
    ref = oracle.Pipeline(h, w, c)
    _run_pair(gpu, ref, Clip(h, w, c, seed=17).frames(6))


def test_pipeline_4k_open_close_bit_exact(oracle):
    """BASELINE config 3 geometry."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 2160, 3840, 3
    k7 = np.ones((7, 7), np.uint8)
    gpu = ForegroundPipeline((h, w, c), morph=[("erode", k7), ("dilate", k7), ("dilate", k7), ("erode", k7)])
    ref = oracle.Pipeline(h, w, c, morph=(("erode", 7), ("dilate", 7), ("dilate", 7), ("erode", 7)))
    _run_pair(gpu, ref, Clip(h, w, c, seed=3).frames(3))


def test_pipeline_full_hd_multi_frame(oracle):
    """The benchmark's own form at its own geometry: 1080p, 8 frames per launch,
    two streams, three launches; every mask and the final model state against the
    frame-by-frame oracle (multi-frame kernel, row flags, skipping morphology)."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c, S, T = 1080, 1920, 3, 2, 8
    clips = [Clip(h, w, c, seed=40 + s, n_objects=4) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    for r in range(3):
        frames = np.stack([np.stack([cl.frame(T * r + i) for cl in clips]) for i in range(T)])
        got = gpu.apply_many(frames)
        for i in range(T):
            for s in range(S):
                want = refs[s].apply(frames[i, s])
                assert np.array_equal(got[i, s], want), (T * r + i, s, int((got[i, s] != want).sum()))
    st = gpu.mog2_state()
    n = h * w
    for s in range(S):
        got = {"modes": st["modes"][s * n:(s + 1) * n], "weight": st["weight"][:, s * n:(s + 1) * n],
               "var": st["var"][:, s * n:(s + 1) * n], "mean": st["mean"][:, :, s * n:(s + 1) * n]}
        _assert_state_equal(got, refs[s].export())


def test_pipeline_sharding_unit_8x1080p(oracle):
    """The per-GPU share of BASELINE config 4 and the unit bench.py times: EIGHT 1080p streams
    through 8-frame launches (two of them: the learning-rate ramp, new modes, trails), every mask
    of every stream and the final float32 model state of every stream against the oracle."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c, S, T = 1080, 1920, 3, 8, 8
    clips = [Clip(h, w, c, seed=70 + s, n_objects=3 + s % 3) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    for r in range(2):
        frames = np.stack([np.stack([cl.frame(T * r + i) for cl in clips]) for i in range(T)])
        got = gpu.apply_many(frames)
        assert got.shape == (T, S, h, w)
        for i in range(T):
            for s in range(S):
                want = refs[s].apply(frames[i, s])
                assert np.array_equal(got[i, s], want), (T * r + i, s, int((got[i, s] != want).sum()))
        del frames, got
    st = gpu.mog2_state()
    n = h * w
    for s in range(S):
        got = {"modes": st["modes"][s * n:(s + 1) * n], "weight": st["weight"][:, s * n:(s + 1) * n],
               "var": st["var"][:, s * n:(s + 1) * n], "mean": st["mean"][:, :, s * n:(s + 1) * n]}
        _assert_state_equal(got, refs[s].export())


def test_pipeline_4k_open_close_multi_frame(oracle):
    """BASELINE config 3 geometry through an 8-frame launch and the 7x7 open / close chain."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 2160, 3840, 3
    k7 = np.ones((7, 7), np.uint8)
    gpu = ForegroundPipeline((h, w, c), morph=[("erode", k7), ("dilate", k7), ("dilate", k7), ("erode", k7)])
    ref = oracle.Pipeline(h, w, c, morph=(("erode", 7), ("dilate", 7), ("dilate", 7), ("erode", 7)))
    frames = np.stack(Clip(h, w, c, seed=5, n_objects=3).frames(8))
    got = gpu.apply_many(frames)
    for i in range(8):
        assert np.array_equal(got[i], ref.apply(frames[i])), i


def test_pipeline_errors():
    from pygofpid_amd.foreground import ForegroundPipeline
    with pytest.raises(ValueError):
        ForegroundPipeline((10, 10, 3), detector="blabla")      # tests/test_pipeline.py:112-118
    with pytest.raises(ValueError):
        ForegroundPipeline((10, 10, 2))
    with pytest.raises(ValueError):
        ForegroundPipeline((10, 10, 3), blur=(4, 4, 1.0))
    p = ForegroundPipeline((10, 10, 3))
    with pytest.raises(ValueError):
        p.apply(np.zeros((12, 10, 3), np.uint8))                # tests/test_pipeline.py:31-41


def test_c_abi_error_codes():
    """The C ABI itself: bad arguments come back as GOF_EINVAL with a message, never a crash."""
    import ctypes as C
    from pygofpid_amd import _lib
    lib = _lib.load()
    ctx = _lib.default_context()
    buf = np.zeros((8, 8, 3), np.uint8)
    out = np.zeros((8, 8, 3), np.uint8)
    p8 = _lib.p8
    assert lib.gof_gaussian_blur_host(ctx.handle, None, p8(out), 8, 8, 3, 3, 3, 4.0, 0.0) == _lib.EINVAL
    assert lib.gof_gaussian_blur_host(ctx.handle, p8(buf), p8(out), 0, 8, 3, 3, 3, 4.0, 0.0) == _lib.EINVAL
    assert lib.gof_gaussian_blur_host(ctx.handle, p8(buf), p8(out), 8, 8, 2, 3, 3, 4.0, 0.0) == _lib.EINVAL
    assert lib.gof_gaussian_blur_host(ctx.handle, p8(buf), p8(out), 8, 8, 3, 33, 3, 4.0, 0.0) == _lib.EINVAL
    assert b"ksize" in lib.gof_last_error()
    assert lib.gof_morph_host(ctx.handle, 7, p8(buf), p8(out), 8, 8, None, 3, 3) == _lib.EINVAL
    assert lib.gof_morph_host(ctx.handle, 0, p8(buf), p8(out), 8, 8, None, 0, 3) == _lib.EINVAL
    h = C.c_void_p()
    assert lib.gof_mog2_create(ctx.handle, 1, 8, 8, 5, None, C.byref(h)) == _lib.EINVAL and not h
    p = _lib.Mog2Params()
    lib.gof_mog2_default_params(C.byref(p))
    p.nmixtures = 6
    assert lib.gof_mog2_create(ctx.handle, 1, 8, 8, 3, C.byref(p), C.byref(h)) == _lib.EINVAL
    bad = C.c_void_p()
    assert lib.gof_create(99, C.byref(bad)) == _lib.EINVAL and not bad
    # a pipeline: n_frames out of range, NULL buffers
    from pygofpid_amd.foreground import ForegroundPipeline
    pipe = ForegroundPipeline((8, 8, 3))
    assert lib.gof_pipeline_apply_host_many(pipe._h, p8(buf), p8(out), 0) == _lib.EINVAL
    assert lib.gof_pipeline_apply_host(pipe._h, None, p8(out)) == _lib.EINVAL
    assert lib.gof_pipeline_apply_dev_many(pipe._h, None, None, 1, None) == _lib.EINVAL
    # in-place device calls are refused
    import torch
    t = torch.zeros(64 * 3, dtype=torch.uint8, device="cuda")
    assert lib.gof_gaussian_blur_dev(ctx.handle, t.data_ptr(), t.data_ptr(), 1, 8, 8, 3, 3, 3, 4.0, 0.0,
                                     None) == _lib.EINVAL
    assert lib.gof_morph_dev(ctx.handle, 0, t.data_ptr(), t.data_ptr(), 1, 8, 8, None, 3, 3,
                             None) == _lib.EINVAL
    assert lib.gof_strerror(_lib.EHIP) == b"HIP runtime error"
    # round 3: KNN and the asynchronous slots
    k = C.c_void_p()
    assert lib.gof_knn_create(ctx.handle, 1, 8, 8, 2, None, C.byref(k)) == _lib.EINVAL and not k
    kp = _lib.KnnParams()
    lib.gof_knn_default_params(C.byref(kp))
    kp.nsamples = 17
    assert lib.gof_knn_create(ctx.handle, 1, 8, 8, 3, C.byref(kp), C.byref(k)) == _lib.EINVAL and not k
    assert b"KNN" in lib.gof_last_error()
    assert lib.gof_knn_create(ctx.handle, 1, 8, 8, 3, None, C.byref(k)) == _lib.OK and k
    m8 = np.zeros((8, 8), np.uint8)
    assert lib.gof_knn_apply_host(k, None, p8(m8)) == _lib.EINVAL
    assert lib.gof_knn_apply_dev(k, None, None, None) == _lib.EINVAL
    assert lib.gof_knn_export(k, None, None) == _lib.EINVAL
    assert lib.gof_knn_apply_host(k, p8(buf), p8(m8)) == _lib.OK and lib.gof_knn_nframes(k) == 1
    lib.gof_knn_destroy(k)
    assert lib.gof_pipeline_knn(pipe._h) is None            # a MOG2 pipeline has no KNN model
    from pygofpid_amd.blobs import BlobExtractor
    ex = BlobExtractor(1, 8, 8, ctx=pipe._ctx)
    ptrs = (C.c_void_p * 1)(buf.ctypes.data)
    assert lib.gof_pipeline_submit_host_blobs_ptrs(pipe._h, ptrs, 1, ex.handle, 100.0, 2) == _lib.EINVAL   # slot
    assert lib.gof_pipeline_submit_host_blobs_ptrs(pipe._h, None, 1, ex.handle, 100.0, 0) == _lib.EINVAL
    assert lib.gof_pipeline_submit_host_blobs_ptrs(pipe._h, ptrs, 2, ex.handle, 100.0, 0) == _lib.EINVAL   # scratch of 1 image
    null = (C.c_void_p * 1)(None)
    assert lib.gof_pipeline_submit_host_blobs_ptrs(pipe._h, null, 1, ex.handle, 100.0, 0) == _lib.EINVAL
    assert lib.gof_pipeline_wait_slot(pipe._h, 3) == _lib.EINVAL
    assert lib.gof_pipeline_wait_slot(pipe._h, 1) == _lib.OK                    # nothing in flight there
    assert lib.gof_pipeline_fetch_masks_slot(pipe._h, p8(m8), 1, 1) == _lib.EINVAL   # no masks of slot 1 on the device
    assert lib.gof_pipeline_submit_host_blobs_ptrs(pipe._h, ptrs, 1, ex.handle, 100.0, 0) == _lib.OK
    assert lib.gof_pipeline_wait_slot(pipe._h, 0) == _lib.OK
    assert lib.gof_pipeline_fetch_masks_slot(pipe._h, p8(m8), 1, 0) == _lib.OK


# ---- KNN (cv.createBackgroundSubtractorKNN, reference pipeline.py:233-234) ------------------

def _assert_knn_state_equal(got, want):
    assert got["samples"].shape == want["samples"].shape and got["state"].shape == want["state"].shape
    assert np.array_equal(got["state"], want["state"])          # refresh slots and write indices
    assert np.array_equal(got["samples"], want["samples"])      # every sample byte and flag


@pytest.mark.parametrize("shape", [(32, 20), (20, 32, 1), (32, 20, 3), (6, 5, 4), (1, 1, 3), (67, 131, 3)])
def test_knn_uniform_noise(cv, oracle, shape):
    """The reference's own KNN case (tests/test_pipeline.py:98-109: uniform noise, 5 frames; more here):
    masks and the whole sample model -- bytes, flags, slots, indices -- equal the CPU restatement.
    (Parity with cv2 itself is unpinnable: its slots come from rand() in a row-parallel loop.)"""
    gpu, ref = cv.createBackgroundSubtractorKNN(), oracle.KNN()
    for f in uniform_noise_frames(shape, 14, seed=5):
        got, want = gpu.apply(f), ref.apply(f)
        assert got.shape == tuple(shape[:2]) and np.array_equal(got, want)
    _assert_knn_state_equal(gpu.export_state(), ref.export())


@pytest.mark.parametrize("nsamples,knn,shadows", [(5, 1, True), (3, 2, True), (7, 1, False), (9, 3, True), (7, 2, False), (16, 2, True)])
def test_knn_other_parameters(cv, oracle, nsamples, knn, shadows):
    """Sample counts and kNN other than cv2's 3 x 7 / 2 take the kernels' run-time plane loop (the planes
    are not kept in registers there), kNN = 1 the scan without the two planes requested up front: masks and
    models equal the oracle on a clip with objects and shadows followed by uniform noise, colour and grey."""
    for c in (3, 1):
        h, w = 40, 72
        clip = Clip(h, w, c, seed=21, n_objects=2)
        gpu = cv.createBackgroundSubtractorKNN(detectShadows=shadows)
        gpu._params.nsamples, gpu._params.knn = nsamples, knn
        ref = oracle.KNN(nsamples=nsamples, knn=knn, detect_shadows=int(shadows))
        frames = [clip.frame(t) for t in range(40)] + uniform_noise_frames((h, w, c) if c > 1 else (h, w), 12, seed=3)
        for f in frames:
            assert np.array_equal(gpu.apply(f), ref.apply(f))
        _assert_knn_state_equal(gpu.export_state(), ref.export())


@pytest.mark.parametrize("c", [1, 3])
def test_knn_clip_masks_and_state(cv, oracle, c):
    """Static background + noise + moving objects + shadow patches through the learning-rate ramp
    into long cycles (slots drawn, all three sets refreshed at their own times): background, shadow
    and foreground all occur; masks and model equal the oracle frame by frame."""
    h, w = 72, 100
    clip = Clip(h, w, c, seed=8, n_objects=3)
    gpu, ref = cv.createBackgroundSubtractorKNN(), oracle.KNN()
    seen = set()
    for t in range(120):
        f = clip.frame(t)
        got, want = gpu.apply(f), ref.apply(f)
        assert np.array_equal(got, want), (t, int((got != want).sum()))
        seen |= set(np.unique(got).tolist())
        if t in (3, 40, 119):
            _assert_knn_state_equal(gpu.export_state(), ref.export())
    assert seen == {0, 127, 255}


@pytest.mark.parametrize("geom", [(60, 88, 3), (45, 75, 3), (50, 132, 1), (33, 70, 4)])
def test_knn_pipeline_streams(oracle, geom):
    """KNN inside the pipeline (blur -> KNN -> erode -> dilate), two streams, several frames per call:
    aligned rows take the kernel with the 3x3 blur inside (ragged tiles included), the others the
    separate blur kernel; masks equal the oracle pipeline, the model equals per stream."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = geom
    S = 2
    clips = [Clip(h, w, c, seed=20 + s, n_objects=2) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S, detector="KNN")
    refs = [oracle.Pipeline(h, w, c, detector="KNN") for _ in range(S)]
    for r in range(4):
        frames = np.stack([np.stack([cl.frame(5 * r + i) for cl in clips]) for i in range(5)])
        got = gpu.apply_many(frames).reshape(5, S, h, w)
        for i in range(5):
            for s in range(S):
                assert np.array_equal(got[i, s], refs[s].apply(frames[i, s])), (r, i, s)
    st = gpu.knn_state()
    n = h * w
    for s in range(S):
        want = refs[s].export()
        assert np.array_equal(st["samples"][:, :, s * n:(s + 1) * n], want["samples"])
        assert np.array_equal(st["state"][:, s * n:(s + 1) * n], want["state"])


def test_knn_pipeline_streams_and_1080p(oracle):
    """... and with the box filter, and one 1080p stream."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c, S = 60, 88, 3, 2
    gb, rb = ForegroundPipeline((h, w, c), detector="KNN", blur=("box", 3, 3)), oracle.Pipeline(h, w, c, detector="KNN", blur_k=-3)
    for t in range(6):
        f = Clip(h, w, c, seed=4).frame(t)
        assert np.array_equal(gb.apply(f), rb.apply(f)), t
    clips = [Clip(h, w, c, seed=20 + s, n_objects=2) for s in range(S)]
    gpu = ForegroundPipeline((h, w, c), streams=S, detector="KNN")
    refs = [oracle.Pipeline(h, w, c, detector="KNN") for _ in range(S)]
    for r in range(4):
        frames = np.stack([np.stack([cl.frame(5 * r + i) for cl in clips]) for i in range(5)])
        got = gpu.apply_many(frames)
        for i in range(5):
            for s in range(S):
                assert np.array_equal(got[i, s], refs[s].apply(frames[i, s])), (r, i, s)
    st = gpu.knn_state()
    n = h * w
    for s in range(S):
        want = refs[s].export()
        assert np.array_equal(st["samples"][:, :, s * n:(s + 1) * n], want["samples"])
        assert np.array_equal(st["state"][:, s * n:(s + 1) * n], want["state"])
    h, w = 1080, 1920
    clip = Clip(h, w, 3, seed=2, n_objects=4)
    gpu, ref = ForegroundPipeline((h, w, 3), detector="KNN"), oracle.Pipeline(h, w, 3, detector="KNN")
    for t in range(6):
        f = clip.frame(t)
        assert np.array_equal(gpu.apply(f), ref.apply(f)), t
