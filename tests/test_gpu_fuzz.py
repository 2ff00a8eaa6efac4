"""Seeded randomised sweep of the fused path against the oracle: random geometry
(ragged sizes, 1 / 3 channels), streams, frames per call, blur kind, detector and
morphology list.  Every mask of every frame must be bit-identical."""
import numpy as np
import pytest

from pygofpid_amd.synth import Clip

pytestmark = pytest.mark.gpu


def _configs(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        h = int(rng.integers(1, 150))
        w = int(rng.integers(1, 300))
        if rng.random() < 0.3:
            w = int(rng.integers(1, 20)) * 4          # dword-aligned rows (fast paths)
        c = int(rng.choice([1, 3, 3, 4]))
        streams = int(rng.choice([1, 1, 2, 3]))
        T = int(rng.choice([1, 2, 5, 8, 9]))
        blur = [None, ("gauss", 3, 4.0), ("gauss", 3, 0.0), ("gauss", 5, 4.0), ("box", 3)][int(rng.integers(0, 5))]
        det = "MOG2" if rng.random() < 0.75 else "FD"
        nm = int(rng.integers(0, 4))
        morph = [(str(rng.choice(["erode", "dilate"])), int(rng.choice([3, 5, 7, 9, 11]))) for _ in range(nm)]
        yield i, h, w, c, streams, T, blur, det, morph


@pytest.mark.parametrize("cfg", list(_configs(120, seed=20240)), ids=lambda c: "cfg%d" % c[0])
def test_fuzz_fused_path(oracle, cfg):
    from pygofpid_amd.foreground import ForegroundPipeline
    i, h, w, c, S, T, blur, det, morph = cfg
    if blur is None:
        gblur, ok, osig = None, 0, 0.0
    elif blur[0] == "box":
        gblur, ok, osig = ("box", blur[1], blur[1]), -blur[1], 0.0
    else:
        gblur, ok, osig = (blur[1], blur[1], blur[2]), blur[1], blur[2]
    gpu = ForegroundPipeline((h, w, c), streams=S, blur=gblur, detector=det, fd_threshold=6.5,
                             morph=[(op, np.ones((k, k), np.uint8)) for op, k in morph])
    refs = [oracle.Pipeline(h, w, c, blur_k=ok, blur_sigma=osig, detector=det, fd_thr=6.5,
                            morph=tuple(morph)) for _ in range(S)]
    clips = [Clip(h, w, c, seed=1000 + 10 * i + s, n_objects=2) for s in range(S)]
    t = 0
    for _ in range(3):
        frames = np.stack([np.stack([cl.frame(t + k) for cl in clips]) for k in range(T)])
        got = gpu.apply_many(frames).reshape(T, S, h, w)
        for k in range(T):
            for s in range(S):
                want = refs[s].apply(frames[k, s])
                assert np.array_equal(got[k, s], want), (cfg, t + k, s, int((got[k, s] != want).sum()))
        t += T


def _aligned_configs(n, seed):
    """Dword-aligned rows (the multi-frame MOG2 / FD kernels and the morphology's
    zero-region skipping), more frames per call, several launch chunks."""
    rng = np.random.default_rng(seed)
    for i in range(n):
        c = int(rng.choice([1, 3, 3, 4]))
        w = int(rng.integers(1, 90)) * 4
        h = int(rng.integers(1, 120))
        streams = int(rng.choice([1, 2, 3]))
        T = int(rng.choice([2, 3, 7, 8, 9, 12, 17]))
        blur = ("gauss", 3, float(rng.choice([4.0, 0.0, 1.2])))
        det = "MOG2" if rng.random() < 0.6 else "FD"
        nm = int(rng.integers(0, 3))
        morph = [(str(rng.choice(["erode", "dilate"])), int(rng.choice([3, 5, 7]))) for _ in range(nm)]
        yield 1000 + i, h, w, c, streams, T, blur, det, morph


@pytest.mark.parametrize("cfg", list(_aligned_configs(80, seed=777)), ids=lambda c: "cfg%d" % c[0])
def test_fuzz_multi_frame_kernels(oracle, cfg):
    test_fuzz_fused_path(oracle, cfg)


def _round3_configs(n, seed):
    """The round-3 paths: KNN (with and without the blur inside the kernel), step 0 inside the MOG2 / FD
    kernels or as gray_quad_kernel / gray_kernel (3 and 4 source channels, both orders), the 3x3 box filter
    inside the detectors -- aligned and ragged geometries, one and several frames per call."""
    rng = np.random.default_rng(seed)
    for i in range(n):
        h = int(rng.integers(2, 110))
        w = int(rng.integers(2, 200)) if rng.random() < 0.4 else int(rng.integers(1, 60)) * 4
        convert = [None, "BGR2GRAY", "RGB2GRAY"][int(rng.integers(0, 3))]
        c = int(rng.choice([3, 4])) if convert else int(rng.choice([1, 3, 4]))
        streams = int(rng.choice([1, 2, 3]))
        T = int(rng.choice([1, 2, 8, 9]))
        blur = [None, ("gauss", 3, 4.0), ("box", 3), ("box", 3), ("gauss", 5, 1.5)][int(rng.integers(0, 5))]
        det = str(rng.choice(["MOG2", "FD", "KNN", "KNN"]))
        nm = int(rng.integers(0, 3))
        morph = [(str(rng.choice(["erode", "dilate"])), int(rng.choice([3, 5]))) for _ in range(nm)]
        yield 2000 + i, h, w, c, streams, T, blur, det, morph, convert


@pytest.mark.parametrize("cfg", list(_round3_configs(90, seed=31337)), ids=lambda c: "cfg%d" % c[0])
def test_fuzz_round3_paths(oracle, cfg):
    from pygofpid_amd.foreground import ForegroundPipeline
    i, h, w, c, S, T, blur, det, morph, convert = cfg
    if blur is None:
        gblur, ok, osig = None, 0, 0.0
    elif blur[0] == "box":
        gblur, ok, osig = ("box", blur[1], blur[1]), -blur[1], 0.0
    else:
        gblur, ok, osig = (blur[1], blur[1], blur[2]), blur[1], blur[2]
    gpu = ForegroundPipeline((h, w, c), streams=S, blur=gblur, detector=det, fd_threshold=6.5, convert=convert,
                             morph=[(op, np.ones((k, k), np.uint8)) for op, k in morph])
    mc = 1 if convert else c
    refs = [oracle.Pipeline(h, w, mc, blur_k=ok, blur_sigma=osig, detector=det, fd_thr=6.5,
                            morph=tuple(morph)) for _ in range(S)]
    clips = [Clip(h, w, min(c, 3), seed=5000 + 10 * i + s, n_objects=2) for s in range(S)]
    rng = np.random.default_rng(i)

    def frame(cl, t):
        f = cl.frame(t)
        if f.ndim == 2:
            f = f[:, :, None]
        if c == 4:
            f = np.concatenate([f, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
        return f.reshape(h, w, c) if c > 1 else f.reshape(h, w)

    def seen(f):     # what the detector's model sees
        if not convert:
            return f
        return oracle.cvt_gray(np.ascontiguousarray(f[:, :, :3]), rgb_order=(convert == "RGB2GRAY"))

    t = 0
    for _ in range(3):
        frames = np.stack([np.stack([frame(cl, t + k) for cl in clips]) for k in range(T)])
        got = gpu.apply_many(frames).reshape(T, S, h, w)
        for k in range(T):
            for s in range(S):
                want = refs[s].apply(seen(frames[k, s]))
                assert np.array_equal(got[k, s], want), (cfg, t + k, s, int((got[k, s] != want).sum()))
        t += T
