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
