"""Synthetic surveillance clips (no video or dataset is available offline).

Follows the recipe fixed in BASELINE.md section 3: seed 17 (the seed of the
reference's own tests, tests/test_pipeline.py:7), a static gradient + texture
background, per-frame N(0, 2) sensor noise, bright rectangles (+60) moving
3 px/frame, each trailed by a 0.7x darkened "shadow" patch so that MOG2's
shadow branch (mask value 127) is exercised.  Also the reference-style worst
case: i.i.d. uniform uint8 noise (np.random.randint(0, 255), as in
tests/test_pipeline.py:36).
"""
import numpy as np


class Clip:
    """Deterministic frame generator; frame(t) depends only on (seed, t)."""

    def __init__(self, h, w, c=3, seed=17, n_objects=6, noise_sigma=2.0, speed=3):
        self.h, self.w, self.c = h, w, c
        self.seed = seed
        self.noise_sigma = noise_sigma
        self.speed = speed
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:h, 0:w]
        grad = 60.0 + 100.0 * (xx / max(1, w - 1)) + 40.0 * (yy / max(1, h - 1))
        tex = rng.integers(-12, 13, size=(h, w)).astype(np.float64)
        bg = grad + tex
        if c > 1:
            tint = np.array([1.0, 0.92, 0.85, 1.0][:c])
            bg = bg[..., None] * tint
        self.bg = bg
        oh, ow = max(4, h // 6), max(3, w // 16)
        self.objs = []
        for _ in range(n_objects):
            self.objs.append(dict(
                x0=int(rng.integers(0, max(1, w))), y0=int(rng.integers(0, max(1, h - oh))),
                h=int(oh * rng.uniform(0.7, 1.3)), w=int(ow * rng.uniform(0.7, 1.3)),
                dir=int(rng.choice([-1, 1]))))

    def frame(self, t):
        rng = np.random.default_rng([self.seed, t])
        f = self.bg.copy()
        for o in self.objs:
            x = (o["x0"] + o["dir"] * self.speed * t) % self.w
            xs = slice(x, min(self.w, x + o["w"]))
            ys = slice(o["y0"], min(self.h, o["y0"] + o["h"]))
            # shadow patch beside the object, then the object itself
            sx = slice(min(self.w, x + o["w"]), min(self.w, x + 2 * o["w"]))
            f[ys, sx] = f[ys, sx] * 0.7
            f[ys, xs] = f[ys, xs] + 60.0
        f = f + rng.normal(0.0, self.noise_sigma, size=f.shape)
        return np.clip(np.rint(f), 0, 255).astype(np.uint8)

    def frames(self, n, start=0):
        return [self.frame(t) for t in range(start, start + n)]


def uniform_noise_frames(shape, n, seed=17):
    """Reference-test style input: randint(0, high=255) uint8 (255 exclusive)."""
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 255, size=shape, dtype=np.uint8) for _ in range(n)]
