#!/usr/bin/env python3
"""KNN pipeline (GaussianBlur 3x3 -> KNN -> erode / dilate 5x5) on device-resident frames: step time and
the detector kernel's time, 8 x 1080p streams, benchmark scene and uniform noise; algorithmic bytes of the
KNN kernel = frame in + mask out + 4 B per sample plane examined + 8 B of slots / indices (+ refresh writes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline

S, h, w, P = 8, 1080, 1920, 64
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
masks = torch.empty((1, S, h, w), dtype=torch.uint8, device=dev)
for name, frames in (("scene", bench.make_pool(torch, S, P, h, w, dev, seed=17)),
                     ("uniform noise", bench.make_noise_pool(torch, S, P, h, w, dev, 99))):
    pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, detector="KNN")
    for i in range(1200):
        pipe.apply_device(frames[i % P].data_ptr(), masks.data_ptr())
    pipe.synchronize()
    n = 400
    pipe.profile(True, n)
    t0 = time.perf_counter()
    for i in range(n):
        pipe.apply_device(frames[i % P].data_ptr(), masks.data_ptr())
    pipe.synchronize()
    dt = (time.perf_counter() - t0) / n
    a, b, k = pipe.profile_query()
    npx = S * h * w
    fg = float((masks != 0).float().mean().item())
    print("%-14s step %.1f us (%.0f frames/s)   blur + KNN kernels %.1f us   morphology %.1f us   foreground %.3f"
          % (name, dt * 1e6, S / dt, 1e3 * a / k, 1e3 * b / k, fg))
    print("               bounds on the KNN kernel's bytes per step: %.2f GB (3 planes examined per pixel) .. %.2f GB (all 21)"
          % (npx * (3 + 1 + 3 * 4 + 8) / 1e9, npx * (3 + 1 + 21 * 4 + 8) / 1e9))
    del pipe
