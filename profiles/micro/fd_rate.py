"""FrameDifferencing pipeline (blur -> FD -> erode/dilate) on the bench geometry: kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w = 8, 32, 1080, 1920
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17)
masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx = _lib.Context(0)
for T in (1, 8):
    pipe = ForegroundPipeline((h, w, 3), streams=S, detector="FD", ctx=ctx)
    done = 0
    for i in range(64 + 128):
        if i == 64:
            pipe.synchronize(); pipe.profile(True, 256)
        pipe.apply_device(frames[done % P].data_ptr(), masks.data_ptr(), n_frames=T)
        done += T
    pipe.synchronize()
    a, b, n = pipe.profile_query()
    px = S * T * h * w
    print("FD T=%d: blur+detector %.1f us / step (%.2f us per frame-set, %.0f GB/s on 10 B/px), morph %.1f us"
          % (T, 1e3 * a / n, 1e3 * a / n / T, px * 10 / (a / n * 1e-3) / 1e9, 1e3 * b / n))
