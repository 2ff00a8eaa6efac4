"""FrameDifferencing pipeline (blur -> FD -> erode/dilate) on the bench geometry: kernel times for BGR frames,
grey frames, and BGR frames converted to grey (step 0 fused / GOF_NO_FUSE_STEP0=1 in the environment: own kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w = 8, 32, 1080, 1920
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17)
gray = frames[..., 1].contiguous()
masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx = _lib.Context(0)
for name, shape, src, kw, bpp in (("BGR", (h, w, 3), frames, {}, 4), ("grey", (h, w), gray, {}, 2),
                                  ("BGR -> grey", (h, w, 3), frames, {"convert": "BGR2GRAY"}, 4)):
    for T in (1, 8):
        pipe = ForegroundPipeline(shape, streams=S, detector="FD", ctx=ctx, **kw)
        done = 0
        for i in range(64 + 128):
            if i == 64:
                pipe.synchronize(); pipe.profile(True, 256)
            pipe.apply_device(src[done % P].data_ptr(), masks.data_ptr(), n_frames=T)
            done += T
        pipe.synchronize()
        a, b, n = pipe.profile_query()
        px = S * T * h * w
        print("FD %-11s T=%d: detector (+ blur, + step 0 when fused) %.1f us / step (%.2f us per frame-set, %.0f GB/s on %d B/px), morph %.1f us, foreground %.4f"
              % (name, T, 1e3 * a / n, 1e3 * a / n / T, px * bpp / (a / n * 1e-3) / 1e9, bpp, 1e3 * b / n,
                 float((masks[:T] != 0).float().mean())))
