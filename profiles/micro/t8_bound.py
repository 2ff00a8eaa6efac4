"""What bounds the 8-frame MOG2 launch?  Static scene (every wave on the single-mode path),
fused 3x3 blur vs no blur (raw 3 B/px input, no halo), T = 8 and T = 1.
   GOFPID_LIB=<other build> selects the library under test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w = 8, 32, 1080, 1920
nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17, n_objects=nobj)
masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx = _lib.Context(0)
for blur in ((3, 3, 4.0), None):
    for T in (8, 1):
        pipe = ForegroundPipeline((h, w, 3), streams=S, blur=blur, ctx=ctx)
        done = 0
        def step():
            global done
            pipe.apply_device(frames[done % P].data_ptr(), masks.data_ptr(), n_frames=T)
            done += T
        for _ in range(512 // T):
            step()
        pipe.synchronize()
        pipe.profile(True, 256)
        for _ in range(128):
            step()
        pipe.synchronize()
        a, b, n = pipe.profile_query()
        print("objects %d blur %-12s T=%d: detector %.1f us / launch (%.2f us per frame), morph %.1f us, mbar %.3f"
              % (nobj, blur, T, 1e3 * a / n, 1e3 * a / n / T, 1e3 * b / n, pipe.mean_modes()))
        del pipe
