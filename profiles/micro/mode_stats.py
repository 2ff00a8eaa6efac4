"""How are multi-mode pixels distributed over 64-px wave segments in the bench scene?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w = 8, 32, 1080, 1920
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17)
masks = torch.empty((S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=_lib.Context(0))
for T in (64, 192, 320):
    t0 = 0 if T == 64 else T - 128
    for t in range(t0, T):
        pipe.apply_device(frames[t % P].data_ptr(), masks.data_ptr())
    pipe.synchronize()
    nm = pipe.mog2_state()["modes"].reshape(S * h, w // 64, 64)
    mx = nm.max(axis=2)
    print("after %d frames: mean modes %.4f | px with >=2: %.3f  >=3: %.3f | 64-px segments with any >=2: %.3f  >=3: %.3f  >=4: %.3f"
          % (T, nm.mean(), (nm >= 2).mean(), (nm >= 3).mean(), (mx >= 2).mean(), (mx >= 3).mean(), (mx >= 4).mean()))
    seg32 = pipe.mog2_state()["modes"].reshape(S * h, w // 32, 32).max(axis=2)
    print("    32-px (128-B line) segments with any >=2: %.3f" % (seg32 >= 2).mean())
