"""Bench scene after warm-up: how many 64-px wave segments are 'settled' (every pixel one mode
of weight exactly 1.0f)?  Decides whether the host-precomputed weight shortcut can pay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w = 8, 32, 1080, 1920
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17)
masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=_lib.Context(0))
done = 0
for T in (512, 520, 1024):
    while done < T:
        pipe.apply_device(frames[done % P].data_ptr(), masks.data_ptr())
        done += 1
    pipe.synchronize()
    st = pipe.mog2_state()
    nm = st["modes"].reshape(S * h, w // 64, 64)
    w0 = st["weight"][0].reshape(S * h, w // 64, 64)
    one = (nm == 1) & (w0 == np.float32(1.0))
    print("after %d frames: px single-mode %.4f | px single & w==1.0f %.4f | segments all single %.4f | segments all settled %.4f"
          % (T, (nm == 1).mean(), one.mean(), (nm.max(axis=2) == 1).mean(), one.all(axis=2).mean()))
    vals, cnt = np.unique(w0[nm == 1], return_counts=True)
    o = np.argsort(-cnt)[:6]
    print("    commonest single-mode weights:", [(float(vals[i]), int(cnt[i])) for i in o])
    v0 = st["var"][0].reshape(S * h, w // 64, 64)
    at_min = (nm == 1) & (v0 == np.float32(4.0))
    print("    px single-mode with variance at var_min: %.4f | segments where all are: %.4f"
          % (at_min.mean(), at_min.all(axis=2).mean()))
