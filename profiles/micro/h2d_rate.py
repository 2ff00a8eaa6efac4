#!/usr/bin/env python3
"""What does the host link give?  H2D / D2H time of one 1080p BGR frame (6.2 MB) and of 8, from
pageable and from pinned host memory, blocking copies on one stream; and the stages of one
GOFPID.detect() call.  The numbers the drop-in path's "fraction of H2D bandwidth" is quoted against."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

h, w = 1080, 1920
dev = torch.device("cuda", 0)


def rate(src, dst, n=200):
    for _ in range(5):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e6, src.numel() / dt / 1e9


for frames in (1, 8):
    nbytes = frames * h * w * 3
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pageable = torch.from_numpy(np.random.randint(0, 255, nbytes, dtype=np.uint8))
    pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    pinned.copy_(pageable)
    for name, hbuf in (("pageable", pageable), ("pinned", pinned)):
        us, gbs = rate(hbuf, d)
        print("H2D %d frame(s) %-8s: %7.1f us  %5.1f GB/s" % (frames, name, us, gbs))
        us, gbs = rate(d, hbuf)
        print("D2H %d frame(s) %-8s: %7.1f us  %5.1f GB/s" % (frames, name, us, gbs))
    # host memcpy pageable -> pinned (what a staging ring costs the CPU), 1 thread
    t0 = time.perf_counter()
    for _ in range(20):
        pinned.copy_(pageable)
    dt = (time.perf_counter() - t0) / 20
    print("host copy pageable -> pinned, %d frame(s), torch (intra-op threads %d): %7.1f us  %5.1f GB/s"
          % (frames, torch.get_num_threads(), dt * 1e6, nbytes / dt / 1e9))

# small D2H (blob records)
d = torch.empty(32768, dtype=torch.uint8, device=dev)
p = torch.empty(32768, dtype=torch.uint8).pin_memory()
us, _ = rate(d, p)
print("D2H 32 KB pinned: %.1f us" % us)

from pygofpid_amd.foreground import ForegroundPipeline  # noqa: E402
from pygofpid_amd.pipeline import GOFPID  # noqa: E402
from pygofpid_amd.synth import Clip  # noqa: E402

clip = Clip(h, w, 3, seed=17)
frames = np.stack(clip.frames(16))
g = GOFPID().initialize()
for f in frames[:8]:
    g.detect(f)
n = 300
t0 = time.perf_counter()
for i in range(n):
    g.detect(frames[i % 16])
dt = (time.perf_counter() - t0) / n
print("GOFPID.detect: %.1f us per frame (%.0f frames/s)" % (dt * 1e6, 1 / dt))
# the device part alone (upload + kernels + blob records down), no host steps 5-7
pb = ForegroundPipeline((h, w, 3))
for f in frames[:4]:
    pb.apply_blobs(f)
t0 = time.perf_counter()
for i in range(n):
    pb.apply_blobs(frames[i % 16])
dt2 = (time.perf_counter() - t0) / n
print("ForegroundPipeline.apply_blobs: %.1f us per frame -> host steps 5-7 + Python glue: %.1f us" % (dt2 * 1e6, (dt - dt2) * 1e6))
# kernels alone on device-resident frames
dfr = torch.from_numpy(frames).to(dev)
masks = torch.empty((1, 1, h, w), dtype=torch.uint8, device=dev)
pk = ForegroundPipeline((h, w, 3))
for i in range(8):
    pk.apply_device(dfr[i].data_ptr(), masks.data_ptr())
pk.synchronize()
t0 = time.perf_counter()
for i in range(n):
    pk.apply_device(dfr[i % 16].data_ptr(), masks.data_ptr())
pk.synchronize()
dt3 = (time.perf_counter() - t0) / n
print("steps 1-3 on device-resident frames, 1 frame per call, back to back: %.1f us per frame" % (dt3 * 1e6))
