#!/usr/bin/env python3
"""Where GOFPID.detect() spends a frame (one 1080p BGR stream, host frames, benchmark scene):
the device call (upload + kernels + blob records down), the host steps 5-7 on the recorded
records, and a cProfile of detect() itself."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from pygofpid_amd.pipeline import GOFPID

h, w, P = 1080, 1920, 256
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, 1, P, h, w, dev, seed=17)
host = [frames[i, 0].cpu().numpy() for i in range(P)]
g = GOFPID().initialize()
for f in host:
    g.detect(f)
t0 = time.perf_counter()
for f in host:
    g.detect(f)
torch.cuda.synchronize()
dt_detect = (time.perf_counter() - t0) / P
fused = g._fused
res = []
t0 = time.perf_counter()
for f in host:
    res.append(fused.apply_blobs(f, min_area=100)[0])
dt_dev = (time.perf_counter() - t0) / P
t0 = time.perf_counter()
for r in res:
    g._gpu_blobs = r
    g._find_blob(), g._track_blob(), g._post_filter(), g._maintain_alarm(g._detect_blob())
dt_host = (time.perf_counter() - t0) / P
nb = sum(len(r[0]) for r in res if r is not None) / P
print("GOFPID.detect: %.1f us per frame; apply_blobs (upload, kernels, records down): %.1f us; host steps 5-7: %.1f us (%.1f blobs per frame, %d tracked)"
      % (dt_detect * 1e6, dt_dev * 1e6, dt_host * 1e6, nb, len(g.tracked_blobs_)))
# the device call in two halves: submit (upload -- blocking from pageable memory -- and the enqueue of every kernel) / wait + fetch
ts = tw = 0.0
for f in host:
    a = time.perf_counter()
    fused.submit_blobs_frames([f], 100, 0)
    b = time.perf_counter()
    fused.wait_blobs(0)
    c = time.perf_counter()
    ts += b - a
    tw += c - b
print("submit (upload + enqueue) %.1f us, wait for the records + fetch %.1f us" % (ts / P * 1e6, tw / P * 1e6))
import ctypes as C
from pygofpid_amd import _lib
lib = _lib.load()
ex = fused._slot_blobs[0]
t0 = time.perf_counter()
for f in host:
    fused.submit_blobs_frames([f], 100, 0)
    _lib.check(lib.gof_pipeline_wait_slot(fused._h, 0))
dt = (time.perf_counter() - t0) / P
print("submit + wait without fetching the records: %.1f us" % (dt * 1e6))
pr = cProfile.Profile()
pr.enable()
for f in host:
    g.detect(f)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
