#!/usr/bin/env python3
"""GOFPID.detect() on host frames under `rocprofv3 --kernel-trace --memory-copy-trace`: 300 frames of
one 1080p BGR stream (benchmark scene); profiles/micro/detect_trace_read.py turns the trace into the
per-frame timeline (upload, each kernel, gaps, records down)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from pygofpid_amd.pipeline import GOFPID

h, w, P = 1080, 1920, 100
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, 1, P, h, w, dev, seed=17)
host = [frames[i, 0].cpu().numpy() for i in range(P)]
del frames
g = GOFPID().initialize()
for _ in range(3):
    for f in host:
        g.detect(f)
