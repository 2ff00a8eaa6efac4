"""detect() on the examples/video.py geometry (320x240, BASELINE config 0) and at 720x576: is the GPU
blob extraction worth its fixed cost on small masks, or is mask-down + host follower quicker?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pygofpid_amd.pipeline import GOFPID
from pygofpid_amd.synth import Clip


def detect_host(g, f):
    if not g.input_shape_:
        g.input_shape_ = f.shape; g._calib_first_frame()
    g._foreground(f); g._gpu_blobs = None
    g._find_blob(); g._track_blob(); g._post_filter()
    return g._maintain_alarm(g._detect_blob())


for (h, w) in ((240, 320), (576, 720), (1080, 1920)):
    frames = Clip(h, w, 3, seed=17).frames(32)
    for name, fn in (("device contours", lambda g, f: g.detect(f)), ("host contours", detect_host)):
        g = GOFPID().initialize()
        for f in frames[:8]:
            fn(g, f)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2:
            fn(g, frames[n % 32]); n += 1
        print("%4dx%-4d detect(), %-16s %7.0f frames/s" % (w, h, name + ":", n / (time.perf_counter() - t0)))
