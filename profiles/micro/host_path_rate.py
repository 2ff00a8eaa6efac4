"""End-to-end (host buffer in, host mask out) rate of the drop-in path at 1080p:
what GOFPID.detect() pays on top of the kernels is PCIe (6.2 MB in + 2.1 MB out
per frame) and, for detect(), the host-side steps 4-7."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pygofpid_amd.foreground import ForegroundPipeline
from pygofpid_amd.pipeline import GOFPID
from pygofpid_amd.synth import Clip

h, w = 1080, 1920
clip = Clip(h, w, 3, seed=17)
frames = np.stack(clip.frames(16))
p = ForegroundPipeline((h, w, 3))
for f in frames[:4]:
    p.apply(f)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3:
    p.apply(frames[n % 16]); n += 1
print("ForegroundPipeline.apply      (1 frame / call, pageable host memory): %.0f frames/s" % (n / (time.perf_counter() - t0)))
p8 = ForegroundPipeline((h, w, 3))
p8.apply_many(frames[:8])
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3:
    p8.apply_many(frames[(n % 2) * 8:(n % 2) * 8 + 8]); n += 8
print("ForegroundPipeline.apply_many (8 frames / call):                       %.0f frames/s" % (n / (time.perf_counter() - t0)))
g = GOFPID().initialize()
for f in frames[:4]:
    g.detect(f)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3:
    g.detect(frames[n % 16]); n += 1
print("GOFPID.detect (steps 1-3 GPU + steps 4-7 host):                        %.0f frames/s" % (n / (time.perf_counter() - t0)))
# step 4's front end: on the device (records + points down) vs on the host (mask down, border following)
pb = ForegroundPipeline((h, w, 3))
for f in frames[:4]:
    pb.apply_blobs(f)
t0 = time.perf_counter(); n = 0; nbytes = 0
while time.perf_counter() - t0 < 3:
    r = pb.apply_blobs(frames[n % 16])[0]; n += 1
    nbytes += 16 + r[0].nbytes + r[1].nbytes // 2
print("ForegroundPipeline.apply_blobs (steps 1-3 + GPU contours / area filter / moments):  %.0f frames/s, "
      "%.1f KB down per frame (mask: %.0f KB)" % (n / (time.perf_counter() - t0), nbytes / n / 1e3, h * w / 1e3))
from pygofpid_amd import blobs as gb
ph = ForegroundPipeline((h, w, 3))
for f in frames[:4]:
    ph.apply(f)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3:
    gb.host_result(ph.apply(frames[n % 16]), 100.0); n += 1
print("same result through the host: apply() + host border follower + filter + sums:        %.0f frames/s" % (n / (time.perf_counter() - t0)))
gh = GOFPID().initialize()
def detect_host(g, f):
    if not g.input_shape_:
        g.input_shape_ = f.shape; g._calib_first_frame()
    g._foreground(f); g._gpu_blobs = None
    g._find_blob(); g._track_blob(); g._post_filter()
    return g._maintain_alarm(g._detect_blob())
for f in frames[:4]:
    detect_host(gh, f)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3:
    detect_host(gh, frames[n % 16]); n += 1
print("GOFPID.detect with step 4 forced onto the host path (the r01 form):                  %.0f frames/s" % (n / (time.perf_counter() - t0)))
gm = GOFPID().initialize()
gm.detect_many(list(frames[:8]))
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3:
    gm.detect_many(list(frames[(n % 2) * 8:(n % 2) * 8 + 8])); n += 8
print("GOFPID.detect_many, 8 frames per call (one upload, 8-frame kernels, one download):   %.0f frames/s" % (n / (time.perf_counter() - t0)))
