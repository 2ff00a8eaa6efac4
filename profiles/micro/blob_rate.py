"""Cost of the GPU blob extraction (csrc/gof_blobs.hip) on the benchmark scene's final masks:
64 masks (8 streams x 8 frames) per call and 1 mask per call, device-resident, torch events;
and the end-to-end detect() forms on host frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.blobs import BlobExtractor
from pygofpid_amd.foreground import ForegroundPipeline

dev = torch.device("cuda", 0)
S, h, w, T, P = 8, 1080, 1920, 8, 64
frames = bench.make_pool(torch, S, P, h, w, dev, seed=17)
masks = torch.empty((T, S, h, w), dtype=torch.uint8, device=dev)
ctx = _lib.Context(0)
pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx)
for i in range(64):
    pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
pipe.synchronize()
fg = float((masks != 0).float().mean())
ex = BlobExtractor(T * S, h, w, ctx=ctx)
stream = torch.cuda.ExternalStream(ctx.stream, device=dev)


def timed(n_img, reps):
    for _ in range(3):
        ex.extract_device(masks.data_ptr(), n_img)
    ctx.synchronize()
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            ex.extract_device(masks.data_ptr(), n_img)
        e1.record(stream)
    ctx.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t1 = timed(1, 50)
import ctypes as C
dbg = (C.c_uint32 * 16)()
_lib.load().gof_blobs_debug_counters(ex.handle, 0, dbg)
d = list(dbg)
print("1-mask call: cracks %d roots %d flags %d tiles %d; finishing kernel phases A..F (us): %s"
      % (d[0], d[1], d[2], d[3], [round(((d[5 + i] - d[4 + i]) & 0xffffffff) / 100.0, 1) for i in range(7)]))
t64 = timed(T * S, 20)
res = [ex.result(i) for i in range(T * S)]
nb = sum(len(r[0]) for r in res)
npts = sum(len(r[1]) for r in res)
print("scene: foreground fraction %.4f; 64 masks: %d blobs >= 100 px^2, %d contour points" % (fg, nb, npts))
print("blob extraction, 64 x 1080p masks per call: %.1f us  (%.2f us per mask; %.1f GB/s of mask bytes)"
      % (t64, t64 / 64, 64 * h * w / t64 / 1e3))
print("blob extraction,  1 x 1080p mask  per call: %.1f us" % t1)
