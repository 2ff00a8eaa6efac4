"""Does the memory-bound morphology hide under the issue-bound MOG2 launch of OTHER streams?
One pipeline of 8 streams on one HIP stream vs two pipelines of 4 streams on two HIP streams
(same frames, same total work), 8 frames per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w, T = 8, 32, 1080, 1920, 8
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17)          # [P][S][h][w][3]
torch.cuda.synchronize()

def run(nq):
    per = S // nq
    ctxs = [_lib.Context(0) for _ in range(nq)]
    pipes = [ForegroundPipeline((h, w, 3), streams=per, ctx=c) for c in ctxs]
    # each queue gets its own contiguous copy of its streams' frames and its own masks
    fr = [frames[:, q * per:(q + 1) * per].contiguous() for q in range(nq)]
    mk = [torch.empty((T, per, h, w), dtype=torch.uint8, device=dev) for _ in range(nq)]
    torch.cuda.synchronize()
    def step(t):
        for q in range(nq):
            pipes[q].apply_device(fr[q][(t * T) % P].data_ptr(), mk[q].data_ptr(), n_frames=T)
    for t in range(64):
        step(t)
    for p in pipes:
        p.synchronize()
    K = 256
    t0 = time.perf_counter()
    for t in range(64, 64 + K):
        step(t)
    for p in pipes:
        p.synchronize()
    dt = time.perf_counter() - t0
    print("%d queue(s) x %d streams: %.1f us per step of %d frames, %.0f frames/s"
          % (nq, per, dt / K * 1e6, S * T, S * T * K / dt))

for nq in (1, 2, 4, 1, 2):
    run(nq)
