#!/usr/bin/env python3
"""Step 0 (grey conversion) and the 3x3 box filter fused into the detector kernel vs their own
kernels (GOF_NO_FUSE_STEP0 / GOF_NO_FUSE_BOX): whole-step time on device-resident frames,
8 x 1080p streams, 8 frames per step and 1 frame per step.  Each arm in its own process."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(kind, T):
    sys.path.insert(0, ROOT)
    import torch

    import bench
    from pygofpid_amd import _lib
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, P = 8, 1080, 1920, 64
    dev = torch.device("cuda", 0)
    frames = bench.make_pool(torch, S, P, h, w, dev, seed=17)
    masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
    ctx = _lib.Context(0)
    if kind == "convert":
        pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, convert="BGR2GRAY")
    elif kind in ("fd_convert", "knn_convert"):
        pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, convert="BGR2GRAY", detector=kind[:-8].upper())
    else:
        pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, blur=("box", 3, 3))
    i = 0

    def step():
        nonlocal i
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
        i += 1
    for _ in range(600 // T):
        step()
    pipe.synchronize()
    n = 400 // T * 2
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    pipe.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"us_per_step": dt * 1e6, "frames_per_s": S * T / dt,
                      "mbar": pipe.mean_modes() if pipe.detector == "MOG2" else 0.0}))


def main():
    for kind, knob in (("convert", "GOF_NO_FUSE_STEP0"), ("box", "GOF_NO_FUSE_BOX"),
                       ("fd_convert", "GOF_NO_FUSE_STEP0"), ("knn_convert", "GOF_NO_FUSE_STEP0")):
        for T in (8, 1):
            row = []
            for fused in (True, False):
                env = dict(os.environ, GOF_FUSION_CHILD="%s,%d" % (kind, T))
                if not fused:
                    env[knob] = "1"
                out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                row.append(json.loads(line[-1]) if line else {"us_per_step": float("nan"), "frames_per_s": 0, "mbar": 0})
            what = {"convert": "BGR -> grey (step 0) + GaussianBlur 3x3 + MOG2 (grey model) + 5x5 erode / dilate",
                    "box": "cv.blur 3x3 + MOG2 (BGR model) + 5x5 erode / dilate",
                    "fd_convert": "BGR -> grey (step 0) + GaussianBlur 3x3 + frame differencing + 5x5 erode / dilate",
                    "knn_convert": "BGR -> grey (step 0) + GaussianBlur 3x3 + KNN (grey model) + 5x5 erode / dilate"}[kind]
            print("%s, 8 x 1080p, %d frame(s) per step:\n    fused %8.1f us per step (%7.0f frames/s)   own kernel %8.1f us per step (%7.0f frames/s)   m-bar %.2f"
                  % (what, T, row[0]["us_per_step"], row[0]["frames_per_s"], row[1]["us_per_step"], row[1]["frames_per_s"], row[0]["mbar"]))


if __name__ == "__main__":
    if os.environ.get("GOF_FUSION_CHILD"):
        k, t = os.environ["GOF_FUSION_CHILD"].split(",")
        child(k, int(t))
    else:
        main()
