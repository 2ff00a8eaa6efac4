#!/usr/bin/env python3
"""A/B of builds of libgofpid_hip.so on bench.py's scene (8 x 1080p, 256-frame pool, steady state):
detector + morphology kernel time per 8-frame launch, HIP events on the launch stream.

    python profiles/micro/ab_scene.py [name=path.so[@ENV=VALUE...] ...]      (no argument: the in-tree build)

Each build runs in its own process (the library is chosen at import time); the builds are run
round-robin, REPEAT (env, default 2) times each, so that box drift shows up as spread.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import bench
    from pygofpid_amd import _lib
    from pygofpid_amd.foreground import ForegroundPipeline
    S, T, h, w = 8, int(os.environ.get("FRAMES", "8")), 1080, 1920
    pool = int(os.environ.get("POOL", "256"))
    dev = torch.device("cuda", 0)
    if os.environ.get("NOISE"):
        frames = bench.make_noise_pool(torch, S, 64, h, w, dev, 99)
    else:
        frames = bench.make_pool(torch, S, pool, h, w, dev, seed=17, n_objects=int(os.environ.get("OBJECTS", "6")))
    P = frames.shape[0]
    masks = torch.empty((T, S, h, w), dtype=torch.uint8, device=dev)
    ctx = _lib.Context(0)
    morph = [("erode", np.ones((5, 5), np.uint8)), ("dilate", np.ones((5, 5), np.uint8))]
    pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, morph=morph)
    i = 0
    for _ in range(int(os.environ.get("PREROLL_STEPS", "928"))):
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
        i += 1
    pipe.synchronize()
    n = 4 * (P // T)
    pipe.profile(True, n)
    for _ in range(n):
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
        i += 1
    pipe.synchronize()
    a, b, k = pipe.profile_query()
    print(json.dumps({"det_us": 1e3 * a / k, "morph_us": 1e3 * b / k, "n": k, "mbar": pipe.mean_modes()}))


def main():
    builds = [a.split("=", 1) for a in sys.argv[1:]] or [["in-tree", ""]]
    rep = int(os.environ.get("REPEAT", "2"))
    res = {n: [] for n, _ in builds}
    for _ in range(rep):
        for name, path in builds:
            env = dict(os.environ, GOF_AB_CHILD="1")
            path, *knobs = path.split("@")   # name=path@ENV=VALUE@...: environment knobs of this arm
            for kv in knobs:
                k, v = kv.split("=", 1)
                env[k] = v
            if path:
                env["GOFPID_LIB"] = os.path.abspath(path)
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-400:])
                continue
            res[name].append(json.loads(line[-1]))
    for name, rs in res.items():
        if rs:
            print("%-14s detector %s us   morph %s us   (m-bar %.3f, %d launches each)"
                  % (name, " / ".join("%.1f" % r["det_us"] for r in rs), " / ".join("%.1f" % r["morph_us"] for r in rs),
                     rs[-1]["mbar"], rs[-1]["n"]))


if __name__ == "__main__":
    child() if os.environ.get("GOF_AB_CHILD") else main()
