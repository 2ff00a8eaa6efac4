#!/usr/bin/env python3
"""Which path do the waves of the 8-frame MOG2 launch take on the benchmark scene?

Runs bench.py's scene (8 x 1080p, 256-frame pool, pre-rolled to the steady state) through a
-DGOF_PATH_STATS build of the library (make -C pygofpid_amd/csrc stats) and prints the wave-frame
counts per loop of mog2_frames_kernel and, for the general update, what inside it actually runs.
Never the product library: GOFPID_LIB must point at libgofpid_hip_stats.so.

    GOFPID_LIB=pygofpid_amd/csrc/libgofpid_hip_stats.so python profiles/micro/path_stats.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GOFPID_LIB", os.path.join(ROOT, "pygofpid_amd", "csrc", "libgofpid_hip_stats.so"))

import torch  # noqa: E402

import bench  # noqa: E402
from pygofpid_amd import _lib  # noqa: E402
from pygofpid_amd.foreground import ForegroundPipeline  # noqa: E402

NAMES = ["settled", "single", "first_fits", "general", "gen_lanes_not_ok", "gen_insert", "gen_reorder",
         "gen_not_bg", "gen_lanes_not_bg", "gen_sum_max_modes", "gen_fit_late", "gen_slot1", "gen_slot2",
         "gen_slot3", "gen_slot4", "waves", "waves_multi", "fit0", "fit1", "fit2", "fit3", "fit4",
         "shadow_iters", "gen_lanes_insert", "insert_reorder"]


def main():
    S, T, h, w = 8, 8, 1080, 1920
    pool = int(os.environ.get("POOL", "256"))
    noise = bool(os.environ.get("NOISE"))
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    lib = _lib.load()
    fn = lib.gof_debug_path_stats   # exported by the stats build only
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 32)()
    frames = (bench.make_noise_pool(torch, S, 64, h, w, dev, 99) if noise
              else bench.make_pool(torch, S, pool, h, w, dev, seed=17))
    P = frames.shape[0]
    masks = torch.empty((T, S, h, w), dtype=torch.uint8, device=dev)
    import numpy as np
    morph = [("erode", np.ones((5, 5), np.uint8)), ("dilate", np.ones((5, 5), np.uint8))]
    pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, morph=morph)
    steps = int(os.environ.get("PREROLL_STEPS", "928"))
    for i in range(steps):
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
    pipe.synchronize()
    assert fn(buf) == 0   # clear
    n = P // T           # one pass over the pool
    for i in range(steps, steps + n):
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
    pipe.synchronize()
    assert fn(buf) == 0
    v = dict(zip(NAMES, [int(x) for x in buf]))
    wf = v["settled"] + v["single"] + v["first_fits"] + v["general"]
    print("scene: %s, %d launches of %d x %d x %dx%d after %d pre-roll launches, mean modes %.3f"
          % ("noise" if noise else "pool %d" % P, n, S, T, w, h, steps, pipe.mean_modes()))
    print("waves per launch %.0f (multi-mode on entry: %.1f %%)" % (v["waves"] / n, 100.0 * v["waves_multi"] / v["waves"]))
    print("wave-frames per launch %.0f" % (wf / n))
    for k in ("settled", "single", "first_fits", "general"):
        print("  %-12s %6.2f %%" % (k, 100.0 * v[k] / wf))
    g = max(1, v["general"])
    print("general update, per wave-frame that runs it:")
    print("  lanes whose first mode does not fit / not background  %5.1f of 64" % (v["gen_lanes_not_ok"] / g))
    print("  largest mode count in the wave                        %5.2f" % (v["gen_sum_max_modes"] / g))
    for m in range(1, 5):
        print("  slot %d live in some lane                              %5.1f %%" % (m, 100.0 * v["gen_slot%d" % m] / g))
    for m in range(5):
        print("  lanes fitting at slot %d                               %5.1f of 64" % (m, v["fit%d" % m] / g))
    print("  some lane fits at a slot >= 1                         %5.1f %%" % (100.0 * v["gen_fit_late"] / g))
    print("  some lane re-orders after its fit                     %5.1f %%" % (100.0 * v["gen_reorder"] / g))
    print("  some lane opens a new mode                            %5.1f %%  (%.2f lanes)" % (100.0 * v["gen_insert"] / g, v["gen_lanes_insert"] / g))
    print("  ... and a swap happens in the insertion chain         %5.2f per frame" % (v["insert_reorder"] / g))
    print("  some lane not background (shadow test runs)           %5.1f %%  (%.1f lanes)" % (100.0 * v["gen_not_bg"] / g, v["gen_lanes_not_bg"] / g))
    print("  shadow-test slots examined                            %5.2f" % (v["shadow_iters"] / g))
    print("raw:", v)


if __name__ == "__main__":
    main()
