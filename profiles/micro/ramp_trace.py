"""Why did the driver's `bench.py --steps 20 --warmup 5` (r01) time the 8-frame launch at 547 us
when the 256-step run timed it at 404 us?  Per-launch HIP-event durations from a cold start:
  (b) constant learning rate (frame counter preset far past `history`), cold GPU first;
  (a) fresh model (rate 1/(2t) until t = 250, i.e. step 31), GPU already warm.
A decay in (b) is clocks / power state; a decay confined to the first ~31 steps of (a) is the
learning-rate ramp (fewer settled waves while the rate is high)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline

dev = torch.device("cuda", 0)
S, h, w, T, P = 8, 1080, 1920, 8, 64
frames = bench.make_pool(torch, S, P, h, w, dev, seed=17)
masks = torch.empty((T, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx = _lib.Context(0)


def series(preset, n):
    """n back-to-back launches (no host synchronisation in between), one event pair each"""
    pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx)
    if preset:
        pipe.set_nframes(preset)
    pipe.profile(True, n)
    for i in range(n):
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)
    pipe.synchronize()
    d, m = pipe.profile_read(n)
    pipe.profile(False)
    mb = pipe.mean_modes()
    pipe.close()
    return [round(float(x) * 1e3, 1) for x in d], mb


b, mb_b = series(100000, 400)
a, mb_a = series(0, 400)
pick = lambda v: {str(i): v[i] for i in (0, 1, 2, 3, 4, 5, 8, 12, 16, 20, 24, 28, 31, 32, 40, 48, 64, 96, 128, 192, 256, 320, 399)}
print(json.dumps({"const_rate_cold_us": pick(b), "fresh_model_warm_us": pick(a),
                  "mean_modes": {"const_rate": mb_b, "fresh": mb_a},
                  "const_rate_mean_first20": sum(b[:20]) / 20, "const_rate_mean_last100": sum(b[-100:]) / 100,
                  "fresh_mean_steps5_25": sum(a[5:25]) / 20, "fresh_mean_last100": sum(a[-100:]) / 100}))
