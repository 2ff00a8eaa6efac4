"""One-off soak: 1080p, 2 streams, 1 200 frames through 8-frame launches (learning-rate ramp,
steady state, objects with shadows entering / leaving, modes pruned) against the CPU oracle:
every mask and the final model state, bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cpu as oracle
from pygofpid_amd.foreground import ForegroundPipeline
from pygofpid_amd.synth import Clip

oracle.build()
h, w, c, S = 1080, 1920, 3, 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8          # frames per call
DET = sys.argv[3] if len(sys.argv) > 3 else "MOG2"
clips = [Clip(h, w, c, seed=900 + s, n_objects=5, speed=3) for s in range(S)]
gpu = ForegroundPipeline((h, w, c), streams=S, detector=DET)
refs = [oracle.Pipeline(h, w, c, detector=DET) for _ in range(S)]
t0 = time.time()
bad = 0
for t in range(0, N, T):
    frames = np.stack([np.stack([cl.frame(t + i) for cl in clips]) for i in range(T)])
    got = gpu.apply_many(frames).reshape(T, S, h, w)
    for i in range(T):
        for s in range(S):
            want = refs[s].apply(frames[i, s])
            if not np.array_equal(got[i, s], want):
                bad += 1
                print("MISMATCH frame %d stream %d: %d px" % (t + i, s, int((got[i, s] != want).sum())))
    if t % 200 < T:
        print("frame %d, %.0f s" % (t, time.time() - t0), flush=True)
st = gpu.mog2_state() if DET == "MOG2" else None
n = h * w
for s in range(S if DET == "MOG2" else 0):
    ref = refs[s].export()
    live = np.arange(5)[:, None] < ref["modes"][None, :]
    ok = (np.array_equal(st["modes"][s * n:(s + 1) * n], ref["modes"])
          and np.array_equal(st["weight"][:, s * n:(s + 1) * n][live], ref["weight"][live])
          and np.array_equal(st["var"][:, s * n:(s + 1) * n][live], ref["var"][live])
          and all(np.array_equal(st["mean"][:, ch, s * n:(s + 1) * n][live], ref["mean"][:, ch][live]) for ch in range(c)))
    print("stream %d final model state identical: %s" % (s, ok))
    bad += 0 if ok else 1
if DET == "KNN":
    kst = gpu.knn_state()
    for s in range(S):
        ref = refs[s].export()
        ok = (np.array_equal(kst["samples"][:, :, s * n:(s + 1) * n], ref["samples"])
              and np.array_equal(kst["state"][:, s * n:(s + 1) * n], ref["state"]))
        print("stream %d final sample model identical: %s" % (s, ok))
        bad += 0 if ok else 1
print("soak %s (%s): %d frames x %d streams, %d mismatches" % ("OK" if bad == 0 else "FAILED", DET, N, S, bad))
sys.exit(1 if bad else 0)
