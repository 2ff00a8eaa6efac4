"""Where does the general MOG2 loop spend its time?  Busy scene (20 objects), T = 8, with the
shadow test on / off and with fewer mixture slots."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w, T = 8, 32, 1080, 1920, 8
nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17, n_objects=nobj)
masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx = _lib.Context(0)
for label, over in (("defaults", {}), ("no shadow test", {"detect_shadows": 0}), ("3 mixtures", {"nmixtures": 3})):
    p = _lib.Mog2Params()
    _lib.load().gof_mog2_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    pipe = ForegroundPipeline((h, w, 3), streams=S, mog2_params=p, ctx=ctx)
    done = 0
    for i in range(64 + 128):
        if i == 64:
            pipe.synchronize(); pipe.profile(True, 256)
        pipe.apply_device(frames[done % P].data_ptr(), masks.data_ptr(), n_frames=T)
        done += T
    pipe.synchronize()
    a, b, n = pipe.profile_query()
    print("%d objects, %-15s: detector %.1f us / launch, morph %.1f us, mbar %.3f" % (nobj, label, 1e3 * a / n, 1e3 * b / n, pipe.mean_modes()))
