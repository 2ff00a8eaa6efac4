"""One configuration for counter collection: static scene, fused blur, T = 8 (or argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pygofpid_amd import _lib
from pygofpid_amd.foreground import ForegroundPipeline
S, P, h, w = 8, 32, 1080, 1920
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nobj = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
frames = bench.make_pool(torch, S, P, h, w, dev, 17, n_objects=nobj)
masks = torch.empty((8, S, h, w), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=_lib.Context(0))
done = 0
for _ in range(512 // T + 64):
    pipe.apply_device(frames[done % P].data_ptr(), masks.data_ptr(), n_frames=T)
    done += T
pipe.synchronize()
print("mbar", pipe.mean_modes())
