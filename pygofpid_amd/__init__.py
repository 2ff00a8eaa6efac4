"""pygofpid_amd: PyGOFPID's per-frame foreground path on MI355X (gfx950).

    from pygofpid_amd.pipeline import GOFPID          # drop-in for pygofpid.pipeline.GOFPID
    from pygofpid_amd import cv                       # the cv2 calls of steps 1-3, on the GPU
    from pygofpid_amd.segmentation import FrameDifferencing
    from pygofpid_amd.foreground import ForegroundPipeline   # fused steps 1-3, batch of streams

The HIP library (csrc/libgofpid_hip.so, C ABI in include/gofpid_hip.h) is loaded
on first use; there is no CPU fallback.
"""
__version__ = "0.1.0"
