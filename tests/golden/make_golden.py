"""Regenerates tests/golden/*.npz.

PROVENANCE: these vectors are produced by THIS repo's CPU oracle
(oracle/gof_oracle.c), not by the reference: the reference's arithmetic lives
in opencv-python, which is absent here, and the reference's own tests hold no
value fixtures for this path -> "parity unpinned".  They freeze
the oracle so that a later edit of it (or of the kernels) cannot drift silently,
and they travel to the GPU box where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cpu  # noqa: E402
from pygofpid_amd.synth import Clip, uniform_noise_frames  # noqa: E402


def main():
    cpu.build()
    # 1. default pipeline on a small synthetic clip (BGR): blur 3x3 sigma 4 ->
    #    MOG2 -> erode 5x5 -> dilate 5x5, with every intermediate.
    clip = Clip(40, 56, 3, seed=17, n_objects=3)
    frames = np.stack(clip.frames(32))
    mog2 = cpu.MOG2()
    k5 = np.ones((5, 5), np.uint8)
    blurred, raw, final = [], [], []
    for f in frames:
        b = cpu.gaussian_blur(f, (3, 3), 4.0)
        m = mog2.apply(b)
        blurred.append(b)
        raw.append(m)
        final.append(cpu.dilate(cpu.erode(m, k5), k5))
    st = mog2.export()
    np.savez_compressed(os.path.join(HERE, "pipeline_mog2_bgr.npz"), frames=frames,
                        blurred=np.stack(blurred), mog2_mask=np.stack(raw),
                        final_mask=np.stack(final), weight=st["weight"], var=st["var"],
                        mean=st["mean"], modes=st["modes"])
    # 2. frame differencing, reference-test style input, threshold 7.5
    #    (reference tests/test_segmentation.py:13)
    fr = np.stack(uniform_noise_frames((32, 28, 3), 4, seed=17))
    fd = cpu.FrameDifferencing(7.5)
    np.savez_compressed(os.path.join(HERE, "fd_bgr.npz"), frames=fr,
                        masks=np.stack([fd.apply(f) for f in fr]))
    # 3. grey MOG2 on uniform noise (reference tests/test_pipeline.py:98-109 style)
    fr = np.stack(uniform_noise_frames((32, 20), 6, seed=18))
    m = cpu.MOG2()
    np.savez_compressed(os.path.join(HERE, "mog2_gray_noise.npz"), frames=fr,
                        masks=np.stack([m.apply(f) for f in fr]))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
