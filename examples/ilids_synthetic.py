"""i-LIDS sterile-zone configuration on a synthetic stand-in (BASELINE.json config 5).

The reference's examples/dataset_ilids.py runs one GOFPID per video of the
i-LIDS dataset with a per-view perimeter / perspective (lines 52-66 for view 1)
and alarm_def duration_min = 15 (lines 118-121).  The dataset is not available
offline, so this script feeds 720x576 synthetic clips through the same
configuration, one GOFPID per stream, streams sharded round-robin over the
ranks (one process per GPU, no collective):

    python examples/ilids_synthetic.py --frames 120
    python -m torch.distributed.run --nproc-per-node 8 examples/ilids_synthetic.py
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pygofpid_amd import distributed as gd          # noqa: E402
from pygofpid_amd.pipeline import GOFPID            # noqa: E402
from pygofpid_amd.synth import Clip                 # noqa: E402

# the 33 test videos of examples/dataset_ilids.py:28-44; character 5 is the view
VIDEO_NAMES = (
    ['SZTEA101a', 'SZTEA101b', 'SZTEA102a', 'SZTEA102b', 'SZTEA103a', 'SZTEA104a', 'SZTEA105a',
     'SZTEA201a', 'SZTEA201b', 'SZTEA202a', 'SZTEA202b', 'SZTEA203a', 'SZTEA204a']
    + ['SZTEN101' + c for c in 'abcd'] + ['SZTEN102' + c for c in 'abcd'] + ['SZTEN103a', 'SZTEN103b']
    + ['SZTEN201' + c for c in 'abcde'] + ['SZTEN202' + c for c in 'abcd'] + ['SZTEN203a'])

VIEW = {
    '1': dict(perimeter=np.array([[1.000, 0.138], [0.600, 0.138], [0.463, 0.338], [0.000, 0.700],
                                  [0.000, 1.000], [1.000, 1.000]]),
              perspective=np.array([[0.65138889, 0.56770833], [0.57500000, 0.85243056],
                                    [0.68333333, 0.20833333], [0.72777778, 0.34548611]])),
    '2': dict(perimeter=np.array([[1.000, 0.750], [0.628, 0.388], [0.487, 0.205], [0.487, 0.112],
                                  [0.432, 0.000], [0.000, 0.000], [0.000, 1.000], [1.000, 1.000]]),
              perspective=np.array([[0.49444444, 0.56250000], [0.40555556, 0.88541667],
                                    [0.31388889, 0.06944444], [0.27500000, 0.19965278]])),
}


def get_gofpid(view):
    cfg = VIEW[view]
    return GOFPID(
        post_filter={'perimeter': cfg['perimeter'].copy(), 'anchor': 'bottom',
                     'perspective': cfg['perspective'].copy(), 'perspective_coeff': 0.5,
                     'presence_min': 3, 'distance_min': 0.25},
        tracking={'factor': 1.5, 'absence_max': 10},
        alarm_def={'keep_alarm': True, 'duration_min': 15},
    ).initialize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    args = ap.parse_args()
    rank, world = gd.init()
    mine = gd.shard_streams(len(VIDEO_NAMES), world, rank)
    el, frames, alarms = 0.0, 0, 0
    for s in mine:
        name = VIDEO_NAMES[s]
        gofpid = get_gofpid(name[5])
        clip = Clip(576, 720, 3, seed=100 + s, n_objects=2)     # i-LIDS frames are 720 x 576
        video = clip.frames(args.frames)                        # "decode" outside the timed part
        t0 = time.perf_counter()
        for frame in video:
            alarms += bool(gofpid.detect(frame))
        el += time.perf_counter() - t0
        frames += len(video)
    total = gd.sum_over_ranks(frames)
    slowest = gd.max_over_ranks(el)
    if rank == 0:
        print("i-LIDS synthetic: %d videos over %d rank(s), %d frames in %.1f s -> %.0f frames/s "
              "end to end (host frames, steps 1-3 on the GPU, steps 4-7 on the host)"
              % (len(VIDEO_NAMES), world, int(total), slowest, total / slowest))


if __name__ == "__main__":
    main()
