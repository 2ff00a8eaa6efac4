#!/usr/bin/env python3
"""Headline benchmark: frames/s of the per-frame foreground path at 1080p.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1920x1080 BGR frames, GaussianBlur 3x3
(sigma 4, as the reference calls it) -> MOG2 -> erode 5x5 -> dilate 5x5, i.e.
steps 1-3 of GOFPID.detect (pygofpid/pipeline.py:461-476).  One "step" is one
pass of the path over one batch: one new frame for each of `--streams`
independent video streams (8 by default, so that the background models,
~0.4-1.7 GB in use, do not fit the 256 MiB Infinity Cache).  Frames are
synthetic and already resident in HBM when the timed region starts.

N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and
the MAX over ranks); every rank steps its own streams -- the path shards by
stream with no data-path collective ("weak" scaling).

Prints ONE JSON line on rank 0, including
  roofline      -- dominant kernel (fused blur + MOG2): algorithmic bytes per
                   launch / HIP-event time on the launch stream vs 8 TB/s
  cpu_baseline  -- the CPU oracle ("port" of the OpenCV algorithms, NOT OpenCV)
                   timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the CPU leg is a throughput run: let the OpenMP workers spin between the
# per-frame parallel regions (read by libgomp when oracle/ is loaded)
os.environ.setdefault("OMP_WAIT_POLICY", "active")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
EVENT_STRIDE = 4     # timed steps between two event-bracketed ones


def make_pool(torch, streams, pool, h, w, device, seed, n_objects=6):
    """[pool][streams][h][w][3] uint8 on the device: BASELINE.md section 3 recipe
    (gradient + texture background, N(0,2) noise, +60 rectangles moving 3 px per
    frame with a 0.7x shadow patch beside each)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    grad = 60.0 + 100.0 * xx / max(1, w - 1) + 40.0 * yy / max(1, h - 1)
    tint = torch.tensor([1.0, 0.92, 0.85], device=device)
    frames = torch.empty((pool, streams, h, w, 3), dtype=torch.uint8, device=device)
    oh, ow = max(4, h // 6), max(3, w // 16)
    for s in range(streams):
        tex = torch.randint(-12, 13, (h, w), generator=g, device=device).float()
        bg = (grad + tex)[..., None] * tint
        objs = []
        for _ in range(n_objects):
            objs.append((int(torch.randint(0, w, (1,), generator=g, device=device)),
                         int(torch.randint(0, max(1, h - oh), (1,), generator=g, device=device)),
                         1 if float(torch.rand(1, generator=g, device=device)) < 0.5 else -1))
        for t in range(pool):
            f = bg.clone()
            for (x0, y0, d) in objs:
                x = (x0 + d * 3 * t) % w
                f[y0:y0 + oh, min(w, x + ow):min(w, x + 2 * ow)] *= 0.7
                f[y0:y0 + oh, x:min(w, x + ow)] += 60.0
            f += torch.randn(f.shape, generator=g, device=device) * 2.0
            frames[t, s] = f.round().clamp(0, 255).to(torch.uint8)
    return frames


def cpu_baseline(frames_host, h, w, budget_s):
    """Time the CPU oracle on this host (checker reused as the baseline leg).
    The thread count is the best of a short sweep up to the CPUs this process
    may use; `cores` reports the count actually used for the timed sample."""
    from oracle import cpu as oracle
    oracle.build()
    n = len(frames_host)
    budget = oracle.cpu_budget()
    cands = sorted({t for t in (8, 16, 32, 64, 96, 128, budget) if t <= budget} or {1})

    def rate(threads, seconds, warm):
        p = oracle.Pipeline(h, w, 3, threads=threads)
        for i in range(warm):
            p.apply(frames_host[i % n])
        done, t0 = 0, time.perf_counter()
        while True:
            p.apply(frames_host[done % n])
            done += 1
            el = time.perf_counter() - t0
            if el >= seconds or done >= 4096:
                return done / el, done, el, p

    best = max(cands, key=lambda t: rate(t, 0.6, 4)[0]) if len(cands) > 1 else cands[0]
    fps, done, el, p = rate(best, budget_s, 8)
    return {"value": round(fps, 2), "unit": "frames/s", "cores": best, "kind": "port",
            "sample": "%d frames of 1 stream %dx%d BGR in %.1f s, OpenMP row bands over %d threads "
                      "(best of %s on a %d-CPU budget); OpenCV-algorithm CPU restatement "
                      "(oracle/gof_oracle.c), not OpenCV" % (done, w, h, el, best, cands, budget),
            "mean_modes": round(p.mean_modes(), 4)}


def load_traffic(path, key):
    """HBM bytes per launch from the committed PMC passes (profiles/), if the
    recorded configuration matches this run; else None."""
    # the named file, then the per-frame record kept next to it
    for cand in (path, os.path.join(os.path.dirname(path), "r01_traffic_T1.json")):
        try:
            with open(cand) as f:
                rec = json.load(f)
            if rec.get("config") == key:
                return rec.get("hbm_bytes_per_launch")
        except Exception:
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--streams", type=int, default=8, help="video streams stepped per GPU")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--pool", type=int, default=32, help="distinct synthetic frames per stream")
    ap.add_argument("--frames-per-step", type=int, default=8,
                    help="consecutive frames of every stream handed over per step (temporal "
                         "blocking: the MOG2 model crosses HBM once per 8 frames)")
    ap.add_argument("--objects", type=int, default=6, help="moving rectangles per stream")
    ap.add_argument("--morph", default="erode5,dilate5",
                    help="mat_morph list, e.g. erode7,dilate7,dilate7,erode7 (BASELINE config 3)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU leg")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "traffic_latest.json"))
    ap.add_argument("--no-events", action="store_true",
                    help="do not bracket kernels with HIP events (roofline block omitted)")
    args = ap.parse_args()

    import torch

    from pygofpid_amd import distributed as gd
    rank, world, local = gd.env_rank()
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local % max(1, ndev)          # one GPU per rank on the 8-GPU node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # RCCL ("nccl") on the GPU node; GOF_DIST_BACKEND=gloo lets several ranks share one
    # card for rehearsals (RCCL refuses two ranks on the same device)
    backend = os.environ.get("GOF_DIST_BACKEND", "nccl")
    gd.init(backend=backend, device=device if backend == "nccl" else None)   # no-op for 1 process
    red_dev = device if backend == "nccl" else "cpu"

    from pygofpid_amd import _lib
    from pygofpid_amd.foreground import ForegroundPipeline

    h, w, S, K, W = args.height, args.width, args.streams, args.steps, args.warmup
    T = args.frames_per_step
    if T < 1 or args.pool % T:
        raise SystemExit("--frames-per-step must divide --pool")
    frames = make_pool(torch, S, args.pool, h, w, device, seed=17 + rank, n_objects=args.objects)
    masks = torch.empty((T, S, h, w), dtype=torch.uint8, device=device)
    torch.cuda.synchronize()

    ctx = _lib.Context(dev_index)
    import numpy as np
    morph = []
    for item in [m for m in args.morph.split(",") if m]:
        op = "erode" if item.startswith("erode") else "dilate"
        kk = int(item[len(op):])
        morph.append((op, np.ones((kk, kk), np.uint8)))
    pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, morph=morph)
    P = args.pool

    def barrier():
        pipe.synchronize()
        torch.cuda.synchronize()
        gd.barrier()

    def step(i):   # frames [iT, iT+T) of every stream, consecutive in the pool
        pipe.apply_device(frames[(i * T) % P].data_ptr(), masks.data_ptr(), n_frames=T)

    for t in range(W):
        step(t)
    barrier()
    mbar_start = pipe.mean_modes()
    if not args.no_events:
        # HIP events around the kernels of every EVENT_STRIDE-th timed step (each record
        # costs the stream a few microseconds; a sample keeps that off `value`)
        pipe.profile(True, K, stride=EVENT_STRIDE)
    barrier()
    if os.environ.get("GOF_ABLATE_TIMED"):          # profiles/ ablations: flip after the warm-up
        os.environ["GOF_ABLATE"] = os.environ["GOF_ABLATE_TIMED"]
    t0 = time.perf_counter()
    for t in range(W, W + K):
        step(t)
    pipe.synchronize()
    torch.cuda.synchronize()
    gd.barrier()
    elapsed = gd.max_over_ranks(time.perf_counter() - t0, device=red_dev)

    det_ms = morph_ms = 0.0
    n_meas = 0
    if not args.no_events:
        det_ms, morph_ms, n_meas = pipe.profile_query()
        pipe.profile(False)
    mbar_end = pipe.mean_modes()
    mbar = 0.5 * (mbar_start + mbar_end)
    fg = float((masks != 0).float().mean().item())

    # outside the timed region: the same kernel in its one-frame-per-launch form
    # (what GOFPID.detect() runs, and the HBM-bound form of the kernel), for the record
    single = None
    if T > 1 and not args.no_events:
        n1 = 32
        pipe.profile(True, n1)
        for t in range(n1):
            pipe.apply_device(frames[t % P].data_ptr(), masks.data_ptr(), n_frames=1)
        pipe.synchronize()
        d1, m1, c1 = pipe.profile_query()
        pipe.profile(False)
        if c1:
            _, b1 = pipe.algorithmic_bytes(pipe.mean_modes(), 1)
            s1 = d1 / c1 * 1e-3
            single = {"avg_kernel_us": round(s1 * 1e6, 2), "achieved": round(b1 / s1 / 1e9, 1),
                      "frac": round(b1 / s1 / 1e9 / HBM_PEAK_GBS, 4),
                      "algorithmic_bytes_per_launch": int(b1), "launches_timed": c1,
                      "morph_kernel_us": round(m1 / c1 * 1e3, 2)}

    if rank == 0:
        total_frames = S * T * K * world
        out = {
            "metric": "frames/sec at 1080p MOG2+5x5 morph",
            "value": round(total_frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d BGR uint8, GaussianBlur 3x3 (sigma 4) -> MOG2 (cv2 defaults, "
                            "shadows on) -> %s; %d streams per GPU, %d "
                            "consecutive frame(s) per stream per step; frames resident in HBM"
                            % (w, h, " -> ".join("%s %dx%d" % (o, k.shape[0], k.shape[1]) for o, k in morph)
                               or "no morphology", S, T),
                "streams_per_gpu": S, "frames_per_stream_per_step": T,
                "frames_per_step": S * T * world, "height": h, "width": w,
                "channels": 3, "mean_modes": round(mbar, 4), "foreground_fraction": round(fg, 5),
                "parallelism": "streams sharded %d per GPU, no data-path collective" % S,
            },
        }
        if n_meas:
            # per-unit figure B_px = C + 1 + 2(1 + 4(2+C) m) bytes per
            # pixel-frame (model read + written once per frame) x the pixel-frames
            # one launch processes.  With T frames per launch the kernel moves the
            # model once per launch, so it can exceed this single-frame roofline;
            # its own byte count is reported next to it.
            _, per_frame_b = pipe.algorithmic_bytes(mbar, 1)
            total_b, launch_b = pipe.algorithmic_bytes(mbar, T)
            det_s = det_ms / n_meas * 1e-3
            unit_b = per_frame_b * T
            achieved = unit_b / det_s / 1e9
            key = {"streams": S, "height": h, "width": w, "frames_per_step": T}
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "%s (fused 3x3 blur + MOG2 update + classify, %d frame(s) per launch)"
                          % ("mog2_frames_kernel<3>" if T > 1 else "mog2_kernel<3,true,false,true>",
                             min(T, 8)),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": load_traffic(args.traffic_file, key),
                "algorithmic_bytes_per_launch": int(unit_b),
                "bytes_per_pixel_frame": round(per_frame_b / (S * h * w), 3),
                "pixel_frames_per_launch": S * T * h * w,
                "avg_kernel_us": round(det_s * 1e6, 2),
                "launches_timed": n_meas,
                "temporal_blocking": {
                    "frames_per_launch": T,
                    "launch_bytes_model_once": int(launch_b),
                    "achieved_on_launch_bytes_GBps": round(launch_b / det_s / 1e9, 1),
                    "note": "T frames per launch keep the per-pixel model in registers; 'achieved' "
                            "prices every frame at the single-frame algorithmic bytes (SURVEY 8d), "
                            "so frac > 1 means the kernel beats the one-frame-per-pass HBM roofline",
                } if T > 1 else None,
                "per_frame_launch": single,
                "morph_kernel_us": round(morph_ms / n_meas * 1e3, 2),
                "whole_step_GBps": round(total_b / (elapsed / K) / 1e9, 1),
            }
        if args.cpu_seconds > 0 and world == 1:   # CPU leg: rank 0 at N = 1 only
            host = [frames[i, 0].cpu().numpy() for i in range(P)]
            out["cpu_baseline"] = cpu_baseline(host, h, w, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        gd.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
