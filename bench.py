#!/usr/bin/env python3
"""Headline benchmark: frames/s of the per-frame foreground path at 1080p.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1920x1080 BGR frames, GaussianBlur 3x3
(sigma 4, as the reference calls it) -> MOG2 -> erode 5x5 -> dilate 5x5, i.e.
steps 1-3 of GOFPID.detect (pygofpid/pipeline.py:461-476).  One "step" is one
pass of the path over one batch: `--frames-per-step` consecutive frames of each
of `--streams` independent video streams (8 x 8 by default; the background
models, ~0.5-1.7 GB in use, do not fit the 256 MiB Infinity Cache).  Frames are
synthetic and already resident in HBM when the timed region starts.

Protocol:
  1. pre-roll: real steps for >= --preroll-seconds (0.5 s) AND >= 256 frames per
     stream, whatever --warmup says: the learning rate 1/min(2t, history) has
     reached 1/history and the clocks have settled under load;
  2. W further untimed warm-up steps;
  3. R (--rounds, 10) rounds of EXACTLY K steps, each bracketed by a barrier +
     device synchronisation on both sides and reduced with MAX over ranks;
     the MEDIAN round is reported (`ms_per_step`, `value`), min / max alongside;
  4. two more rounds of K steps with HIP events around every kernel, on the
     launch stream, for the roofline block (kept out of the timed rounds: an
     event pair costs the stream a few microseconds).

N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and
the MAX over ranks); every rank steps its own streams -- the path shards by
stream with no data-path collective ("weak" scaling).

Prints ONE JSON line on rank 0, with
  roofline      -- the dominant kernel (fused blur + MOG2) against HBM: achieved = the launch's
                   ALGORITHMIC bytes (frame in, mask out, live model read + written once
                   per launch) / the kernel's average duration, measured here with HIP events on
                   the launch stream (raw event time; the marker-corrected value sits beside it);
                   peak 8 TB/s.  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC
                   passes of this same command (profiles/pmc_latest.json), `valu_issue` = what the
                   8-frame launch is co-limited by (SQ_INSTS_VALU / t against 1024 SIMDs x 2.4 GHz /
                   2 cycles).  Both come from a PMC record that carries the hash of the native
                   sources it was taken from (pygofpid_amd/buildid.py): if it differs from this
                   tree, `"pmc_stale": true` and everything derived from counters is null.
  per_frame     -- the same path with ONE frame per launch (what GOFPID.detect()
                   runs), whole-step timed by the same protocol + its kernel record
  detect_path   -- steps 1-3 + the GPU blob extraction of step 4 on every mask of a step
  detect_host   -- GOFPID.detect() / detect_many() on host frames (PCIe-inclusive; never `value`)
  worst_case    -- reference-test style input (i.i.d. uniform uint8 frames,
                   /root/reference/tests/test_pipeline.py:98-109): every pixel carries
                   all 5 modes (206 B/px/frame), nothing is ever "settled"
  cpu_baseline  -- the CPU oracle ("port" of the OpenCV algorithms, NOT OpenCV) timed on this
                   host's cores on a bounded sample, started from the model state the GPU leg
                   runs at (so both legs see the same mean mode count)
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the CPU leg is a throughput run: let the OpenMP workers spin between the
# per-frame parallel regions (read by libgomp when oracle/ is loaded)
os.environ.setdefault("OMP_WAIT_POLICY", "active")

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD = 1024            # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9         # max shader clock
VALU_ISSUE_CYCLES = 2    # a SIMD-32 issues one wave64 vector instruction per 2 cycles at best
                         # (measured per opcode: 2.4-4.9, profiles/r01_valu_issue_rates.txt)
VALU_PEAK_GINST = N_SIMD * CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9
MIN_PREROLL_FRAMES = 256


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


def make_noise_pool(torch, streams, pool, h, w, device, seed):
    """Reference-test style frames: np.random.randint(0, high=255) uint8, i.i.d."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 255, (pool, streams, h, w, 3), generator=g, device=device,
                         dtype=torch.uint8)


def cpu_baseline(frames_host, h, w, budget_s, state=None, nframes=0, start=0):
    """Time the CPU oracle on this host (checker reused as the baseline leg).
    The thread count is the best of a short sweep up to the CPUs this process
    may use; `cores` reports the count actually used for the timed sample.
    state / nframes: the trained model (SoA planes of ONE stream) and frame count the
    run starts from -- the state of the GPU leg, so that both see the same clip state;
    start: index of the next frame of the pool."""
    from oracle import cpu as oracle
    oracle.build()
    n = len(frames_host)
    budget = oracle.cpu_budget()
    cands = sorted({t for t in (8, 16, 32, 64, 96, 128, budget) if t <= budget} or {1})

    def rate(threads, seconds, warm):
        p = oracle.Pipeline(h, w, 3, threads=threads)
        if state is not None:
            p.import_state(state, nframes)
        k = start
        for _ in range(warm):
            p.apply(frames_host[k % n])
            k += 1
        m0 = p.mean_modes()
        done, t0 = 0, time.perf_counter()
        while True:
            p.apply(frames_host[k % n])
            k += 1
            done += 1
            el = time.perf_counter() - t0
            if el >= seconds or done >= 4096:
                return done / el, done, el, p, m0

    best = max(cands, key=lambda t: rate(t, 0.6, 4)[0]) if len(cands) > 1 else cands[0]
    fps, done, el, p, m0 = rate(best, budget_s, 8)
    return {"value": round(fps, 2), "unit": "frames/s", "cores": best, "kind": "port",
            "sample": "%d frames of 1 stream %dx%d BGR (the %d-frame pool, continued from the model state "
                      "and pool position of the GPU leg) in %.1f s, OpenMP row bands over %d threads "
                      "(best of %s on a %d-CPU budget); OpenCV-algorithm CPU restatement "
                      "(oracle/gof_oracle.c), not OpenCV" % (done, w, h, n, el, best, cands, budget),
            "mean_modes": round(0.5 * (m0 + p.mean_modes()), 4)}


def load_pmc(path, key):
    """The committed PMC record (profiles/) of this configuration, or {}: HBM bytes
    per launch (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction) and the SQ
    instruction counts per launch.  Counters cannot be read inside this process;
    they come from separate `rocprofv3 --pmc` passes of this same command.
    -> (record, stale): stale when the record was taken from other native sources
    than this tree's (pygofpid_amd/buildid.py) -- nothing may be derived from it then."""
    from pygofpid_amd.buildid import build_id
    try:
        with open(path) as f:
            recs = json.load(f)
    except Exception:
        return {}, True
    for rec in recs if isinstance(recs, list) else [recs]:
        if rec.get("config") == key:
            return rec, rec.get("build_id") != build_id()
    return {}, True


class Runner:
    """One pipeline + its frame pool, stepped by the protocol in the module docstring."""

    def __init__(self, torch, gd, pipe, frames, masks, T, red_dev):
        self.torch, self.gd, self.pipe = torch, gd, pipe
        self.frames, self.masks, self.T, self.P = frames, masks, T, frames.shape[0]
        self.red_dev = red_dev
        self.i = 0
        self.pre = None       # step index -> device pointer of the masks to write (default: self.masks)
        self.post = None      # (step index, T): enqueued after every step (the detect-path block's extraction)

    def step(self, T=None):
        T = T or self.T
        base = (self.i * T) % self.P
        if base + T > self.P:
            base = 0
        mptr = self.pre(self.i) if self.pre else self.masks.data_ptr()
        self.pipe.apply_device(self.frames[base].data_ptr(), mptr, n_frames=T)
        if self.post:
            self.post(self.i, T)
        self.i += 1

    def sync(self):
        self.pipe.synchronize()
        self.torch.cuda.synchronize()

    def preroll(self, seconds, min_frames, T=None):
        """Real steps until both the time and the frame count are reached (every rank
        the same number of steps: the count is fixed by rank 0's clock in units of 16)."""
        T = T or self.T
        done, t0 = 0, time.perf_counter()
        while True:
            for _ in range(16):
                self.step(T)
            done += 16
            self.sync()
            go = 1.0 if (time.perf_counter() - t0 < seconds or done * T < min_frames) else 0.0
            if self.gd.max_over_ranks(go, device=self.red_dev) == 0.0:
                return done

    def rounds(self, K, R, T=None):
        """R rounds of exactly K steps -> list of per-round elapsed seconds (MAX over ranks)."""
        out = []
        for _ in range(R):
            self.sync()
            self.gd.barrier()
            t0 = time.perf_counter()
            for _ in range(K):
                self.step(T)
            self.sync()
            out.append(self.gd.max_over_ranks(time.perf_counter() - t0, device=self.red_dev))
        return out

    def kernel_times(self, K, step_us, T=None):
        """HIP events around the detector and the morphology kernel of K steps, on the
        launch stream.  The events are created beforehand and 32 un-bracketed steps run
        first, without a host synchronisation in between (an idle gap costs the first
        ~30 launches their clocks, profiles/r02_clock_ramp.json).  An event marker
        serialises the stream: three per step lengthen the step by 3 d, one d inside each
        bracket; d = (bracketed period - un-bracketed period `step_us`) / 3 is taken off
        both kernels' raw means.  -> dict or None."""
        self.pipe.profile(True, K, stride=1)      # creates the events
        self.pipe.profile(False)
        self.sync()
        for _ in range(32):
            self.step(T)
        self.pipe.profile(True, K, stride=1)
        for _ in range(K):
            self.step(T)
        self.sync()
        span_us = self.pipe.profile_span() * 1e3
        det, morph = self.pipe.profile_read(K)
        self.pipe.profile(False)
        n = len(det)
        if not n:
            return None
        det_raw, morph_raw = float(det.mean()) * 1e3, float(morph.mean()) * 1e3
        # span covers n brackets and n - 1 gaps
        period = (span_us + (span_us - n * (det_raw + morph_raw)) / max(1, n - 1)) / n
        d = max(0.0, (period - step_us) / 3.0)
        return {"det_us": det_raw - d, "morph_us": morph_raw - d, "n": n, "det_raw_us": det_raw,
                "morph_raw_us": morph_raw, "marker_us": d,
                "det_median_raw_us": float(statistics.median(det.tolist())) * 1e3}


def summarize_rounds(el, K, frames_per_step):
    med = statistics.median(el)
    return {"ms_per_step": round(med / K * 1e3, 4),
            "frames_per_s": round(frames_per_step * K / med, 1),
            "rounds": len(el),
            "round_ms": {"median": round(med * 1e3, 3), "min": round(min(el) * 1e3, 3),
                         "max": round(max(el) * 1e3, 3)}}


def kernel_record(pipe, mbar, T, kt, pmc_and_stale, kernel_name):
    """Roofline statement of one detector launch of T frames per stream (kt: Runner.kernel_times).
    Primary time = the raw HIP-event mean; the marker-corrected mean is reported beside it."""
    from pygofpid_amd.buildid import build_id
    pmc, stale = pmc_and_stale
    if stale:
        pmc = dict(pmc, hbm_bytes_per_launch=None, sq_insts_valu=None)
    _, launch_b = pipe.algorithmic_bytes(mbar, T)
    det_us, morph_us = kt["det_raw_us"], kt["morph_raw_us"]
    t = det_us * 1e-6
    hbm = launch_b / t / 1e9
    traffic = pmc.get("hbm_bytes_per_launch")
    rec = {"kernel": kernel_name, "avg_kernel_us": round(det_us, 2), "launches_timed": kt["n"],
           "morph_kernel_us": round(morph_us, 2),
           "event_timing": {"what": "avg_kernel_us / morph_kernel_us are raw HIP-event means on the launch stream; "
                                    "an event marker serialises the stream, the corrected values take the measured "
                                    "marker cost off",
                            "marker_corrected_kernel_us": round(kt["det_us"], 2),
                            "marker_corrected_morph_us": round(kt["morph_us"], 2),
                            "raw_median_kernel_us": round(kt["det_median_raw_us"], 2),
                            "marker_overhead_us": round(kt["marker_us"], 2)},
           "mean_modes": round(mbar, 4),
           "algorithmic_bytes_per_launch": int(launch_b),
           "bound": "hbm", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(hbm / HBM_PEAK_GBS, 4),
           "traffic": traffic,
           "traffic_frac": round(traffic / t / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
           "build_id": build_id(), "pmc_build_id": pmc.get("build_id"), "pmc_stale": bool(stale),
           "pmc_source": pmc.get("source")}
    if T > 1:
        insts = pmc.get("sq_insts_valu")
        rec["valu_issue"] = {
            "what": "the multi-frame launch is co-limited by vector-instruction issue: SQ_INSTS_VALU per launch "
                    "(PMC pass) / t against %d SIMDs x %.1f GHz / %d cycles per wave64 instruction"
                    % (N_SIMD, CLOCK_HZ / 1e9, VALU_ISSUE_CYCLES),
            "valu_insts_per_launch": insts,
            "achieved": round(insts / t / 1e9, 1) if insts else None,
            "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instr/s",
            "frac": round(insts / t / 1e9 / VALU_PEAK_GINST, 4) if insts else None,
            "cycles_per_valu_inst": round(N_SIMD * CLOCK_HZ * t / insts, 2) if insts else None}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=10, help="timed rounds of --steps steps; the median is reported")
    ap.add_argument("--preroll-seconds", type=float, default=0.5)
    ap.add_argument("--streams", type=int, default=8, help="video streams stepped per GPU")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--pool", type=int, default=256, help="distinct synthetic frames per stream")
    ap.add_argument("--frames-per-step", type=int, default=8,
                    help="consecutive frames of every stream handed over per step (temporal "
                         "blocking: the MOG2 model crosses HBM once per 8 frames)")
    ap.add_argument("--objects", type=int, default=6, help="moving rectangles per stream")
    ap.add_argument("--morph", default="erode5,dilate5",
                    help="mat_morph list, e.g. erode7,dilate7,dilate7,erode7 (BASELINE config 3)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU leg")
    ap.add_argument("--pmc-file", default=os.path.join(ROOT, "profiles", "pmc_latest.json"))
    ap.add_argument("--no-events", action="store_true",
                    help="no HIP-event rounds (roofline blocks omitted); what the rocprofv3 passes run")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-frame and worst-case blocks")
    ap.add_argument("--noise", action="store_true", help="main workload on the worst-case input instead")
    args = ap.parse_args()

    # Everything but the final JSON line goes to stderr: RCCL prints a version banner on
    # stdout when its first communicator is created, libraries may print warnings.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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

    import numpy as np

    from pygofpid_amd import _lib
    from pygofpid_amd.foreground import ForegroundPipeline

    h, w, S, K, W, R = args.height, args.width, args.streams, args.steps, args.warmup, max(1, args.rounds)
    T = args.frames_per_step
    if T < 1 or args.pool % T:
        raise SystemExit("--frames-per-step must divide --pool")
    morph = []
    for item in [m for m in args.morph.split(",") if m]:
        op = "erode" if item.startswith("erode") else "dilate"
        kk = int(item[len(op):])
        morph.append((op, np.ones((kk, kk), np.uint8)))
    ctx = _lib.Context(dev_index)
    if args.noise:
        frames = make_noise_pool(torch, S, args.pool, h, w, device, seed=17 + rank)
    else:
        frames = make_pool(torch, S, args.pool, h, w, device, seed=17 + rank, n_objects=args.objects)
    masks = torch.empty((max(T, 8), S, h, w), dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    pipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, morph=morph)
    run = Runner(torch, gd, pipe, frames, masks, T, red_dev)

    # 1-3: pre-roll, warm-up, timed rounds
    pre_steps = run.preroll(args.preroll_seconds, MIN_PREROLL_FRAMES)
    for _ in range(W):
        run.step()
    mbar_start = pipe.mean_modes()
    if os.environ.get("GOF_ABLATE_TIMED"):          # profiles/ ablations: flip after the warm-up
        os.environ["GOF_ABLATE"] = os.environ["GOF_ABLATE_TIMED"]
    el = run.rounds(K, R)
    main_sum = summarize_rounds(el, K, S * T * world)
    mbar = 0.5 * (mbar_start + pipe.mean_modes())
    fg = float((masks[:T] != 0).float().mean().item())
    # the CPU leg starts from here: model of stream 0, its frame count, the pool position
    cpu_state = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        run.sync()
        st = pipe.mog2_state()
        npx = h * w
        cpu_state = ({"weight": st["weight"][:, :npx].copy(), "var": st["var"][:, :npx].copy(),
                      "mean": st["mean"][:, :, :npx].copy(), "modes": st["modes"][:npx].copy()},
                     int(st["nframes"]), (run.i * T) % frames.shape[0])
        del st

    # 4: kernel times, outside the timed rounds
    kt = None
    if not args.no_events:
        kt = run.kernel_times(max(32, min(2 * K, 512)), main_sum["ms_per_step"] * 1e3)

    morph_txt = " -> ".join("%s %dx%d" % (o, k.shape[0], k.shape[1]) for o, k in morph) or "no morphology"
    scene = ("i.i.d. uniform uint8 frames (reference-test style worst case)" if args.noise else
             "synthetic surveillance scene, %d moving objects with shadows per stream, %d-frame pool"
             % (args.objects, args.pool))
    out = None
    if rank == 0:
        out = {
            "metric": "frames/sec at 1080p MOG2+5x5 morph",
            "value": main_sum["frames_per_s"],
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": main_sum["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d BGR uint8, GaussianBlur 3x3 (sigma 4) -> MOG2 (cv2 defaults, "
                            "shadows on) -> %s; %d streams per GPU, %d consecutive frame(s) per "
                            "stream per step; %s; frames resident in HBM"
                            % (w, h, morph_txt, S, T, scene),
                "streams_per_gpu": S, "frames_per_stream_per_step": T,
                "frames_per_step": S * T * world, "height": h, "width": w,
                "channels": 3, "mean_modes": round(mbar, 4), "foreground_fraction": round(fg, 5),
                "parallelism": "streams sharded %d per GPU, no data-path collective" % S,
            },
            "protocol": {"preroll_steps": pre_steps, "preroll_frames_per_stream": pre_steps * T,
                         "rounds": R, "round_ms": main_sum["round_ms"],
                         "statistic": "median round of K steps; MAX over ranks per round"},
        }
        if kt:
            key = {"streams": S, "height": h, "width": w, "frames_per_step": T,
                   "input": "noise" if args.noise else "scene"}
            name = ("mog2_frames_kernel<3> (fused 3x3 blur + MOG2 update + classify, %d frames per launch)"
                    % min(T, 8)) if T > 1 else \
                "mog2_kernel<3,true,false,true> (fused 3x3 blur + MOG2 update + classify, 1 frame per launch)"
            roof = kernel_record(pipe, mbar, T, kt, load_pmc(args.pmc_file, key), name)
            _, per_frame_b = pipe.algorithmic_bytes(mbar, 1)
            det_us = kt["det_us"]
            roof["kernels_us_per_step"] = round(kt["det_us"] + kt["morph_us"], 2)   # (marker-corrected)
            roof["step_us"] = round(main_sum["ms_per_step"] * 1e3, 2)
            if T > 1:
                # NOT a roofline: what one-frame-per-pass traffic (6 + 40 m B/px/frame)
                # would have needed at the HBM peak, over what the launch took
                roof["speedup_vs_one_frame_per_pass_at_hbm_peak"] = round(
                    per_frame_b * T / (HBM_PEAK_GBS * 1e9) / (det_us * 1e-6), 3)
            out["roofline"] = roof

    # the drop-in form: ONE frame per launch, same pipeline (model trained by the above)
    if not args.no_extra and T > 1 and not args.no_events:
        run.preroll(0.1, 32, T=1)
        el1 = run.rounds(K, R, T=1)
        s1 = summarize_rounds(el1, K, S * world)
        kt1 = run.kernel_times(max(32, min(2 * K, 256)), s1["ms_per_step"] * 1e3, T=1)
        if rank == 0 and kt1:
            mb1 = pipe.mean_modes()
            key = {"streams": S, "height": h, "width": w, "frames_per_step": 1,
                   "input": "noise" if args.noise else "scene"}
            rec = kernel_record(pipe, mb1, 1, kt1, load_pmc(args.pmc_file, key),
                                "mog2_kernel<3,true,false,true> (1 frame per launch: what GOFPID.detect() runs)")
            rec.update({"value": s1["frames_per_s"], "unit_value": "frames/s",
                        "ms_per_step": s1["ms_per_step"], "round_ms": s1["round_ms"],
                        "frames_per_step": S * world})
            out["per_frame"] = rec

    # steps 1-3 + step 4's front end on the device (contours, area filter, blob moments of every
    # mask; pygofpid/helpers.py:125-168, pipeline.py:555-572): what a batch caller of detect() runs
    if not args.no_extra and not args.noise and world == 1 and not args.no_events:
        from pygofpid_amd.blobs import BlobExtractor
        ex = BlobExtractor(T * S, h, w, ctx=ctx)
        # The extraction is latency-bound (union-find, list ranking), the detector launch
        # issue-bound: they overlap.  Masks are double-buffered; the extraction of step i runs
        # on a second HIP stream while the pipeline's stream computes step i + 1.
        masks2 = [masks[:T], torch.empty_like(masks[:T])]
        s_pipe = torch.cuda.ExternalStream(ctx.stream, device=device)
        s_blob = torch.cuda.Stream(device=device, priority=-1)   # its small grids must not queue behind 64 800 workgroups
        done = [torch.cuda.Event(), torch.cuda.Event()]      # masks2[k] written
        free = [torch.cuda.Event(), torch.cuda.Event()]      # masks2[k] read by its extraction
        for e in free:
            e.record(s_blob)

        def pre(i):
            s_pipe.wait_event(free[i & 1])
            return masks2[i & 1].data_ptr()

        def post(i, t):
            k = i & 1
            done[k].record(s_pipe)
            s_blob.wait_event(done[k])
            ex.extract_device(masks2[k].data_ptr(), t * S, 100.0, stream=s_blob.cuda_stream)
            free[k].record(s_blob)

        run.pre, run.post = pre, post
        run.preroll(0.1, 32)
        elb = run.rounds(K, R)
        s_blob.synchronize()
        sb = summarize_rounds(elb, K, S * T)
        res = [ex.result(i) for i in range(T * S)]
        run.pre = run.post = None
        ok = [r for r in res if r is not None]
        out["detect_path"] = {
            "what": "steps 1-3 + GPU blob extraction (external contours, cv.contourArea >= 100, moments, "
                    "bounding boxes, ordered contour points) on all %d masks of a step, on a second HIP stream "
                    "beside the next step's kernels (double-buffered masks); only blob records and contour "
                    "points are copied to the host" % (T * S),
            "value": sb["frames_per_s"], "unit_value": "frames/s", "ms_per_step": sb["ms_per_step"],
            "round_ms": sb["round_ms"],
            "blobs_per_mask": round(sum(len(r[0]) for r in ok) / max(1, len(ok)), 2),
            "contour_points_per_mask": round(sum(len(r[1]) for r in ok) / max(1, len(ok)), 1),
            "bytes_to_host_per_mask": round(sum(16 + r[0].nbytes + 4 * len(r[1]) for r in ok) / max(1, len(ok)), 1),
            "masks_without_result": len(res) - len(ok),
        }
        del ex

    # the drop-in API on host frames: GOFPID.detect() / detect_many() (pygofpid/pipeline.py:434-493),
    # PCIe-inclusive -- reported beside the kernels, never as `value`
    if not args.no_extra and not args.noise and world == 1 and not args.no_events:
        from pygofpid_amd.pipeline import GOFPID
        host = [frames[i, 0].cpu().numpy() for i in range(frames.shape[0])]
        nb = host[0].nbytes
        # what the link gives for one frame from the same (pageable) buffers
        dst = torch.empty(nb, dtype=torch.uint8, device=device)
        srcs = [torch.from_numpy(f.reshape(-1)) for f in host[:16]]
        for k in range(8):
            dst.copy_(srcs[k], non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(64):
            dst.copy_(srcs[k % 16], non_blocking=True)
        torch.cuda.synchronize()
        h2d = nb * 64 / (time.perf_counter() - t0) / 1e9
        g = GOFPID().initialize()
        g.detect_many(host)                                  # trains the model on one pass of the pool
        n1, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            g.detect(host[n1 % len(host)])
            n1 += 1
        r1 = n1 / (time.perf_counter() - t0)
        nm, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.5:
            g.detect_many(host[:64] if nm % 128 == 0 else host[64:128])
            nm += 64
        rm = nm / (time.perf_counter() - t0)
        out["detect_host"] = {
            "what": "GOFPID on host frames (one 1080p BGR stream, pageable numpy buffers): upload, steps 1-3 and the "
                    "blob extraction on the GPU, blob records down, steps 5-7 on the host (Python); detect_many "
                    "pipelines batches of 8 frames (upload of batch k + 1 beside the kernels of batch k and the host "
                    "steps of batch k)",
            "detect_frames_per_s": round(r1, 1), "detect_many_frames_per_s": round(rm, 1),
            "bytes_up_per_frame": nb,
            "h2d_GBps_measured": round(h2d, 1), "h2d_bound_frames_per_s": round(h2d * 1e9 / nb, 1),
            "detect_many_h2d_GBps": round(rm * nb / 1e9, 1),
            "detect_many_frac_of_h2d": round(rm * nb / 1e9 / h2d, 3),
            "detect_frac_of_h2d": round(r1 * nb / 1e9 / h2d, 3)}
        del g, host, srcs, dst

    # worst case: uniform-noise input, every pixel carries all five modes
    if not args.no_extra and not args.noise and world == 1 and not args.no_events:
        del run, pipe
        npool = 64
        nframes = make_noise_pool(torch, S, npool, h, w, device, seed=99)
        npipe = ForegroundPipeline((h, w, 3), streams=S, ctx=ctx, morph=morph)
        nrun = Runner(torch, gd, npipe, nframes, masks, T, red_dev)
        nrun.preroll(0.3, MIN_PREROLL_FRAMES)
        Kn = max(4, min(K, 64))
        eln = nrun.rounds(Kn, R)
        sn = summarize_rounds(eln, Kn, S * T)
        ktn = nrun.kernel_times(max(32, Kn), sn["ms_per_step"] * 1e3)
        mbn = npipe.mean_modes()
        key = {"streams": S, "height": h, "width": w, "frames_per_step": T, "input": "noise"}
        rec = kernel_record(npipe, mbn, T, ktn, load_pmc(args.pmc_file, key),
                            "mog2_frames_kernel<3>, general five-mode loop" if T > 1 else "mog2_kernel<3,true,false,true>")
        rec.update({"input": "i.i.d. uniform uint8 frames, randint(0, 255) "
                             "(/root/reference/tests/test_pipeline.py:98-109 style), %d-frame pool" % npool,
                    "value": sn["frames_per_s"], "unit_value": "frames/s", "steps": Kn,
                    "ms_per_step": sn["ms_per_step"], "round_ms": sn["round_ms"],
                    "foreground_fraction": round(float((masks[:T] != 0).float().mean().item()), 5)})
        if T > 1:
            nrun.preroll(0.05, 16, T=1)
            eln1 = nrun.rounds(Kn, 3, T=1)
            kn1 = nrun.kernel_times(32, summarize_rounds(eln1, Kn, S)["ms_per_step"] * 1e3, T=1)
            key1 = dict(key, frames_per_step=1)
            rec["per_frame"] = kernel_record(npipe, npipe.mean_modes(), 1, kn1,
                                             load_pmc(args.pmc_file, key1), "mog2_kernel<3,true,false,true>")
        out["worst_case"] = rec
        del nrun, npipe, nframes

    if rank == 0:
        if args.cpu_seconds > 0 and world == 1 and cpu_state is not None:   # CPU leg: rank 0 at N = 1 only
            host = [frames[i, 0].cpu().numpy() for i in range(frames.shape[0])]
            out["cpu_baseline"] = cpu_baseline(host, h, w, args.cpu_seconds, *cpu_state)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        gd.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
