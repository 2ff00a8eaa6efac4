#!/bin/bash
# The BASELINE.json configurations other than the headline one, same code.
# hbm frac = the launch's algorithmic bytes / kernel time / 8 TB/s.
set -e
mkdir -p gpurun_out
run() { # label, bench args
  label=$1; shift
  python bench.py --cpu-seconds 0 --no-extra --steps 64 --rounds 5 "$@" > gpurun_out/cfg.log 2> gpurun_out/cfg.err || { tail -5 gpurun_out/cfg.err; exit 1; }
  python - "$label" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/cfg.log") if l.startswith("{")][-1])
r = d["roofline"]
hbm = r["frac"] if r["bound"] == "hbm" else r["hbm"]["frac"]
print("%-46s %10.0f fps  step %8.1f us  mog2 %8.1f us (hbm frac %.3f)  morph %6.1f us  mbar %.3f"
      % (sys.argv[1], d["value"], d["ms_per_step"] * 1e3, r["avg_kernel_us"], hbm, r["morph_kernel_us"],
         d["config"]["mean_modes"]))
PY
}
run "0: 320x240, 1 stream, T=1"      --height 240 --width 320 --streams 1 --frames-per-step 1
run "0: 320x240, 64 streams, T=8"    --height 240 --width 320 --streams 64 --frames-per-step 8
run "1: 1080p, 1 stream, T=1"        --streams 1 --frames-per-step 1
run "1: 1080p, 1 stream, T=8"        --streams 1 --frames-per-step 8
run "1: 1080p, 8 streams, T=1"       --frames-per-step 1
run "2: 4K 7x7 open/close, 2 streams, T=1" --height 2160 --width 3840 --streams 2 --frames-per-step 1 --pool 64 --morph erode7,dilate7,dilate7,erode7
run "2: 4K 7x7 open/close, 2 streams, T=8" --height 2160 --width 3840 --streams 2 --frames-per-step 8 --pool 64 --morph erode7,dilate7,dilate7,erode7
run "4: 720x576, 33 streams, T=8"    --height 576 --width 720 --streams 33 --frames-per-step 8
