#!/bin/bash
# BASELINE config 3's "LDS tile-size sweep": rows per staged tile of the per-frame kernel
# (64 px wide, one wave per row: 2, 4 or 8 rows = 128, 256, 512 threads) x {1080p x 8 streams,
# 2160p x 2 streams, 7x7 open/close}, one frame per launch, benchmark scene.  Output: a table.
mkdir -p gpurun_out/tile_sweep
printf "%-34s %5s %10s %9s %8s %8s %10s\n" "configuration" "rows" "kernel us" "GB/s" "frac" "m-bar" "frames/s"
for CFG in "1080p x 8 streams, 5x5 erode/dilate|--height 1080 --width 1920 --streams 8" \
           "2160p x 2 streams, 7x7 open/close|--height 2160 --width 3840 --streams 2 --pool 64 --morph erode7,dilate7,dilate7,erode7"; do
  NAME=${CFG%%|*}; ARGS=${CFG##*|}
  for R in 2 4 8; do
    GOF_T1_TILE_ROWS=$R python bench.py --frames-per-step 1 --no-extra --cpu-seconds 0 --steps 128 --rounds 5 $ARGS \
        > gpurun_out/tile_sweep/out.json 2> gpurun_out/tile_sweep/err.log || { tail -3 gpurun_out/tile_sweep/err.log; exit 1; }
    python - "$NAME" $R <<'PY'
import json, sys
d = json.load(open("gpurun_out/tile_sweep/out.json"))
r = d["roofline"]
print("%-34s %5s %10.1f %9.0f %8.4f %8.3f %10.0f" % (sys.argv[1], sys.argv[2], r["avg_kernel_us"], r["achieved"], r["frac"],
                                                r["mean_modes"], d["value"]))
PY
  done
done
