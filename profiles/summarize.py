"""Condenses rocprofv3 CSV output (kernel trace + PMC passes) into a short text
summary and a traffic JSON that bench.py can cite.  usage: summarize.py <dir> [frames_per_step]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def short(name):
    name = name.replace("gofk::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def main(d, fps=8):
    # kernel trace: durations per kernel
    dur = defaultdict(list)
    for r in rows(os.path.join(d, "trace", "**", "*kernel_trace.csv")):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== kernel trace (ns) ==")
    tot = sum(sum(v) for v in dur.values()) or 1
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print("%-70s calls %5d  avg %10.1f  med %10.1f  min %10d  max %10d  %5.1f%%"
              % (k, len(v), sum(v) / len(v), v2[len(v) // 2], v2[0], v2[-1], 100.0 * sum(v) / tot))
    print("== timed region only (the last 256 dispatches = bench.py's K steps) ==")
    for k, v in dur.items():
        if ("mog2" in k or "morph" in k) and len(v) >= 256:
            t = v[-256:]
            print("%-70s avg %10.1f ns over %d dispatches" % (k, sum(t) / len(t), len(t)))
    # PMC: average per dispatch per kernel
    pmc = defaultdict(lambda: defaultdict(list))
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for r in rows(os.path.join(sub, "**", "*counter_collection.csv")):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n== PMC (mean per dispatch) ==")
    out = {}
    for k, cs in pmc.items():
        if "mog2" not in k and "morph" not in k:
            continue
        print(k)
        for c, v in sorted(cs.items()):
            # skip warm-up dispatches: use the last half
            v = v[len(v) // 2:]
            m = sum(v) / len(v)
            out.setdefault(k, {})[c] = m
            print("    %-28s %16.1f   (n=%d)" % (c, m, len(v)))
    for k, cs in out.items():
        if "mog2" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            # MI355X_MICROARCH.md HBM: counters are in KiB; on gfx950 FETCH_SIZE
            # reports 1/2 of the bytes of a wide coalesced streaming read -> x2.
            fetch_raw = cs["FETCH_SIZE"] * 1024.0
            write = cs["WRITE_SIZE"] * 1024.0
            rec = {"config": {"streams": 8, "height": 1080, "width": 1920, "frames_per_step": fps},
                   "kernel": k, "fetch_size_raw_bytes": fetch_raw, "fetch_bytes_corrected": 2 * fetch_raw,
                   "write_bytes": write, "hbm_bytes_per_launch": 2 * fetch_raw + write,
                   "note": "per launch, mean of the last half of the dispatches; FETCH_SIZE x2 per "
                           "MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B), factor "
                           "re-calibrated for 4-B/lane streams in profiles/r01_counter_calibration.txt"}
            print("\ntraffic:", json.dumps(rec))
            with open(os.path.join(d, "traffic.json"), "w") as f:
                json.dump(rec, f)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
