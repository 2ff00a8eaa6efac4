"""Condenses rocprofv3 CSV output (kernel trace + PMC passes) of one bench.py
configuration into a short text summary and a PMC record that bench.py cites
(profiles/pmc_latest.json holds one record per configuration).

    summarize.py <dir> <frames_per_step> <scene|noise> <timed dispatches> [streams h w]

<dir> holds trace/ (--kernel-trace) and pmc_*/ (one --pmc pass each).  The timed
region of a `bench.py --no-events --no-extra` run is its LAST rounds x steps
dispatches of the detector kernel.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygofpid_amd.buildid import build_id  # noqa: E402  (hash of the native sources this run was built from)


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def short(name):
    name = name.replace("gofk::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def main(d, fps, inp, timed, streams=8, h=1080, w=1920):
    # kernel trace: durations per kernel, in dispatch order
    dur = defaultdict(list)
    trace = sorted(rows(os.path.join(d, "trace", "**", "*kernel_trace.csv")),
                   key=lambda r: int(r["Start_Timestamp"]))
    for r in trace:
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== kernel trace (ns), whole process ==")
    tot = sum(sum(v) for v in dur.values()) or 1
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
        v2 = sorted(v)
        print("%-70s calls %5d  avg %10.1f  med %10.1f  min %10d  max %10d  %5.1f%%"
              % (k, len(v), sum(v) / len(v), v2[len(v) // 2], v2[0], v2[-1], 100.0 * sum(v) / tot))
    print("== timed region only (the last %d dispatches = bench.py's rounds x steps) ==" % timed)
    timed_avg = {}
    for k, v in dur.items():
        if ("mog2" in k or "morph" in k or "fd_" in k) and len(v) >= timed:
            t = v[-timed:]
            timed_avg[k] = sum(t) / len(t)
            print("%-70s avg %10.1f ns  med %10.1f  over %d dispatches"
                  % (k, timed_avg[k], sorted(t)[len(t) // 2], len(t)))
    # PMC: mean per dispatch per kernel over the last `timed` dispatches
    pmc = defaultdict(lambda: defaultdict(list))
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for r in rows(os.path.join(sub, "**", "*counter_collection.csv")):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n== PMC (mean per dispatch over the timed region) ==")
    out = {}
    for k, cs in pmc.items():
        if "mog2" not in k and "morph" not in k:
            continue
        print(k)
        for c, v in sorted(cs.items()):
            v = v[-timed:] if len(v) >= timed else v[len(v) // 2:]
            m = sum(v) / len(v)
            out.setdefault(k, {})[c] = m
            print("    %-28s %16.1f   (n=%d)" % (c, m, len(v)))
    rec = None
    for k, cs in out.items():
        if "mog2" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            # MI355X_MICROARCH.md HBM: counters are in KiB; on gfx950 FETCH_SIZE
            # reports 1/2 of the bytes of a wide coalesced streaming read -> x2
            # (re-calibrated for dword-per-lane streams, profiles/r01_counter_calibration.txt)
            fetch_raw = cs["FETCH_SIZE"] * 1024.0
            write = cs["WRITE_SIZE"] * 1024.0
            rec = {"config": {"streams": streams, "height": h, "width": w, "frames_per_step": fps,
                              "input": inp},
                   "build_id": build_id(),
                   "kernel": k, "fetch_size_raw_bytes": fetch_raw, "fetch_bytes_corrected": 2 * fetch_raw,
                   "write_bytes": write, "hbm_bytes_per_launch": 2 * fetch_raw + write,
                   "sq_insts_valu": cs.get("SQ_INSTS_VALU"), "sq_insts_salu": cs.get("SQ_INSTS_SALU"),
                   "sq_busy_cycles": cs.get("SQ_BUSY_CYCLES"), "sq_waves": cs.get("SQ_WAVES"),
                   "kernel_trace_avg_ns": timed_avg.get(k),
                   "note": "means per launch over the timed region; FETCH_SIZE x2 per MI355X_MICROARCH.md "
                           "(gfx950 tallies 128-B requests as 64 B)"}
            if cs.get("SQ_INSTS_VALU") and timed_avg.get(k):
                t = timed_avg[k] * 1e-9
                rec["valu_issue_frac"] = cs["SQ_INSTS_VALU"] * 2 / (1024 * 2.4e9 * t)   # 2 cycles / wave64 instr
                rec["cycles_per_valu_inst"] = 1024 * 2.4e9 * t / cs["SQ_INSTS_VALU"]
                rec["hbm_traffic_GBps"] = rec["hbm_bytes_per_launch"] / t / 1e9
            print("\npmc record:", json.dumps(rec))
            with open(os.path.join(d, "pmc.json"), "w") as f:
                json.dump(rec, f)
    return rec


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]), a[3], int(a[4]), *[int(x) for x in a[5:8]])
