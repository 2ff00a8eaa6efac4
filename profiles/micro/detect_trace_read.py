#!/usr/bin/env python3
"""Per-frame timeline of GOFPID.detect() from a rocprofv3 trace directory (kernel_trace.csv +
memory_copy_trace.csv): a frame = one large H2D copy and everything up to the next one."""
import csv
import glob
import statistics
import sys

d = sys.argv[1]
ev = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].split("<")[0].split("::")[-1]))
for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        big = (e - s) > 50000
        ev.append((s, e, ("H2D frame" if big else "copy " + r.get("Direction", "?").replace("MEMORY_COPY_", ""))))
ev.sort()
starts = [i for i, x in enumerate(ev) if x[2] == "H2D frame"]
frames = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])][-200:]
sig = statistics.mode(tuple(x[2] for x in f) for f in frames)
frames = [f for f in frames if tuple(x[2] for x in f) == sig]
print("%d frames with the modal event sequence; medians in us" % len(frames))
prev_end = None
for k, name in enumerate(sig):
    dur = statistics.median((f[k][1] - f[k][0]) / 1e3 for f in frames)
    gap = statistics.median((f[k][0] - f[k - 1][1]) / 1e3 for f in frames) if k else 0.0
    print("  %-28s gap before %6.1f   duration %6.1f" % (name, gap, dur))
print("  first event start -> last event end: %.1f us; frame period (start to next start): %.1f us"
      % (statistics.median((f[-1][1] - f[0][0]) / 1e3 for f in frames),
         statistics.median((b[0][0] - a[0][0]) / 1e3 for a, b in zip(frames[:-1], frames[1:]))))
