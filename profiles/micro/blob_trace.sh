#!/bin/bash
# per-kernel durations of the blob extraction (1-mask calls first, then 64-mask calls)
export TMPDIR=/tmp
mkdir -p gpurun_out/blob_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/blob_trace/t -o t -- python3 profiles/micro/blob_rate.py > gpurun_out/blob_trace/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/blob_trace/t/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        m = re.search(r"(blob_\w+)", r["Kernel_Name"])
        if m:
            d[m.group(1)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot1 = tot64 = 0
for k, v in d.items():
    v1, v64 = v[3:53], v[-20:]          # 3 + 50 one-mask calls, 3 + 20 sixty-four-mask calls
    a1, a64 = sum(v1) / len(v1) / 1e3, sum(v64) / len(v64) / 1e3
    tot1 += a1; tot64 += a64
    print("%-22s 1-mask call: %7.1f us    64-mask call: %8.1f us" % (k, a1, a64))
print("%-22s 1-mask call: %7.1f us    64-mask call: %8.1f us" % ("sum of kernels", tot1, tot64))
PY
cat gpurun_out/blob_trace/run.log | tail -4
rm -rf gpurun_out/blob_trace/t
