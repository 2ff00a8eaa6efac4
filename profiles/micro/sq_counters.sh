#!/bin/bash
# A/B of the 8-frame MOG2 launch at the counter level: time (kernel trace), instruction counts by
# kind, VALU busy / wait cycles.  usage: bash profiles/micro/sq_counters.sh <tag> [library.so]
# (run from the repo root via gpurun; the library defaults to the in-tree build)
set -u
export TMPDIR=/tmp
TAG=${1:-cur}
LIBSO=${2:-}
[ -n "$LIBSO" ] && export GOFPID_LIB=$(realpath $LIBSO)
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 32 --rounds 2 --cpu-seconds 0 --no-events --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o trace -- python3 $ARGS > $OUT/t.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o pmc -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -o pmc -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_IFETCH SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/c -o pmc -- python3 $ARGS > $OUT/c.log 2>&1
if [ "${WITH_TCC:-0}" = "1" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d -o pmc -- python3 $ARGS > $OUT/d.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/e -o pmc -- python3 $ARGS > $OUT/e.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/f -o pmc -- python3 $ARGS > $OUT/f.log 2>&1
fi
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for sub in ("a", "b", "c", "d", "e", "f"):
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        for r in csv.DictReader(open(path)):
            if "mog2_frames" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for path in glob.glob("%s/t/**/*kernel_trace.csv" % out, recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "mog2_frames" in r["Kernel_Name"]]
d = dur[-64:]
print("== %s: mog2_frames_kernel<3>, mean per dispatch over the last 64 dispatches ==" % out)
if d:
    print("  %-28s %16.1f" % ("duration_us (kernel trace)", sum(d) / len(d) / 1e3))
for k, v in sorted(acc.items()):
    v = v[-64:]
    print("  %-28s %16.1f" % (k, sum(v) / len(v)))
PY
rm -rf $OUT/t $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e $OUT/f
cat $OUT/summary.txt
