#!/bin/bash
# What saturates in the 8-frame MOG2 launch?  VALU busy time and the instruction mix by class,
# benchmark scene and worst-case noise input (two PMC passes each).  usage: bash profiles/micro/valu_counters.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/valu_counters
mkdir -p $OUT
for INPUT in scene noise; do
  EXTRA=""; [ "$INPUT" = "noise" ] && EXTRA="--noise --pool 64"
  ARGS="bench.py --steps 32 --rounds 2 --cpu-seconds 0 --no-events --no-extra $EXTRA"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_CYCLES --output-format csv -d $OUT/${INPUT}_a -o pmc -- python3 $ARGS > $OUT/${INPUT}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $OUT/${INPUT}_b -o pmc -- python3 $ARGS > $OUT/${INPUT}_b.log 2>&1
done
python3 - <<'PY' > gpurun_out/valu_counters/summary.txt
import csv, glob, collections
for inp in ("scene", "noise"):
    acc = collections.defaultdict(list)
    for sub in ("a", "b"):
        for path in glob.glob("gpurun_out/valu_counters/%s_%s/**/*counter_collection.csv" % (inp, sub), recursive=True):
            for r in csv.DictReader(open(path)):
                if "mog2_frames" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s: mog2_frames_kernel<3>, mean per dispatch over the last 64 dispatches ==" % inp)
    for k, v in sorted(acc.items()):
        v = v[-64:]
        print("  %-28s %16.1f" % (k, sum(v) / len(v)))
PY
rm -rf $OUT/*_a $OUT/*_b
cat $OUT/summary.txt
