#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box.
#   usage: bash profiles/run_profiles.sh <tag>        (run from the repo root via gpurun)
# Outputs under gpurun_out/prof_<tag>/ ; summaries are then copied into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
FPS=${2:-8}   # frames per stream per step (bench.py default 8; 1 = per-frame form)
# same K / W as the default bench.py run, so that the per-kernel averages are comparable
ARGS="bench.py --frames-per-step $FPS --cpu-seconds 0 --no-events"
# 1. per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace failed" >> $OUT/trace.log
# 2-5. PMC passes, one counter group per run (TCC: FETCH_SIZE needs 3 slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || echo failed >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || echo failed >> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $ARGS > $OUT/pmc_sq1.log 2>&1 || echo failed >> $OUT/pmc_sq1.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || echo failed >> $OUT/pmc_sq2.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1 || echo failed >> $OUT/pmc_tcc.log
python3 profiles/summarize.py $OUT $FPS > $OUT/summary.txt 2>&1
# keep the judged artefacts small: stats CSVs + summary; drop per-dispatch dumps
mkdir -p $OUT/keep
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/keep/ \; 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_tcc
cat $OUT/summary.txt
