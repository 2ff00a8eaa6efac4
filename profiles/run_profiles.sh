#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py configuration on the GPU box.
#   usage: bash profiles/run_profiles.sh <tag> <frames per step> <scene|noise> [steps] [rounds]
#   (run from the repo root via gpurun)
# Outputs under gpurun_out/prof_<tag>/ (summary.txt, pmc.json, keep/*kernel_stats.csv);
# the summaries judged are then copied into profiles/ and pmc.json merged into
# profiles/pmc_latest.json (profiles/merge_pmc.py).
set -u
TAG=${1:-r02}
FPS=${2:-8}      # frames per stream per step (bench.py default 8; 1 = per-frame form)
INPUT=${3:-scene}
STEPS=${4:-128}
ROUNDS=${5:-4}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
NOISE=""
[ "$INPUT" = "noise" ] && NOISE="--noise --pool 64"
# the bench.py protocol itself (pre-roll, warm-up, rounds), without the HIP-event rounds and
# the extra blocks, so that the last ROUNDS x STEPS detector dispatches are the timed region
ARGS="bench.py --frames-per-step $FPS --steps $STEPS --rounds $ROUNDS --cpu-seconds 0 --no-events --no-extra $NOISE"
# 1. per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace failed" >> $OUT/trace.log
# 2-4. PMC passes, one counter group per run (TCC: FETCH_SIZE needs 3 slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || echo failed >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || echo failed >> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $ARGS > $OUT/pmc_sq1.log 2>&1 || echo failed >> $OUT/pmc_sq1.log
if [ "${GOF_PROF_FULL:-0}" = "1" ]; then
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || echo failed >> $OUT/pmc_sq2.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1 || echo failed >> $OUT/pmc_tcc.log
fi
python3 profiles/summarize.py $OUT $FPS $INPUT $((STEPS * ROUNDS)) > $OUT/summary.txt 2>&1
# the bench line of the traced run, for the record
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
# keep the judged artefacts small: stats CSVs + summary; drop per-dispatch dumps
mkdir -p $OUT/keep
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/keep/ \; 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_tcc
cat $OUT/summary.txt
