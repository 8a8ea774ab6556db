#!/bin/bash
# rocprofv3 evidence for round 2 (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_r02.sh
# 1. kernel trace + stats of the default bench command (training epoch, scoring kernels, ball kernels)
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md) of the same command
# Outputs under gpurun_out/prof_r02/; the summaries to commit are copied into profiles/ by scripts/collect_profiles_r02.py
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"   # (one signal per GPU only: every launch of a kernel is a like launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/bench_write.json 2> $OUT/bench_write.err
cd $REPO
python3 scripts/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/r02_pmc_traffic.json 2> $OUT/pmc_traffic.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -40 "$f" > $OUT/r02_kernel_stats.csv
# keep gpurun_out small: counter_collection / kernel_trace CSVs of the PMC passes are tens of MB
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT
tail -2 $OUT/bench_trace.err
