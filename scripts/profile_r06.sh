#!/bin/bash
# rocprofv3 evidence for round 6 (run on the GPU box through gpurun from the repo root):  bash scripts/profile_r06.sh
# 1. kernel trace + stats of the default bench command            -> profiles/r06_kernel_stats.csv
# 2. two PMC passes of it (FETCH_SIZE, WRITE_SIZE: separate passes)  -> profiles/r06_pmc_traffic.json
# 3. one PMC pass with the SQ MFMA-busy / busy / waves counters + GRBM_GUI_ACTIVE over scripts/profile_mfma_target.py
#                                                                 -> profiles/r06_mfma_util.json
# 4. kernel trace + stats at 8 and 32 signals per GPU             -> profiles/r06_signals{8,32}_kernel_stats.csv
# (programs directly after `--`; counters never together with a sys/hip trace)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-drop-in --no-sharded-scoring --no-extra-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/bench_write.json 2> $OUT/bench_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 $REPO/scripts/profile_mfma_target.py > $OUT/mfma_target.log 2> $OUT/mfma_target.err
for n in 8 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sig$n -o trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --signals-per-gpu $n --no-graph --no-cpu-baseline --no-scoring --no-drop-in --no-extra-configs > $OUT/bench_sig$n.json 2> $OUT/bench_sig$n.err
  f=$(find $OUT/trace_sig$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $OUT/r06_signals${n}_kernel_stats.csv
done
cd $REPO
python3 scripts/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/r06_pmc_traffic.json 2> $OUT/pmc_traffic.err
python3 scripts/mfma_util.py $OUT/pmc_mfma > $OUT/r06_mfma_util.json 2> $OUT/mfma_util.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -40 "$f" > $OUT/r06_kernel_stats.csv
cp $OUT/bench_trace.json $OUT/r06_bench_under_rocprof.json
# keep gpurun_out small: counter_collection / kernel_trace CSVs of the PMC passes are tens of MB
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT
for f in $OUT/bench_trace.err $OUT/mfma_target.err $OUT/mfma_util.err; do tail -n 2 $f; done
