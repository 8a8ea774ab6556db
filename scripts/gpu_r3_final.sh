#!/bin/bash
# round 3, closing GPU pass: the suite three times over, the default bench line, the rocprofv3 evidence
mkdir -p gpurun_out/r3final
O=gpurun_out/r3final
bash scripts/repeat_suite.sh 3 > $O/repeat_suite.log 2>&1; echo "repeat_suite rc=$?" > $O/summary.txt
cp gpurun_out/repeat_suite/summary.txt $O/repeat_suite_summary.txt
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?" >> $O/summary.txt
bash scripts/profile_r03.sh > $O/profile.log 2>&1; echo "profile rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/repeat_suite_summary.txt $O/bench_default.time
