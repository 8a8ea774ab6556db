#!/bin/bash
# round 3, closing GPU pass: smoke, the suite three times over, the default bench line, configs[3] stand-in, the rocprofv3 evidence
mkdir -p gpurun_out/r3final
O=gpurun_out/r3final
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/summary.txt
bash scripts/repeat_suite.sh 3 > $O/repeat_suite.log 2>&1; echo "repeat_suite rc=$?" >> $O/summary.txt
cp gpurun_out/repeat_suite/summary.txt $O/repeat_suite_summary.txt
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?" >> $O/summary.txt
python scripts/bench_multivariate.py > $O/bench_multivariate.log 2>&1; echo "multivariate rc=$?" >> $O/summary.txt
python bench.py --euclidean --no-cpu-baseline --no-scoring --no-drop-in > $O/bench_euclidean.json 2> $O/bench_euclidean.err; echo "euclidean rc=$?" >> $O/summary.txt
bash scripts/profile_r03.sh > $O/profile.log 2>&1; echo "profile rc=$?" >> $O/summary.txt
bash scripts/sq_profile.sh > $O/sq_profile.log 2>&1; echo "sq counters rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/repeat_suite_summary.txt $O/bench_default.time; tail -n 3 $O/smoke.log; tail -n 4 $O/bench_multivariate.log
