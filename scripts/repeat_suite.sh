#!/bin/bash
# The -m gpu suite N times over (default 3), fresh processes: every test must give the same verdict every time.  The intermittent
# store hazard of round 2 (tile_gemm.h GBuf::st4) passed single runs more often than not; repetition is what showed it.
#   bash scripts/repeat_suite.sh [N] [pytest args...]      -> gpurun_out/repeat_suite/{run_K.txt,summary.txt}; exit 1 on any difference / failure
n=${1:-3}; shift
out=gpurun_out/repeat_suite; mkdir -p $out
rc=0
for k in $(seq 1 $n); do
  python -m pytest tests -q -m gpu -rA -p no:cacheprovider "$@" 2>&1 | grep -E "^(PASSED|FAILED|ERROR)" | sort > $out/run_$k.txt
  echo "run $k: $(grep -c ^PASSED $out/run_$k.txt) passed, $(grep -vc ^PASSED $out/run_$k.txt) not passed" | tee -a $out/summary.txt
  if [ $(grep -vc ^PASSED $out/run_$k.txt) -ne 0 ]; then rc=1; fi
  if [ $k -gt 1 ] && ! diff -q $out/run_1.txt $out/run_$k.txt > /dev/null; then echo "run $k differs from run 1" | tee -a $out/summary.txt; rc=1; fi
done
echo "verdict: $([ $rc -eq 0 ] && echo identical and green || echo DIFFERENT OR RED)" | tee -a $out/summary.txt
exit $rc
