#!/bin/bash
# round 3, fourth GPU pass: the whole -m gpu suite on the current sources + the store16 hazard sweep (dev library)
mkdir -p gpurun_out/r3d
O=gpurun_out/r3d
timeout 600 python scripts/diag_store16.py > $O/store16.txt 2>&1; echo "store16 rc=$?" > $O/summary.txt
timeout 1700 python -m pytest tests -q -m gpu -x > $O/all_tests.log 2>&1; echo "all tests rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/store16.txt; tail -n 8 $O/all_tests.log
