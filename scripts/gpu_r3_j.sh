#!/bin/bash
mkdir -p gpurun_out/r3j
O=gpurun_out/r3j
for spg in 32 8 1; do bash scripts/ab_variants.sh run "exp0 exp1 exp2 exp3 exp4" scripts/time_dw.py --spg $spg | sort | uniq -c; done > $O/dw_whatif.txt 2>&1
cat $O/dw_whatif.txt
