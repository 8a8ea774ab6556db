#!/bin/bash
# One-source compile-time variant of the product library: bash scripts/ab_one.sh NAME source.hip "-DFLAG=.."  -> ab_libs/NAME.so
# (the other objects are the current build's; run with HYPAD_LIB_PATH=$PWD/ab_libs/NAME.so)
set -eu
mkdir -p ab_libs /tmp/ab_one_$1
python -m hypad_amd.build > /dev/null
obj=/tmp/ab_one_$1/$(basename $2 .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed -Wno-unused-result -DHYPAD_DIAG=0 $3 -c hypad_amd/csrc/$2 -o $obj
others=$(ls hypad_amd/lib/*.o | grep -v "/$(basename $2 .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/$1.so $obj $others
echo built ab_libs/$1.so "$3"
