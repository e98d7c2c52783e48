#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for c in cfg1 cfg4; do
  timeout 600 python scripts/ab_libs.py $c product cl12 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06k_ab_cl.txt
done
timeout 600 python scripts/ab_libs.py cfg1 product cl12 --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06k_ab_cl.txt
