#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for c in cfg1 cfg4; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product fixedclaim --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06i_ab_lean.txt
  timeout 600 python scripts/ab_libs.py $c product min4 min12 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06i_ab_lean.txt
done
timeout 600 python scripts/ab_libs.py cfg1 product@--by-products,1 product fixedclaim --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06i_ab_lean.txt
timeout 600 python scripts/ab_libs.py cfg1 product min4 min12 --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06i_ab_lean.txt
