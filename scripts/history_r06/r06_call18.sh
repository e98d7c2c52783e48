#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lean_decode.py -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -6
for c in cfg1 cfg4; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 r06base product --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06r_ab_lean2.txt
done
timeout 600 python scripts/ab_libs.py cfg1 product@--by-products,1 r06base product --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06r_ab_lean2.txt
timeout 600 python scripts/ab_libs.py cfg0 product@--by-products,1 r06base product --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $out/r06r_ab_lean2.txt
