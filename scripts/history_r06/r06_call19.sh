#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lean_decode.py tests/test_gpu_quad_hash.py -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -6
for c in cfg3 cfg2; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06s_ab_lockstep.txt
done
timeout 600 python scripts/ab_libs.py cfg2 product@--by-products,1 product --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06s_ab_lockstep.txt
