#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_quad_hash.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15; python scripts/experiments/r06_quad_diag.py 2>&1 | grep quad
for c in cfg2 cfg3; do
  timeout 600 python scripts/ab_libs.py $c product@--quad-hash,0 product@--quad-hash,1 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06p_ab_quad.txt
done
timeout 600 python scripts/ab_libs.py cfg2 product@--quad-hash,0 product@--quad-hash,1 --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06p_ab_quad.txt
