#!/bin/bash
# round 6, GPU call 1: the LEAN decode (MP_DECODE_NO_BYPRODUCTS, unordered LDS stage) -- its tests, then A/B against the
# by-products-on launch of the same build at cfg 1 / 3 / 4 / 0 (alternating regions, scripts/ab_libs.py)
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_lean_decode.py -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r06a_pytest_lean.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -25 $out/r06a_pytest_lean.log
for c in cfg1 cfg4 cfg3 cfg0; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06a_ab_lean.txt
  echo "ab $c t=$(( $(date +%s) - t0 ))"
done
echo "done t=$(( $(date +%s) - t0 ))"
