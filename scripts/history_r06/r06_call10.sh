#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
for c in cfg1 cfg4 cfg0; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product fixedclaim --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06j_ab_lean.txt
done
timeout 600 python scripts/ab_libs.py cfg1 product@--by-products,1 product fixedclaim --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06j_ab_lean.txt
timeout 600 python scripts/ab_libs.py cfg3 product@--by-products,1 product --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $out/r06j_ab_lean.txt
echo "ab t=$(( $(date +%s) - t0 ))"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r06j_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -8 $out/r06j_pytest.log
