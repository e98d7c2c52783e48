#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lean_decode.py -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r06f_pytest_lean.log 2>&1
echo "pytest rc=$?"; tail -5 $out/r06f_pytest_lean.log
for c in cfg1; do
  for lean in 1; do
    echo "=== $c MP_LEAN=$lean" | tee -a $out/r06f_phase_spread.txt
    MP_LEAN=$lean timeout 600 python scripts/phase_spread.py $c 8 randn graph 30 2>&1 | grep -v "amdgpu.ids\|Warning\|nanm\|_ureduce\|  st = \|  r0 = \|acc.append\|nan /" | tee -a $out/r06f_phase_spread.txt
  done
done
for c in cfg1 cfg4 cfg0; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06f_ab_lean.txt
done
