#!/bin/bash
# round 6, GPU call: LEAN decode (wave-owned eager gather, no barrier behind the counting): tests, stamps, A/B
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_lean_decode.py -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r06d_pytest_lean.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -12 $out/r06d_pytest_lean.log
for c in cfg1 cfg3; do
  for lean in 1; do
    echo "=== $c MP_LEAN=$lean" | tee -a $out/r06d_phase_spread.txt
    MP_LEAN=$lean timeout 600 python scripts/phase_spread.py $c 8 randn graph 30 2>&1 | grep -v "amdgpu.ids\|Warning\|nanm\|_ureduce\|  st = \|  r0 = \|acc.append\|nan /" | tee -a $out/r06d_phase_spread.txt
  done
done
for c in cfg1 cfg4 cfg3 cfg2 cfg0; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06d_ab_lean.txt
done
timeout 600 python scripts/ab_libs.py cfg1 product@--by-products,1 product --reps 3 --data clustered 2>&1 | grep -v amdgpu.ids | tee -a $out/r06d_ab_lean.txt
echo "done t=$(( $(date +%s) - t0 ))"
