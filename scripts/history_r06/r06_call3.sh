#!/bin/bash
# round 6, GPU call: phase stamps (every workgroup, last launch of a replayed graph) of the by-products-on and the LEAN launch at
# cfg 1 and cfg 3; A/B of the two forms
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
for c in cfg1 cfg3; do
  for lean in 0 1; do
    echo "=== $c MP_LEAN=$lean" | tee -a $out/r06c_phase_spread.txt
    MP_LEAN=$lean timeout 600 python scripts/phase_spread.py $c 8 randn graph 30 2>&1 | grep -v "amdgpu.ids\|Warning\|nanm\|_ureduce\|  st = \|  r0 = \|acc.append" | tee -a $out/r06c_phase_spread.txt
  done
done
for c in cfg1 cfg4 cfg3 cfg2; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06c_ab_lean.txt
done
echo "done t=$(( $(date +%s) - t0 ))"
