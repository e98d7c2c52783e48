#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for c in cfg1 cfg3; do
  for lean in 0 1; do
    echo "=== $c MP_LEAN=$lean" | tee -a $out/r06e_phase_spread.txt
    MP_LEAN=$lean timeout 600 python scripts/phase_spread.py $c 8 randn graph 30 2>&1 | grep -v "amdgpu.ids\|Warning\|nanm\|_ureduce\|  st = \|  r0 = \|acc.append\|nan /" | tee -a $out/r06e_phase_spread.txt
  done
done
