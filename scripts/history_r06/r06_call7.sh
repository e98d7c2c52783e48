#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for c in cfg1 cfg3; do
    echo "=== $c MP_LEAN=1" | tee -a $out/r06g_phase_spread.txt
    MP_LEAN=1 timeout 600 python scripts/phase_spread.py $c 8 randn graph 30 2>&1 | grep -v "amdgpu.ids\|Warning\|nanm\|_ureduce\|  st = \|  r0 = \|acc.append\|nan /" | tee -a $out/r06g_phase_spread.txt
done
echo "=== cfg1 clustered MP_LEAN=1" | tee -a $out/r06g_phase_spread.txt
MP_LEAN=1 timeout 600 python scripts/phase_spread.py cfg1 8 clustered graph 30 2>&1 | grep -v "amdgpu.ids\|Warning\|nanm\|_ureduce\|  st = \|  r0 = \|acc.append\|nan /" | tee -a $out/r06g_phase_spread.txt
