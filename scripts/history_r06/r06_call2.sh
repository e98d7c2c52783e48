#!/bin/bash
# round 6, GPU call: LEAN decode tests, then phase stamps (every workgroup, last launch of a replayed graph) of the by-products-on
# and the LEAN launch at cfg 1 and cfg 3
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_lean_decode.py -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r06b_pytest_lean.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -12 $out/r06b_pytest_lean.log
for c in cfg1 cfg3; do
  for lean in 0 1; do
    echo "=== $c MP_LEAN=$lean" | tee -a $out/r06b_phase_spread.txt
    MP_LEAN=$lean timeout 600 python scripts/phase_spread.py $c 8 randn graph 30 2>&1 | grep -v amdgpu.ids | tee -a $out/r06b_phase_spread.txt
  done
done
echo "done t=$(( $(date +%s) - t0 ))"
