#!/bin/bash
# round 6, item 2: the query SimHash by MFMA -- (a) as its own launch (`decode_mfma_hash`) against the fused VALU prologue on
# today's kernels; (b) the best case of a quad-shared in-launch MFMA hash at one workgroup per head (probe)
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for n in 1 30; do
  echo "== quad_hash_probe, chains of $n launch(es)" | tee -a $out/r06o_quad_probe.txt
  timeout 120 ./scripts/probes/quad_hash_probe $n 2>&1 | tee -a $out/r06o_quad_probe.txt
done
for c in cfg1 cfg2 cfg3 cfg4; do
  timeout 600 python scripts/ab_libs.py $c product@--by-products,1 product@--mfma-hash --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $out/r06o_ab_mfma_hash.txt
done
