#!/bin/bash
# round 6, call 29: lsh_head_body split into phase functions (hash_query_row, cluster_handoff): timings must not move
out=gpurun_out; mkdir -p $out
{
for c in cfg1 cfg3 cfg4 cfg2; do python scripts/ab_libs.py $c product prerefactor --reps 8 2>&1 | grep -v amdgpu.ids | tail -4; done
python scripts/ab_libs.py cfg1 product prerefactor --reps 6 --data clustered 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/ab_libs.py cfg1 product@--by-products,1 prerefactor@--by-products,1 --reps 6 2>&1 | grep -v amdgpu.ids | tail -4
} | tee $out/r06_ab_refactor.txt
