#!/bin/bash
# round 5, GPU call 10: the R = 1 stream -- both chunks + pool in one round (product), both chunks only (stream_id1), round 4's form (stream2)
out=$(pwd)/gpurun_out; mkdir -p $out
for c in "cfg2 clustered" "cfg2 randn" "cfg3 randn"; do
  set -- $c
  timeout 400 python scripts/ab_libs.py $1 product stream_id1 stream2 --data $2 --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05l_ab_stream_variants.txt
