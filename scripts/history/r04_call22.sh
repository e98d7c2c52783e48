#!/bin/bash
# R4-14, one shot with the round's last GPU minute: K | V rows requested at a token's second hit (A/B build, mode by option), one workload per config
out=gpurun_out; mkdir -p $out; : > $out/r04v_prefetch.txt
V=magicpig_amd/lib/variants/prefetch/libmagicpig_hip.so
for c in cfg1 cfg3; do
  timeout 24 python bench.py --config $c --lib $V --ab-option decode_prefetch=0,1,2 --ab-reps 3 --steps 64 --warmup 8 --no-cpu-baseline --no-host-mode --no-clustered-leg 2>&1 | tail -1 >> $out/r04v_prefetch.txt
done
cat $out/r04v_prefetch.txt
