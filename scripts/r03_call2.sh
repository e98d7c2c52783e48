#!/bin/bash
# round 3, GPU call 2: latency probe, 512-thread A/B at cfg 1, two R = 1 half-batch chains at cfg 2 / cfg 3
out=gpurun_out; mkdir -p $out
timeout 300 scripts/probes/lat_probe > $out/r03b_lat_probe.txt 2>&1
for lib in "" "--lib magicpig_amd/lib/variants/t512/libmagicpig_hip.so"; do
  for d in randn clustered; do
    timeout 300 python bench.py --config cfg1 --data $d --no-cpu-baseline $lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg1', '$d', '$lib' or 'product', 'us/layer %.2f launch %.2f' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us']))" >> $out/r03b_t512.txt 2>&1
  done
done
cat $out/r03b_t512.txt
{
for c in cfg2 cfg3; do
  timeout 300 python scripts/two_chain_probe.py $c 1 0 6 2>&1 | tail -1
  timeout 300 python scripts/two_chain_probe.py $c 1 0 6 4 1 2>&1 | tail -1
  for sk in 0 6 10 14 18; do
    timeout 300 python scripts/two_chain_probe.py $c 2 $sk 6 8 1 2>&1 | tail -1
  done
done
} > $out/r03b_two_chains.txt 2>&1
cat $out/r03b_two_chains.txt
