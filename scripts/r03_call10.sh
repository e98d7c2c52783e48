#!/bin/bash
out=gpurun_out; rm -f $out/r03k_ab.txt
for rep in 1 2 3; do
for lib in "" "--lib magicpig_amd/lib/variants/slim/libmagicpig_hip.so"; do
  for c in "cfg1 randn" "cfg1 clustered" "cfg2 randn" "cfg3 randn" "cfg4 randn"; do
    set -- $c
    timeout 300 python bench.py --config $1 --data $2 --no-cpu-baseline --no-host-mode $lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', '${lib:-product}', 'us/layer %.2f launch %.2f' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us']))" >> $out/r03k_ab.txt 2>&1
  done
done
done
sort $out/r03k_ab.txt
