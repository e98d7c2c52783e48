#!/bin/bash
out=gpurun_out; mkdir -p $out; rm -f $out/r03h_ab.txt
for rep in 1 2; do
for lib in "" "--lib magicpig_amd/lib/variants/s00/libmagicpig_hip.so" "--lib magicpig_amd/lib/variants/s11/libmagicpig_hip.so" "--lib magicpig_amd/lib/variants/ticket/libmagicpig_hip.so"; do
  for c in "cfg1 randn" "cfg1 clustered" "cfg4 randn" "cfg2 randn" "cfg3 randn" "cfg2 clustered"; do
    set -- $c
    case "$lib" in *s00*|*s11*) case $1 in cfg1|cfg4) continue;; esac;; esac
    timeout 300 python bench.py --config $1 --data $2 --no-cpu-baseline --no-host-mode $lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', '${lib:-product}', 'us/layer %.2f launch %.2f' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us']))" >> $out/r03h_ab.txt 2>&1
  done
done
done
cat $out/r03h_ab.txt
timeout 300 python scripts/phase_spread.py cfg1 10 randn > $out/r03h_phase_cfg1_randn.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_decode_harness.py -x -q > $out/r03h_pytest_harness.log 2>&1; tail -30 $out/r03h_pytest_harness.log
