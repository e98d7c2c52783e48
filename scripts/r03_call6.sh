#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r03g_pytest.log 2>&1; tail -4 $out/r03g_pytest.log
for rep in 1 2; do
for lib in "" "--lib magicpig_amd/lib/variants/ticket/libmagicpig_hip.so"; do
  for c in "cfg1 randn" "cfg1 clustered" "cfg4 randn" "cfg2 randn" "cfg3 randn" "cfg2 clustered"; do
    set -- $c
    timeout 300 python bench.py --config $1 --data $2 --no-cpu-baseline --no-host-mode $lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', '${lib:-product}', 'us/layer %.2f launch %.2f' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us']))" >> $out/r03g_ab.txt 2>&1
  done
done
done
cat $out/r03g_ab.txt
for c in cfg0 cfg1 cfg4; do timeout 600 python scripts/stress_cluster.py $c 150 2>&1 | tail -2; done > $out/r03g_stress.txt 2>&1
timeout 600 python scripts/stress_cluster.py cfg1 60 contend 2>&1 | tail -2 >> $out/r03g_stress.txt
cat $out/r03g_stress.txt
timeout 300 python scripts/phase_spread.py cfg1 10 randn > $out/r03g_phase_cfg1_randn.txt 2>&1
timeout 300 python scripts/phase_spread.py cfg2 6 randn > $out/r03g_phase_cfg2_randn.txt 2>&1
