#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "host_buffer or fused_decode or pipeline or direct_slots or spill or cfg1_shaped or split_over" > $out/r03d_pytest.log 2>&1; tail -5 $out/r03d_pytest.log
for d in randn clustered skewed; do
  timeout 300 python bench.py --config cfg1 --data $d --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg1', '$d', 'us/layer %.2f launch %.2f host_mode %s' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us'], d.get('host_mode')))" >> $out/r03d_bench.txt 2>&1
done
timeout 300 python bench.py --config cfg2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg2 randn us/layer %.2f launch %.2f host_mode %s' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us'], d.get('host_mode')))" >> $out/r03d_bench.txt 2>&1
timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-host-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 randn us/layer %.2f launch %.2f' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us']))" >> $out/r03d_bench.txt 2>&1
cat $out/r03d_bench.txt
timeout 300 python scripts/phase_spread.py cfg1 10 clustered > $out/r03d_phase_cfg1_clustered.txt 2>&1
timeout 300 python scripts/host_mode_times.py cfg1 > $out/r03d_host_mode_cfg1.txt 2>&1; cat $out/r03d_host_mode_cfg1.txt
