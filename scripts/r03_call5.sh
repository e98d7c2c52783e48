#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "host_buffer or fused_decode or direct_slots or spill or cfg1_shaped or split_over or token_range" > $out/r03f_pytest.log 2>&1; tail -3 $out/r03f_pytest.log
for rep in 1 2; do
for lib in "" "--lib magicpig_amd/lib/variants/slots_old/libmagicpig_hip.so"; do
  for c in "cfg1 randn" "cfg1 clustered" "cfg4 randn"; do
    set -- $c
    timeout 300 python bench.py --config $1 --data $2 --no-cpu-baseline --no-host-mode $lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', '${lib:-product}', 'us/layer %.2f launch %.2f' % (d['sparse_attn_us_per_layer'], d['roofline']['avg_launch_us']))" >> $out/r03f_slots_ab.txt 2>&1
  done
done
done
cat $out/r03f_slots_ab.txt
timeout 300 python scripts/host_mode_times.py cfg1 50 2>&1 | grep -v amdgpu.ids > $out/r03f_host_mode_cfg1.txt; cat $out/r03f_host_mode_cfg1.txt
timeout 300 python scripts/phase_spread.py cfg1 10 randn > $out/r03f_phase_cfg1_randn.txt 2>&1
