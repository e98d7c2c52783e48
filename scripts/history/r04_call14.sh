#!/bin/bash
# ticketless hand-off (self-validating records polled by member 0): correctness subset, A/B by option on one workload, stress, stamps
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -p no:cacheprovider \
  -k "handoff or graph_replay or direct_slots or cfg4 or unusual_shapes or cfg1_full or window" > $out/r04n_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/r04n_pytest.log
for c in "cfg1" "cfg1 --data clustered" "cfg4" "cfg0"; do
  timeout 200 python bench.py --config $c --ab-option decode_ticket=1,0 --ab-reps 4 --no-cpu-baseline --no-host-mode --no-clustered-leg 2>&1 | tail -1
done | tee $out/r04n_ab_option.txt
echo "ab t=$(( $(date +%s) - t0 ))"
bash scripts/ab_multi.sh "r04head product" "cfg1" 2 | tee $out/r04n_ab_lib.txt
for c in cfg1 cfg4; do timeout 200 python scripts/stress_cluster.py $c 60 2>&1 | grep -v amdgpu.ids; done | tee $out/r04n_stress.txt
timeout 200 python scripts/stress_cluster.py cfg1 40 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/r04n_stress.txt
timeout 200 python scripts/phase_spread.py cfg1 8 randn graph 30 > $out/r04n_phase_cfg1.txt 2>&1
echo "done t=$(( $(date +%s) - t0 ))"
