#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04g_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"
tail -8 $out/r04g_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/r04g_bench_driver_style.json 2> $out/r04g_bench_driver_style.err
tail -3 $out/r04g_bench_driver_style.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04g_bench_driver_style.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','sparse_attn_us_per_layer','value_clustered','sparse_attn_us_per_layer_clustered','speedup_vs_cpu')})
print(d['roofline']); print(d.get('host_mode')); print(d.get('cpu_baseline')); print(d.get('observed_clustered'))
PY
echo "bench t=$(( $(date +%s) - t0 ))"
bash scripts/profile_round.sh r04 cfg1 cfg2 cfg3 cfg4 > $out/r04_profile_round.log 2>&1
echo "profile t=$(( $(date +%s) - t0 ))"
ls $out | grep "^r04_" | head -80
