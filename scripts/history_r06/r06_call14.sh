#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r06n_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -5 $out/r06n_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-clustered-leg ) > $out/r06n_bench.json 2> $out/r06n_bench.err
echo "rc=$?"; tail -4 $out/r06n_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06n_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "us/layer", d["sparse_attn_us_per_layer"])
c = d["cpu_baseline"]; print("cpu", c["t_retrieve_us"], c["t_attention_us"], c["value"])
for k in ("host_mode", "host_mode_pinned_results"):
    h = d[k]; print(k, h["us_per_layer"], h.get("mean_us"), h.get("speedup_vs_cpu_layer"), h["attention_calls_served"], h["matches_device_entry"])
PY
