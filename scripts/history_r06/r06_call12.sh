#!/bin/bash
# round 6: the driver-style line with the new legs (cfg3_share, e2e) -- wall time and contents
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/r06l_bench_driver_style.json 2> $out/r06l_bench_driver_style.err
echo "rc=$?"; tail -5 $out/r06l_bench_driver_style.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06l_bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "us/layer", d["sparse_attn_us_per_layer"], "roof", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("form", d["config"]["decode_form"])
print("cpu", d.get("cpu_baseline", {}).get("value"), "speedup", d.get("speedup_vs_cpu"))
print("host", d.get("host_mode", {}).get("us_per_layer"), d.get("host_mode", {}).get("speedup_vs_cpu_layer"))
for k, v in d.get("legs", {}).items():
    if k == "e2e":
        for kk, vv in v.items():
            print("e2e", kk, vv.get("tokens_per_s"), vv.get("ms_per_step"), vv.get("split_ms"), vv.get("split_sum_ms"), vv.get("leg_wall_s"), vv.get("failed"))
    else:
        print(k, v.get("tokens_per_s"), v.get("us_per_layer"), (v.get("roofline") or {}).get("frac"), v.get("leg_wall_s"), v.get("failed"))
print("failed", d.get("failed_legs"))
PY
