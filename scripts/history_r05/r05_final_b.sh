#!/bin/bash
# round 5, final measurement pass, part B: full GPU suite, the bench lines of every configuration (profiles/hbm_traffic_latest.json already
# holds part A's traffic), the driver-style default line, cluster stress, smoke
tag=r05
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/${tag}_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/${tag}_bench_driver_style.json 2> $out/${tag}_bench_driver_style.err; echo "driver-style rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/${tag}_bench_driver_style.err
timeout 900 python bench.py > $out/${tag}_bench_cfg1.json 2> $out/${tag}_bench_cfg1.err; echo "cfg1 rc=$? t=$(( $(date +%s) - t0 ))"
for c in cfg2 cfg3 cfg4 cfg0; do
  timeout 900 python bench.py --config $c > $out/${tag}_bench_$c.json 2> $out/${tag}_bench_$c.err; echo "$c rc=$? t=$(( $(date +%s) - t0 ))"
done
for c in cfg1 cfg2; do
  timeout 900 python bench.py --config $c --data clustered > $out/${tag}_bench_${c}_clustered.json 2> $out/${tag}_bench_${c}_clustered.err; echo "$c clustered rc=$? t=$(( $(date +%s) - t0 ))"
done
for c in cfg1 cfg4 cfg2; do timeout 300 python scripts/stress_cluster.py $c 60 2>&1 | grep -v amdgpu.ids; done | tee $out/${tag}_stress.txt
timeout 300 python scripts/stress_cluster.py cfg1 40 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_stress.txt
timeout 300 python scripts/stress_cluster.py cfg4 40 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_stress.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "done t=$(( $(date +%s) - t0 ))"
