#!/bin/bash
# round 6, final measurement pass, part B: full GPU suite, the bench lines of every configuration (profiles/hbm_traffic_latest.json already
# holds part A's traffic), the driver-style default line, cluster stress, smoke
tag=r06
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/${tag}_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/${tag}_bench_driver_style.json 2> $out/${tag}_bench_driver_style.err; echo "driver-style rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/${tag}_bench_driver_style.err
timeout 900 python bench.py > $out/${tag}_bench_cfg1.json 2> $out/${tag}_bench_cfg1.err; echo "cfg1 rc=$? t=$(( $(date +%s) - t0 ))"
for c in cfg2 cfg3 cfg4 cfg0; do
  timeout 900 python bench.py --config $c > $out/${tag}_bench_$c.json 2> $out/${tag}_bench_$c.err; echo "$c rc=$? t=$(( $(date +%s) - t0 ))"
done
timeout 900 python bench.py --config cfg1 --by-products 1 > $out/${tag}_bench_cfg1_byproducts.json 2> $out/${tag}_bench_cfg1_byproducts.err; echo "cfg1 by-products rc=$? t=$(( $(date +%s) - t0 ))"
for c in cfg1 cfg2; do
  timeout 900 python bench.py --config $c --data clustered > $out/${tag}_bench_${c}_clustered.json 2> $out/${tag}_bench_${c}_clustered.err; echo "$c clustered rc=$? t=$(( $(date +%s) - t0 ))"
done
for c in cfg1 cfg4 cfg2; do timeout 300 python scripts/stress_cluster.py $c 60 2>&1 | grep -v amdgpu.ids; done | tee $out/${tag}_stress.txt
timeout 300 python scripts/stress_cluster.py cfg1 40 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_stress.txt
timeout 300 python scripts/stress_cluster.py cfg4 40 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_stress.txt
for c in cfg1 cfg4 cfg0; do timeout 300 python scripts/stress_cluster.py $c 60 x lean 2>&1 | grep -v amdgpu.ids; done | tee -a $out/${tag}_stress.txt
timeout 300 python scripts/stress_cluster.py cfg1 40 contend lean 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_stress.txt
timeout 300 python scripts/stress_cluster.py cfg4 40 contend lean 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_stress.txt
# stamps of the final build (the -DMP_STAMPS=1 variant of the same sources): by-products on and the lean form, cfg 1 and its clustered keys
{ for lean in 0 1; do echo "=== cfg1 MP_LEAN=$lean"; MP_LEAN=$lean timeout 300 python scripts/phase_spread.py cfg1 6 randn graph 30 2>&1 | grep -v amdgpu.ids; done
  echo "=== cfg1 clustered MP_LEAN=1"; MP_LEAN=1 timeout 300 python scripts/phase_spread.py cfg1 6 clustered graph 30 2>&1 | grep -v amdgpu.ids
  echo "=== cfg4 MP_LEAN=1"; MP_LEAN=1 timeout 300 python scripts/phase_spread.py cfg4 6 randn graph 30 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_phase_spread_final.txt 2>&1
echo "phase spread t=$(( $(date +%s) - t0 ))"
python scripts/host_mode_times.py cfg1 2>&1 | grep -v amdgpu.ids > $out/${tag}_host_mode_times.txt; tail -12 $out/${tag}_host_mode_times.txt
timeout 300 python scripts/key_hash_time.py build product 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_prefill_times.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "done t=$(( $(date +%s) - t0 ))"
