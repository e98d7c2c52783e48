#!/bin/bash
# round 4, final pass: the PMC passes the profile round missed, the bench lines of the final build (now citing the r04
# traffic file), the full GPU suite, replay stress, graph-mode phase stamps
out=gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
c=cfg2
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rm -rf pmc_${c}_cl_$ctr && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${c}_cl_$ctr -- python $root/bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
  db=$(find /tmp/pmc_${c}_cl_$ctr -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-graph --steps 8 --warmup 2 (r04); KB per dispatch as reported (FETCH_SIZE x 2 on gfx950)"; python scripts/rocprof_pmc.py $db decode; } > $out/r04_pmc_${ctr}_${c}_clustered.md 2>&1
done
echo "pmc t=$(( $(date +%s) - t0 ))"
for c in cfg1 cfg2 cfg3 cfg4 cfg0; do
  timeout 900 python bench.py --config $c > $out/r04_bench_$c.json 2> $out/r04_bench_$c.err
done
for c in cfg1 cfg2; do
  timeout 900 python bench.py --config $c --data clustered > $out/r04_bench_${c}_clustered.json 2> $out/r04_bench_${c}_clustered.err
done
echo "bench t=$(( $(date +%s) - t0 ))"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04i_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -6 $out/r04i_pytest.log
for c in cfg1 cfg4 cfg2; do timeout 300 python scripts/stress_cluster.py $c 100 2>&1 | grep -v amdgpu.ids; done | tee $out/r04_stress.txt
timeout 300 python scripts/stress_cluster.py cfg1 60 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/r04_stress.txt
timeout 300 python scripts/stress_cluster.py cfg4 60 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/r04_stress.txt
for c in cfg1 cfg4 cfg3; do timeout 300 python scripts/phase_spread.py $c 8 randn graph 30 > $out/r04_phase_graph_final_$c.txt 2>&1; done
timeout 300 python scripts/phase_spread.py cfg1 8 clustered graph 30 > $out/r04_phase_graph_final_cfg1_clustered.txt 2>&1
echo "done t=$(( $(date +%s) - t0 ))"
