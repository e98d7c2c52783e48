#!/bin/bash
# final measurement pass of round 4 on the final build: bench lines + rocprofv3 kernel stats for every workload, full suite, stress
out=gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
run(){ c=$1; data=$2; suf=$3
  timeout 900 python bench.py --config $c $data > $out/r04_bench_$c$suf.json 2> $out/r04_bench_$c$suf.err
  (cd /tmp && rm -rf prof_$c$suf && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c$suf -- python $root/bench.py --config $c $data --no-cpu-baseline --no-host-mode --no-clustered-leg > /dev/null 2>&1)
  db=$(find /tmp/prof_$c$suf -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $c $data --no-cpu-baseline --no-host-mode --no-clustered-leg (r04 final build; 32 warm-up + 128 timed steps, hipGraph)"; python scripts/rocprof_stats.py $db; } > $out/r04_kernel_stats_$c$suf.md 2>&1
}
run cfg1 "" ""
run cfg1 "--data clustered" _clustered
run cfg2 "" ""
run cfg2 "--data clustered" _clustered
run cfg3 "" ""
run cfg4 "" ""
timeout 600 python bench.py --config cfg0 > $out/r04_bench_cfg0.json 2> $out/r04_bench_cfg0.err
echo "bench+stats t=$(( $(date +%s) - t0 ))"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/r04_bench_driver_style.json 2> $out/r04_bench_driver_style.err; tail -4 $out/r04_bench_driver_style.err
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04m_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -6 $out/r04m_pytest.log
for c in cfg1 cfg4 cfg2; do timeout 300 python scripts/stress_cluster.py $c 100 2>&1 | grep -v amdgpu.ids; done | tee $out/r04_stress.txt
timeout 300 python scripts/stress_cluster.py cfg1 60 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/r04_stress.txt
timeout 300 python scripts/stress_cluster.py cfg4 60 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/r04_stress.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
for c in cfg1 cfg4 cfg3; do timeout 300 python scripts/phase_spread.py $c 8 randn graph 30 > $out/r04_phase_graph_final_$c.txt 2>&1; done
timeout 300 python scripts/phase_spread.py cfg1 8 clustered graph 30 > $out/r04_phase_graph_final_cfg1_clustered.txt 2>&1
echo "done t=$(( $(date +%s) - t0 ))"
