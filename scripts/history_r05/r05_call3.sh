#!/bin/bash
# round 5, GPU call 3: full suite on the build with the LDS-ticket merge as default, the transposed 2-tile key SimHash and the batched
# table-build output; kernel statistics of a cfg 1 bench run (prefill kernels + decode); the default line again (host legs with per-rep
# statistics and fast-path counters); cluster stress with the ticket merge
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r05c_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -6 $out/r05c_pytest.log
timeout 200 python scripts/prefill_phases.py 2>&1 | grep -v amdgpu.ids | tee $out/r05c_prefill_phases.txt
(cd /tmp && rm -rf prof_c1 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -- python $root/bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs > /dev/null 2>&1)
db=$(find /tmp/prof_c1 -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs (r05c; 32 warm-up + 128 timed steps, hipGraph)"; python scripts/rocprof_stats.py $db; } > $out/r05c_kernel_stats_cfg1.md 2>&1
head -16 $out/r05c_kernel_stats_cfg1.md
echo "stats t=$(( $(date +%s) - t0 ))"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/r05c_bench_driver_style.json 2> $out/r05c_bench_driver_style.err; echo "driver-style rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/r05c_bench_driver_style.err
for c in cfg1 cfg4 cfg2; do timeout 300 python scripts/stress_cluster.py $c 60 2>&1 | grep -v amdgpu.ids; done | tee $out/r05c_stress.txt
timeout 300 python scripts/stress_cluster.py cfg1 40 contend 2>&1 | grep -v amdgpu.ids | tee -a $out/r05c_stress.txt
echo "done t=$(( $(date +%s) - t0 ))"
