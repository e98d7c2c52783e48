#!/bin/bash
out=gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_decode_harness.py tests/test_gpu_two_ranks.py -m gpu -q -p no:cacheprovider > $out/r04j_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -15 $out/r04j_pytest.log
c=cfg2
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rm -rf pmc_${c}_cl_$ctr && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${c}_cl_$ctr -- python $root/bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-graph --steps 8 --warmup 2 > /tmp/pmc_${c}_cl_$ctr.log 2>&1; echo "rocprof rc=$?"; tail -3 /tmp/pmc_${c}_cl_$ctr.log)
  db=$(find /tmp/pmc_${c}_cl_$ctr -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-graph --steps 8 --warmup 2 (r04); KB per dispatch as reported (FETCH_SIZE x 2 on gfx950)"; python scripts/rocprof_pmc.py $db decode; } > $out/r04_pmc_${ctr}_${c}_clustered.md 2>&1
done
echo "pmc t=$(( $(date +%s) - t0 ))"
python scripts/merge_pmc_traffic.py r04 $out
timeout 900 python bench.py --config cfg2 --data clustered > $out/r04_bench_cfg2_clustered.json 2> $out/r04_bench_cfg2_clustered.err
echo "bench t=$(( $(date +%s) - t0 ))"
timeout 900 python bench.py --end-to-end --config cfg1 > $out/r04_bench_e2e_8b.json 2> $out/r04_bench_e2e_8b.err; tail -1 $out/r04_bench_e2e_8b.json | cut -c1-400
timeout 1200 python bench.py --end-to-end --config cfg4 --shard head --emulate-rank 0/8 > $out/r04_bench_e2e_70b_tp8_rank0.json 2> $out/r04_bench_e2e_70b_tp8_rank0.err; tail -1 $out/r04_bench_e2e_70b_tp8_rank0.json | cut -c1-500; tail -3 $out/r04_bench_e2e_70b_tp8_rank0.err
echo "done t=$(( $(date +%s) - t0 ))"
