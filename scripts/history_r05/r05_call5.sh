#!/bin/bash
# round 5, GPU call 5 (same passes after: guard-band path without LDS waits, prologue loads issued together, packed 16-bit tile counters in the table build = two workgroups per CU): the register-blocked key SimHash (4 plane sets per wave, 4-wave workgroups) and the table build with LDS-only barriers:
# full suite, then kernel statistics of a cfg 1 and a cfg 4 bench run (prefill kernels)
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r05g_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -6 $out/r05g_pytest.log
timeout 200 python scripts/prefill_phases.py 2>&1 | grep -v amdgpu.ids | tee $out/r05g_prefill_phases.txt
for c in cfg1 cfg4; do
(cd /tmp && rm -rf prof_$c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 > /dev/null 2>&1)
db=$(find /tmp/prof_$c -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 (r05g)"; python scripts/rocprof_stats.py $db; } > $out/r05g_kernel_stats_$c.md 2>&1
grep -E "simhash_keys|lsh_build|lsh_slots|lsh_subbounds|key_centre|lsh_decode" $out/r05g_kernel_stats_$c.md
done
echo "done t=$(( $(date +%s) - t0 ))"
