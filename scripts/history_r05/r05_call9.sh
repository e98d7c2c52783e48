#!/bin/bash
# round 5, GPU call 9: sub-bounds written by the table build -- parity subset, the two build routes compared (prefill_phases), kernel statistics
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "build or table or slot or golden or fixture or cfg4 or cfg1 or payload or bounds or fill" 2>&1 | tail -3
timeout 200 python scripts/prefill_phases.py 2>&1 | grep -v amdgpu.ids | tee $out/r05k_prefill_phases.txt
for c in cfg4 cfg1; do
(cd /tmp && rm -rf prof_$c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 > /dev/null 2>&1)
db=$(find /tmp/prof_$c -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 (r05k)"; python scripts/rocprof_stats.py $db; } > $out/r05k_kernel_stats_$c.md 2>&1
grep -E "simhash_keys|lsh_build|lsh_slots|lsh_subbounds|key_centre|lsh_decode" $out/r05k_kernel_stats_$c.md
done
