#!/bin/bash
# round 5, GPU call 8: packed table build at NB = 2048 (cfg 4) -- parity subset + kernel statistics
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "build or table or slot or golden or fixture or cfg4 or payload" 2>&1 | tail -3
for c in cfg4 cfg1; do
(cd /tmp && rm -rf prof_$c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 > /dev/null 2>&1)
db=$(find /tmp/prof_$c -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 (r05j)"; python scripts/rocprof_stats.py $db; } > $out/r05j_kernel_stats_$c.md 2>&1
grep -E "simhash_keys|lsh_build|lsh_slots|lsh_subbounds|key_centre|lsh_decode" $out/r05j_kernel_stats_$c.md
done
