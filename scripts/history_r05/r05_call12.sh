#!/bin/bash
# round 5, GPU call 12: tile-size choice of the table build at NB = 2048 (cfg 4) -- parity subset, kernel statistics
root=$(pwd); export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "build or table or slot or cfg4 or bounds or fill or ranking" 2>&1 | tail -2
(cd /tmp && rm -rf prof_k4 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k4 -- python $root/bench.py --config cfg4 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 8 --warmup 2 > /dev/null 2>&1)
python scripts/rocprof_stats.py $(find /tmp/prof_k4 -name "*results.db" | head -1) | grep -E "simhash_keys|lsh_build|lsh_slots"
