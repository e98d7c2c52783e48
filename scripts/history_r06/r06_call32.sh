#!/bin/bash
# round 6, call 32: key SimHash with the rows' norms taken from the store (offload path): parity + rocprofv3 averages in a bench run
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_decode_harness.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for c in cfg1 cfg4; do
  (cd /tmp && rm -rf prof_$c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs > /dev/null 2>&1)
  db=$(find /tmp/prof_$c -name "*results.db" | head -1)
  python scripts/rocprof_stats.py $db | grep "simhash_keys\|lsh_decode" | cut -c1-150
done | tee $out/r06_key_hash_norms_from_store.txt
