#!/bin/bash
# round 6, call 34: key_colsum_kernel with eight loads in flight: parity of the offload fill + rocprofv3 averages in a bench run
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_decode_harness.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
(cd /tmp && rm -rf prof_c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $root/bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs > /dev/null 2>&1)
db=$(find /tmp/prof_c -name "*results.db" | head -1)
python scripts/rocprof_stats.py $db | grep "key_colsum\|key_centre\|simhash_keys\|lsh_build_kernel\|lsh_slots" | cut -c1-130
