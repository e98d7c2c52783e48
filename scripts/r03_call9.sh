#!/bin/bash
out=$(pwd)/gpurun_out; root=$(pwd); export TMPDIR=/tmp
for c in cfg1 cfg2; do
  (cd /tmp && rm -rf ic_$c && timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES -d /tmp/ic_$c -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
  db=$(find /tmp/ic_$c -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES -- python bench.py --config $c --no-cpu-baseline --no-host-mode --no-graph --steps 8 --warmup 2"; python scripts/rocprof_pmc.py $db decode; } > $out/r03j_pmc_icache_$c.md 2>&1
  cat $out/r03j_pmc_icache_$c.md
done
timeout 900 python -m pytest tests/test_gpu_decode_harness.py -x -q 2>&1 | tail -3
