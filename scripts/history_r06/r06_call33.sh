#!/bin/bash
# round 6, call 33: MFMA-busy counters of the final build's key SimHash in a bench run (as profiles/archive/r05_pmc_mfma_cfg1.md)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && rm -rf pmc_mfma && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/pmc_mfma -- python $root/bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
db=$(find /tmp/pmc_mfma -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 (r06, final build)"; python scripts/rocprof_pmc.py $db ""; } > $out/r06_pmc_mfma_cfg1.md 2>&1
grep "simhash_keys" $out/r06_pmc_mfma_cfg1.md | cut -c1-160
