#!/bin/bash
# round 5, GPU call 13: MFMA-busy counters of the final build (key SimHash = the one MFMA kernel; the decode kernel issues none)
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
(cd /tmp && rm -rf sq_m && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/sq_m -- python $root/bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
db=$(find /tmp/sq_m -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 (r05, final build)"; python scripts/rocprof_pmc.py $db "mp::" | grep -E "simhash_keys|lsh_decode|lsh_build|kernel \||---"; } > $out/r05_pmc_mfma_cfg1.md 2>&1
cat $out/r05_pmc_mfma_cfg1.md
