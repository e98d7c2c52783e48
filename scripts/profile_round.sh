#!/bin/bash
# Profile passes of one round on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh <tag> [configs...]      e.g. scripts/profile_round.sh r02 cfg1 cfg2
# For every config: a bench line (with the CPU leg), rocprofv3 --kernel-trace --stats of the same command, and
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE; SQ instruction / busy counters for cfg1 and cfg2) of an eager run.
# Everything lands in gpurun_out/; copy what should be judged into profiles/.
tag=${1:-rXX}; shift
cfgs=${@:-cfg1 cfg2 cfg3 cfg4}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for c in $cfgs; do
  timeout 900 python bench.py --config $c > $out/${tag}_bench_$c.json 2> $out/${tag}_bench_$c.err
  (cd /tmp && rm -rf prof_$c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs > /dev/null 2>&1)
  db=$(find /tmp/prof_$c -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline ($tag; 32 warm-up + 128 timed steps, hipGraph)"; python scripts/rocprof_stats.py $db; } > $out/${tag}_kernel_stats_$c.md 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rm -rf pmc_${c}_$ctr && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${c}_$ctr -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
    db=$(find /tmp/pmc_${c}_$ctr -name "*results.db" | head -1)
    { echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 ($tag); KB per dispatch as reported (FETCH_SIZE x 2 on gfx950)"; python scripts/rocprof_pmc.py $db decode; } > $out/${tag}_pmc_${ctr}_$c.md 2>&1
  done
done
for c in cfg1 cfg2; do
  case " $cfgs " in *" $c "*) ;; *) continue;; esac
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"; do
    n=$(echo $grp | cut -d' ' -f1)
    (cd /tmp && rm -rf sq_${c}_$n && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d /tmp/sq_${c}_$n -- python $root/bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
    db=$(find /tmp/sq_${c}_$n -name "*results.db" | head -1)
    { echo "# rocprofv3 --kernel-trace --pmc $grp -- python bench.py --config $c --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 ($tag)"; python scripts/rocprof_pmc.py $db decode; } > $out/${tag}_pmc_sq_${n}_$c.md 2>&1
  done
done
# the non-isotropic workload: bench line with the CPU leg, kernel stats and HBM traffic for cfg 1 / cfg 2
for c in cfg1 cfg2; do
  case " $cfgs " in *" $c "*) ;; *) continue;; esac
  timeout 900 python bench.py --config $c --data clustered > $out/${tag}_bench_${c}_clustered.json 2> $out/${tag}_bench_${c}_clustered.err
  (cd /tmp && rm -rf prof_${c}_cl && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${c}_cl -- python $root/bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs > /dev/null 2>&1)
  db=$(find /tmp/prof_${c}_cl -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs ($tag; 32 warm-up + 128 timed steps, hipGraph)"; python scripts/rocprof_stats.py $db; } > $out/${tag}_kernel_stats_${c}_clustered.md 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rm -rf pmc_${c}_cl_$ctr && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${c}_cl_$ctr -- python $root/bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>&1)
    db=$(find /tmp/pmc_${c}_cl_$ctr -name "*results.db" | head -1)
    { echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --config $c --data clustered --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 ($tag); KB per dispatch as reported (FETCH_SIZE x 2 on gfx950)"; python scripts/rocprof_pmc.py $db decode; } > $out/${tag}_pmc_${ctr}_${c}_clustered.md 2>&1
  done
done
timeout 600 python bench.py --config cfg0 > $out/${tag}_bench_cfg0.json 2> $out/${tag}_bench_cfg0.err
ls -la $out | tail -40
