#!/bin/bash
# round 5, GPU call 11: fast ranking in the table build -- parity subset (incl. fast == exact, no fallbacks), build routes compared, kernel statistics
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "build or table or slot or golden or fixture or cfg4 or cfg1 or payload or bounds or fill or ranking" 2>&1 | tail -3
timeout 200 python scripts/prefill_phases.py 2>&1 | grep -v amdgpu.ids | tee $out/r05r_prefill_phases.txt
bash scripts/r05_kstats.sh
(cd /tmp && rm -rf prof_k4 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k4 -- python $root/bench.py --config cfg4 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 8 --warmup 2 > /dev/null 2>&1)
python scripts/rocprof_stats.py $(find /tmp/prof_k4 -name "*results.db" | head -1) | grep -E "simhash_keys|lsh_build|lsh_slots"
python - <<'PY'
import magicpig_amd._lib as L
print("build_rank_fallbacks after this process's builds:", L.get_option("build_rank_fallbacks"))
PY
