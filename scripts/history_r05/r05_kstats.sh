root=$(pwd); export TMPDIR=/tmp
for i in 1 2; do
(cd /tmp && rm -rf prof_k && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -- python $root/bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 8 --warmup 2 > /dev/null 2>&1)
db=$(find /tmp/prof_k -name "*results.db" | head -1)
python scripts/rocprof_stats.py $db | grep -E "simhash_keys|lsh_build"
done
