#!/bin/bash
# round 5, GPU call 2: the fixed host-fast-path test, host-buffer mode with pageable / pinned results line by line (the pinned leg of call 1
# read 296 us), direct slots at R = 1 (decode_direct = 2) against the sub-bounds path at cfg 3 / cfg 2, the LDS-ticket merge on the other
# configurations, LDS counters at cfg 3
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "host" -p no:cacheprovider > $out/r05b_pytest_host.log 2>&1
echo "pytest host rc=$? t=$(( $(date +%s) - t0 ))"; tail -4 $out/r05b_pytest_host.log
for mode in "" pinned; do
  echo "== host_mode_times cfg1 60 $mode"; timeout 200 python scripts/host_mode_times.py cfg1 60 $mode 2>&1 | grep -v amdgpu.ids
done | tee $out/r05b_host_mode.txt
echo "host t=$(( $(date +%s) - t0 ))"
for c in cfg3 cfg2; do
  timeout 300 python scripts/ab_libs.py $c product product@--direct-slots,2 --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05b_ab_slots_r1.txt
echo "slots t=$(( $(date +%s) - t0 ))"
for c in cfg4 cfg0 cfg3; do
  timeout 300 python scripts/ab_libs.py $c product ticket --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05b_ab_ticket.txt
timeout 300 python scripts/ab_libs.py cfg1 product ticket --data clustered --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $out/r05b_ab_ticket.txt
echo "ticket t=$(( $(date +%s) - t0 ))"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "LDS|TCP_PENDING|TA_BUSY" | head -80) > $out/r05b_counters_avail.txt 2>&1
pmc(){ tag=$1; lib=$2; shift 2; grp="$*"
  (cd /tmp && rm -rf pmc_$tag && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$tag -- python $root/bench.py --config cfg3 $lib --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>$out/r05b_pmc_$tag.err)
  db=$(find /tmp/pmc_$tag -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $grp -- python bench.py --config cfg3 $lib --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 (r05b)"; python scripts/rocprof_pmc.py $db decode; } > $out/r05b_pmc_lds_${tag}_cfg3.md 2>&1
}
pmc A "" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
pmc B "" SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc C "" SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_BUSY_CYCLES
echo "done t=$(( $(date +%s) - t0 ))"
