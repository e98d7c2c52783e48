#!/bin/bash
# round 5, GPU call 1: full GPU suite on the round's first build (host-buffer pairing fix, one-trip stream at R = 1, bench.py legs),
# the driver-style default line with its legs, A/B of the kernel candidates in alternating regions (scripts/ab_libs.py), LDS counters at cfg 3
out=gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r05a_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -8 $out/r05a_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/r05a_bench_driver_style.json 2> $out/r05a_bench_driver_style.err; echo "driver-style rc=$? t=$(( $(date +%s) - t0 ))"; tail -5 $out/r05a_bench_driver_style.err
for v in ticket setprio ntst; do
  timeout 240 python scripts/ab_libs.py cfg1 product $v --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05a_ab_cfg1.txt
echo "ab cfg1 t=$(( $(date +%s) - t0 ))"
for c in cfg3 cfg2; do
  timeout 300 python scripts/ab_libs.py $c product stream2 --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05a_ab_stream.txt
timeout 300 python scripts/ab_libs.py cfg2 product stream2 --data clustered --reps 4 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $out/r05a_ab_stream.txt
echo "ab stream t=$(( $(date +%s) - t0 ))"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(Name|name).*(LDS|TCP_PENDING|TA_BUSY|TA_TA_BUSY)" | head -60) > $out/r05a_counters_avail.txt 2>&1
pmc(){ tag=$1; lib=$2; shift 2; grp="$*"
  (cd /tmp && rm -rf pmc_$tag && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$tag -- python $root/bench.py --config cfg3 $lib --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 > /dev/null 2>$out/r05a_pmc_$tag.err)
  db=$(find /tmp/pmc_$tag -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $grp -- python bench.py --config cfg3 $lib --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --no-graph --steps 8 --warmup 2 (r05a)"; python scripts/rocprof_pmc.py $db decode; } > $out/r05a_pmc_lds_${tag}_cfg3.md 2>&1
}
pmc A_product "" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
pmc A_stream2 "--lib magicpig_amd/lib/variants/stream2/libmagicpig_hip.so" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
pmc B_product "" SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
echo "done t=$(( $(date +%s) - t0 ))"
