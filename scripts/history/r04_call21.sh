#!/bin/bash
# the final build (decode kernel ISA-identical to call 13's; the retrieve's second count copy travels through part_cnt): host-mode tests,
# then the bench lines in priority order under one time guard
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "host" > $out/r04u_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -2 $out/r04u_pytest.log
line(){ c=$1; data=$2; suf=$3
  [ $(( $(date +%s) - t0 )) -gt 300 ] && { echo "skipped $c$suf (time guard)"; return; }
  timeout 80 python bench.py --config $c $data > $out/r04f_bench_$c$suf.json 2> $out/r04f_bench_$c$suf.err; echo "$c$suf rc=$? t=$(( $(date +%s) - t0 ))"
}
line cfg1 "" ""
[ $(( $(date +%s) - t0 )) -lt 300 ] && timeout 80 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04f_bench_driver_style.json 2> $out/r04f_bench_driver_style.err
line cfg4 "" ""
line cfg3 "" ""
line cfg2 "" ""
line cfg1 "--data clustered" _clustered
line cfg2 "--data clustered" _clustered
line cfg0 "" ""
echo "done t=$(( $(date +%s) - t0 ))"
