#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_rccl.py -x -q > $out/r03c_rccl.log 2>&1; tail -5 $out/r03c_rccl.log
for a in "cfg0 10 randn" "cfg1 10 randn" "cfg1 10 clustered" "cfg2 6 randn"; do
  set -- $a
  timeout 300 python scripts/phase_spread.py $1 $2 $3 > $out/r03c_phase_$1_$3.txt 2>&1
done
cat $out/r03c_phase_cfg0_randn.txt $out/r03c_phase_cfg1_randn.txt
