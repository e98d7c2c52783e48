#!/bin/bash
# R4-15, the round's last GPU seconds: non-temporal loads of slots / bucket records / table ids (same schedule, 121 loads carry `nt`)
out=gpurun_out; mkdir -p $out
timeout 40 bash scripts/ab_multi.sh "product ntslots" "cfg1 cfg3 cfg4" 2 "--steps 64 --warmup 8" | tee $out/r04w_ab_nt.txt
