#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash scripts/ab_multi.sh "r04state product" "cfg1 cfg1:clustered cfg4 cfg2 cfg3 cfg0" 3 > $out/r04l_ab.txt 2>&1
cat $out/r04l_ab.txt
echo "ab t=$(( $(date +%s) - t0 ))"
