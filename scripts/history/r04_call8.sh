#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash scripts/ab_multi.sh "r04prev product" "cfg2 cfg3 cfg2:clustered cfg1" 3 > $out/r04h_ab.txt 2>&1
cat $out/r04h_ab.txt
echo "ab t=$(( $(date +%s) - t0 ))"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04h_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"
tail -6 $out/r04h_pytest.log
