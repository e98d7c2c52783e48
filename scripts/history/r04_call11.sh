#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash scripts/ab_multi.sh "r04prev product" "cfg1 cfg1:clustered cfg4 cfg2 cfg3" 3 > $out/r04k_ab.txt 2>&1
cat $out/r04k_ab.txt
echo "ab t=$(( $(date +%s) - t0 ))"
timeout 900 python -m pytest tests/test_gpu_kn_payload.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $out/r04k_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -5 $out/r04k_pytest.log
