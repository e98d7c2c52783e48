#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04f_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"
tail -12 $out/r04f_pytest.log
bash scripts/ab_multi.sh "r03base product" "cfg1 cfg1:clustered cfg2 cfg3 cfg4 cfg0" 2 > $out/r04f_ab.txt 2>&1
cat $out/r04f_ab.txt
echo "ab t=$(( $(date +%s) - t0 ))"
for rep in 1 2; do
  timeout 300 python scripts/host_mode_times.py cfg1 60 2>&1 | grep -v amdgpu | sed "s/^/new  /" | head -8
  MP_LIB=magicpig_amd/lib/variants/r03base/libmagicpig_hip.so timeout 300 python scripts/host_mode_times.py cfg1 60 2>&1 | grep -v amdgpu | sed "s/^/r03  /" | head -8
done > $out/r04f_host_mode.txt 2>&1
cat $out/r04f_host_mode.txt
echo "host t=$(( $(date +%s) - t0 ))"
MP_STRESS_BIG=1 MP_STRESS_SCENARIOS=live,free_first,stale MP_STRESS_GDB=1 timeout 900 python scripts/stress_host_register.py all 12 > $out/r04f_hostreg_big.txt 2>&1
grep -v "^\[Thread\|amd_mem_obj\|^\[New" $out/r04f_hostreg_big.txt | tail -80
echo "done t=$(( $(date +%s) - t0 ))"
