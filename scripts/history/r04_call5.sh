#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04e_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"
tail -12 $out/r04e_pytest.log
for rep in 1 2; do
  timeout 300 python scripts/host_mode_times.py cfg1 60 2>&1 | grep -v amdgpu | sed "s/^/new  /" | head -8
  MP_OPTIONS=host_flag_wait=1 timeout 300 python scripts/host_mode_times.py cfg1 60 2>&1 | grep -v amdgpu | sed "s/^/flag /" | head -8
  MP_LIB=magicpig_amd/lib/variants/r03base/libmagicpig_hip.so timeout 300 python scripts/host_mode_times.py cfg1 60 2>&1 | grep -v amdgpu | sed "s/^/r03  /" | head -8
done > $out/r04e_host_mode.txt 2>&1
cat $out/r04e_host_mode.txt
echo "host t=$(( $(date +%s) - t0 ))"
bash scripts/ab_multi.sh "r03base product" "cfg1 cfg1:clustered cfg4" 2 > $out/r04e_ab.txt 2>&1
cat $out/r04e_ab.txt
echo "done t=$(( $(date +%s) - t0 ))"
