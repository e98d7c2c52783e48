#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04d_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"
tail -12 $out/r04d_pytest.log
bash scripts/ab_multi.sh "r03base product" "cfg1 cfg1:clustered cfg2" 2 > $out/r04d_ab.txt 2>&1
cat $out/r04d_ab.txt
for r in 16 32; do bash scripts/ab_multi.sh "product" "cfg4" 2 "--cluster $r" | sed "s/product/product-R$r/"; done > $out/r04d_ab_cfg4.txt 2>&1
cat $out/r04d_ab_cfg4.txt
echo "ab t=$(( $(date +%s) - t0 ))"
timeout 300 python scripts/host_mode_times.py cfg1 60 > $out/r04d_host_mode_cfg1.txt 2>&1; grep -v amdgpu $out/r04d_host_mode_cfg1.txt | tail -25
MP_LIB=magicpig_amd/lib/variants/r03base/libmagicpig_hip.so timeout 300 python scripts/host_mode_times.py cfg1 60 > $out/r04d_host_mode_cfg1_r03.txt 2>&1; grep -v amdgpu $out/r04d_host_mode_cfg1_r03.txt | tail -25
echo "host t=$(( $(date +%s) - t0 ))"
timeout 900 python scripts/stress_host_register.py all 150 > $out/r04d_hostreg.txt 2>&1
grep -v "^\[Thread\|amd_mem_obj" $out/r04d_hostreg.txt | tail -40
for c in cfg4 cfg1; do timeout 300 python scripts/stress_cluster.py $c 40 2>&1 | grep -v amdgpu.ids; done
echo "done t=$(( $(date +%s) - t0 ))"
