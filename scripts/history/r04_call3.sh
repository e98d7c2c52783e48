#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04c_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"
tail -25 $out/r04c_pytest.log
bash scripts/ab_multi.sh "r03base product" "cfg1 cfg1:clustered cfg2 cfg3 cfg0" 2 > $out/r04c_ab.txt 2>&1
cat $out/r04c_ab.txt
bash scripts/ab_multi.sh "r03base" "cfg4" 2 > $out/r04c_ab_cfg4.txt 2>&1
for r in 8 16 32; do bash scripts/ab_multi.sh "product" "cfg4" 2 "--cluster $r" | sed "s/product/product-R$r/" >> $out/r04c_ab_cfg4.txt 2>&1; done
cat $out/r04c_ab_cfg4.txt
echo "ab t=$(( $(date +%s) - t0 ))"
for c in cfg4 cfg1; do timeout 300 python scripts/stress_cluster.py $c 60 2>&1 | grep -v amdgpu.ids; done
timeout 300 python scripts/stress_cluster.py cfg4 40 contend 2>&1 | grep -v amdgpu.ids
timeout 200 python scripts/phase_spread.py cfg4 8 randn > $out/r04c_phase_cfg4.txt 2>&1; grep -v "Warn\|ureduce\|nanmedian\|amdgpu" $out/r04c_phase_cfg4.txt | tail -24
echo "done t=$(( $(date +%s) - t0 ))"
