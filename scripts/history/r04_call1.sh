#!/bin/bash
# round 4, GPU call 1: parity of the wave-owned gather (full -m gpu suite), replay stress, A/B against the round-3 library,
# phase stamps of the new chain, and the host_register abort hunt.  Everything into gpurun_out/.
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $out/r04a_pytest.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - t0 ))" | tee -a $out/r04a_summary.txt
tail -5 $out/r04a_pytest.log | tee -a $out/r04a_summary.txt
for c in cfg1 cfg4 cfg2; do
  timeout 300 python scripts/stress_cluster.py $c 60 >> $out/r04a_stress.txt 2>&1
done
timeout 300 python scripts/stress_cluster.py cfg1 40 contend >> $out/r04a_stress.txt 2>&1
cat $out/r04a_stress.txt | grep -v amdgpu.ids | tee -a $out/r04a_summary.txt
echo "stress t=$(( $(date +%s) - t0 ))" | tee -a $out/r04a_summary.txt
# A/B: product (new) vs r03base, alternating, two runs each
timeout 1500 bash scripts/ab_lib.sh r03base cfg1 cfg1:clustered cfg2 cfg3 cfg4 cfg0 cfg2:clustered > $out/r04a_ab.txt 2>&1
cat $out/r04a_ab.txt | tee -a $out/r04a_summary.txt
echo "ab t=$(( $(date +%s) - t0 ))" | tee -a $out/r04a_summary.txt
for c in cfg1 cfg2; do
  timeout 200 python scripts/phase_spread.py $c 10 randn > $out/r04a_phase_$c.txt 2>&1
done
echo "phase t=$(( $(date +%s) - t0 ))" | tee -a $out/r04a_summary.txt
MP_STRESS_GDB=1 timeout 1200 python scripts/stress_host_register.py all 40 > $out/r04a_hostreg.txt 2>&1
tail -60 $out/r04a_hostreg.txt
echo "hostreg t=$(( $(date +%s) - t0 ))" | tee -a $out/r04a_summary.txt
