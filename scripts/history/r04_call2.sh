#!/bin/bash
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash scripts/ab_multi.sh "r03base vfence voff32 vknlds vstate vnostore vall5 product" "cfg1 cfg1:clustered cfg2 cfg3" 2 > $out/r04b_ab_micro.txt 2>&1
cat $out/r04b_ab_micro.txt
echo "micro t=$(( $(date +%s) - t0 ))"
bash scripts/ab_multi.sh "r03base product" "cfg1 cfg3" 1 "--distinct-layers 2" > $out/r04b_distinct2.txt 2>&1
bash scripts/ab_multi.sh "r03base product" "cfg1" 1 "--distinct-layers 8" >> $out/r04b_distinct2.txt 2>&1
cat $out/r04b_distinct2.txt
echo "distinct t=$(( $(date +%s) - t0 ))"
for c in cfg1 cfg3; do
  MP_LIB=magicpig_amd/lib/variants/r03stamps/libmagicpig_hip.so timeout 300 python scripts/phase_spread.py $c 8 randn graph 30 > $out/r04b_phase_graph_${c}_r03.txt 2>&1
  timeout 300 python scripts/phase_spread.py $c 8 randn graph 30 > $out/r04b_phase_graph_${c}_new.txt 2>&1
done
MP_LIB=magicpig_amd/lib/variants/r03stamps/libmagicpig_hip.so timeout 300 python scripts/phase_spread.py cfg1 8 randn > $out/r04b_phase_eager_cfg1_r03.txt 2>&1
grep -v Warn $out/r04b_phase_graph_cfg1_r03.txt | tail -24
grep -v Warn $out/r04b_phase_graph_cfg1_new.txt | tail -24
echo "phase t=$(( $(date +%s) - t0 ))"
