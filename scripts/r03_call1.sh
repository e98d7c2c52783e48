#!/bin/bash
# round 3, GPU call 1: full gpu test suite, bench lines on the three key distributions, phase stamps at cfg 0 / cfg 1
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r03a_pytest.log 2>&1; echo "pytest rc=$?" >> $out/r03a_pytest.log
tail -3 $out/r03a_pytest.log
for d in randn clustered skewed; do
  timeout 600 python bench.py --config cfg1 --data $d > $out/r03a_bench_cfg1_$d.json 2> $out/r03a_bench_cfg1_$d.err
done
timeout 600 python bench.py --config cfg2 --data clustered > $out/r03a_bench_cfg2_clustered.json 2> $out/r03a_bench_cfg2_clustered.err
timeout 300 python scripts/phase_spread.py cfg0 > $out/r03a_phase_cfg0.txt 2>&1
timeout 300 python scripts/phase_spread.py cfg1 > $out/r03a_phase_cfg1.txt 2>&1
ls -la $out | tail -12
