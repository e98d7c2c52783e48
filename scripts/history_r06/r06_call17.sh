#!/bin/bash
# round 6, item 4: cfg 4's share with the lean launch -- slot width and cluster size once more
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python scripts/ab_libs.py cfg4 product product@--slot-log2,3 product@--slot-log2,5 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06q_ab_cfg4.txt
timeout 600 python scripts/ab_libs.py cfg4 product product@--cluster,32 product@--cluster,8 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06q_ab_cfg4.txt
timeout 600 python scripts/ab_libs.py cfg4 product@--cluster,32 product@--cluster,32,--slot-log2,3 product@--direct-slots,0 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06q_ab_cfg4.txt
timeout 600 python scripts/ab_libs.py cfg1 product product@--cluster,16 product@--slot-log2,4 --reps 4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06q_ab_cfg4.txt
