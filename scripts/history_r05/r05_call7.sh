#!/bin/bash
# round 5, GPU call 7: LDS-only barriers on the decode chain (MP_LDS_BARRIERS) against __syncthreads() (variant nolb), alternating regions
out=$(pwd)/gpurun_out; mkdir -p $out
for c in cfg1 cfg3 cfg4 cfg0 cfg2; do
  timeout 300 python scripts/ab_libs.py $c nolb product --reps 6 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05i_ab_lds_barriers.txt
timeout 300 python scripts/ab_libs.py cfg1 nolb product --data clustered --reps 5 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $out/r05i_ab_lds_barriers.txt
timeout 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_kn_payload.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
