#!/bin/bash
# round 6, call 23: host-mode speculation with nothing in front of the retrieve kernel (query snapshot in the store's pinned block,
# norms on the host for few heads)
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_host_speculation.py tests/test_gpu_parity.py tests/test_gpu_alloc_ex.py tests/test_gpu_c_client.py -q -x -p no:cacheprovider > $out/r06x_pytest_host.log 2>&1; echo "pytest rc=$?"; tail -3 $out/r06x_pytest_host.log
for i in 1 2; do python scripts/host_mode_times.py cfg1 2>&1 | grep -v amdgpu.ids; done | tee $out/r06x_host_mode.txt
python scripts/host_mode_times.py cfg2 2>&1 | grep -v amdgpu.ids | tee -a $out/r06x_host_mode.txt
python scripts/host_mode_times.py cfg4 2>&1 | grep -v amdgpu.ids | tee -a $out/r06x_host_mode.txt
