#!/bin/bash
# host-buffer mode: single optimistic attention launch (no relay, q | qn read from pinned memory, rows verified under the kernel)
# + helper threads for the mirror -> rows copy.  Tests first, then alternating timings against the shipped library.
out=gpurun_out; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c_client.py tests/test_host_copy_pool.py -q -x -p no:cacheprovider -k "host or c_client or copy or score" > $out/r04p_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/r04p_pytest.log
for rep in 1 2; do
  echo "== r04head"; MP_LIB=magicpig_amd/lib/variants/r04head/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
  echo "== new";  timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
  echo "== new, host_copy_threads=0"; MP_OPTIONS=host_copy_threads=0 timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
done 2>&1 | tee $out/r04p_host_mode.txt
echo "== new cfg2"; timeout 100 python scripts/host_mode_times.py cfg2 60 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $out/r04p_host_mode.txt
echo "== r04head cfg2"; MP_LIB=magicpig_amd/lib/variants/r04head/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg2 60 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $out/r04p_host_mode.txt
