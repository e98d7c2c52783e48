#!/bin/bash
out=gpurun_out; mkdir -p $out
for rep in 1 2; do
  echo "== r04head"; MP_LIB=magicpig_amd/lib/variants/r04head/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
  echo "== new";  timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
  echo "== new, host_copy_threads=0"; MP_OPTIONS=host_copy_threads=0 timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
done 2>&1 | tee $out/r04q_host_mode.txt
echo "== instrumented, threads 3"; MP_HOST_TIMES=1 MP_LIB=magicpig_amd/lib/variants/hosttimes/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -E "host_times|host buffers" | tee -a $out/r04q_host_mode.txt
echo "== instrumented, threads 0"; MP_OPTIONS=host_copy_threads=0 MP_HOST_TIMES=1 MP_LIB=magicpig_amd/lib/variants/hosttimes/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -E "host_times|host buffers" | tee -a $out/r04q_host_mode.txt
echo "== instrumented, threads 1"; MP_OPTIONS=host_copy_threads=1 MP_HOST_TIMES=1 MP_LIB=magicpig_amd/lib/variants/hosttimes/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -E "host_times|host buffers" | tee -a $out/r04q_host_mode.txt
echo "== r04head cfg2"; MP_LIB=magicpig_amd/lib/variants/r04head/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg2 60 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $out/r04q_host_mode.txt
echo "== new cfg2"; timeout 100 python scripts/host_mode_times.py cfg2 60 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $out/r04q_host_mode.txt
