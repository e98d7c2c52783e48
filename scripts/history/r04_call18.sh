#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c_client.py tests/test_gpu_configs.py -q -x -p no:cacheprovider -k "host or c_client or score or reference_test" > $out/r04r_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/r04r_pytest.log
for rep in 1 2; do
  echo "== r04head"; MP_LIB=magicpig_amd/lib/variants/r04head/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
  echo "== new";  timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
done 2>&1 | tee $out/r04r_host_mode.txt
for c in cfg2 cfg4; do
echo "== r04head $c"; MP_LIB=magicpig_amd/lib/variants/r04head/libmagicpig_hip.so timeout 100 python scripts/host_mode_times.py $c 60 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $out/r04r_host_mode.txt
echo "== new $c"; timeout 100 python scripts/host_mode_times.py $c 60 2>&1 | grep -v amdgpu.ids | head -7 | tee -a $out/r04r_host_mode.txt
done
