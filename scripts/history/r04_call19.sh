#!/bin/bash
out=gpurun_out; mkdir -p $out
for rep in 1 2; do
  echo "== sync";  timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
  echo "== query spin"; MP_OPTIONS=host_flag_wait=2 timeout 100 python scripts/host_mode_times.py cfg1 100 2>&1 | grep -v amdgpu.ids | head -7
done 2>&1 | tee $out/r04s_host_mode.txt
