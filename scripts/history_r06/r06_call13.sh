#!/bin/bash
out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_speculation.py tests/test_gpu_parity.py tests/test_gpu_c_client.py -m gpu -q -p no:cacheprovider -k "host or speculation or unchanged or launch_assumed or destroyed or client" > $out/r06m_pytest.log 2>&1
echo "pytest rc=$?"; tail -30 $out/r06m_pytest.log
for c in cfg1 cfg2; do
timeout 300 python scripts/host_mode_times.py $c 40 2>&1 | grep -v amdgpu.ids | tail -25 | tee -a $out/r06m_host_mode.txt
done
