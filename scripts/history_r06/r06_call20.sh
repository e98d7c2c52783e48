#!/bin/bash
# round 6, call 20: persistent key-SimHash workgroups + row-per-lane code cut + mask-form guard-band queue: parity and timing
out=gpurun_out; mkdir -p $out
python scripts/key_hash_time.py build product skold 2>&1 | grep -v amdgpu.ids | tee $out/r06u_key_hash_persistent.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/r06u_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/r06u_pytest.log
