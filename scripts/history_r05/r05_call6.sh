#!/bin/bash
# round 5, GPU call 6: prefill kernels again (packed table build with 128 VGPRs, slots kernel with 8 pieces in flight + nt stores),
# the footprint table of VERDICT r04 item 6 (direct-slot widths at cfg 1, B = 1 and B = 4), the end-to-end harness at B = 8
out=$(pwd)/gpurun_out; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "build or table or slot or simhash or golden or fixture or direct or payload" > $out/r05h_pytest_subset.log 2>&1
echo "pytest subset rc=$? t=$(( $(date +%s) - t0 ))"; tail -3 $out/r05h_pytest_subset.log
(cd /tmp && rm -rf prof_c1 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -- python $root/bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 > /dev/null 2>&1)
db=$(find /tmp/prof_c1 -name "*results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config cfg1 --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs --steps 16 --warmup 4 (r05h)"; python scripts/rocprof_stats.py $db; } > $out/r05h_kernel_stats_cfg1.md 2>&1
grep -E "simhash_keys|lsh_build|lsh_slots|lsh_subbounds|key_centre|lsh_decode" $out/r05h_kernel_stats_cfg1.md
echo "stats t=$(( $(date +%s) - t0 ))"
for side in "product@--direct-slots,0" "product@--slot-log2,4" "product@--slot-log2,3"; do
  timeout 300 python scripts/ab_libs.py cfg1 product "$side" --reps 5 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $out/r05h_ab_slot_widths_cfg1.txt
for side in "product@--direct-slots,0" "product@--slot-log2,4"; do
  timeout 300 python scripts/ab_libs.py cfg1 product "$side" --data clustered --reps 4 --steps 64 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
done | tee -a $out/r05h_ab_slot_widths_cfg1.txt
B4='{"model":"Llama-3.1-8B (15 of 30 sparse layers)","layers":17,"dense":[0,16],"H":32,"Hkv":8,"D":128,"B":4,"P":98000,"M":98304,"K":10,"L":150}'
timeout 400 python scripts/ab_libs.py "$B4" product "product@--direct-slots,1,--slot-log2,5" "product@--direct-slots,1,--slot-log2,4" --reps 4 --steps 32 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1 | tee $out/r05h_ab_slot_widths_cfg1_B4.txt
echo "slots t=$(( $(date +%s) - t0 ))"
timeout 400 python bench.py --end-to-end --config cfg2 --steps 16 --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 | tee $out/r05h_bench_e2e_8b_B8.json
timeout 400 python bench.py --end-to-end --config cfg1 --steps 16 --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 | tee $out/r05h_bench_e2e_8b_B1.json
echo "done t=$(( $(date +%s) - t0 ))"
