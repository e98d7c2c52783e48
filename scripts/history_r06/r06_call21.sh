#!/bin/bash
# round 6, call 21: what the key SimHash kernel waits for -- SQ counters, separate passes (--kernel-trace + --pmc only)
out=$(pwd)/gpurun_out; mkdir -p $out; root=$(pwd)
export TMPDIR=/tmp
cd /tmp
pass(){ tag=$1; shift
  rm -rf /tmp/pmc_$tag; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -- python $root/scripts/key_hash_time.py --worker product 0 > /tmp/pmc_$tag.log 2>&1
  db=$(find /tmp/pmc_$tag -name "*results.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $* -- python scripts/key_hash_time.py --worker product 0   (cfg 1: 8 kv heads x 97 932 keys, then cfg 4's share: 1 x 131 004; 24 launches each)"; python $root/scripts/rocprof_pmc.py $db simhash_keys; } >> $out/r06v_pmc_key_hash.md 2>&1
  tail -2 /tmp/pmc_$tag.log
}
rm -f $out/r06v_pmc_key_hash.md
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU
pass d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass e SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU
pass f SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC
cat $out/r06v_pmc_key_hash.md | cut -c1-200
