#!/bin/bash
# alternating bench.py runs of several builds of the library on one box:
#   scripts/ab_multi.sh "<lib names: product | variant dir names>" "<cfg[:data] ...>" [reps] [extra bench args]
# prints: cfg lib us_per_layer(step / layers) avg_launch_us(HIP events)
libs=$1; cfgs=$2; reps=${3:-2}; extra=$4
for c in $cfgs; do
  for rep in $(seq $reps); do
    for lib in $libs; do
      if [ $lib = product ]; then L=""; else L="--lib magicpig_amd/lib/variants/$lib/libmagicpig_hip.so"; fi
      python bench.py --config ${c%%:*} $( [ "${c#*:}" != "$c" ] && echo "--data ${c#*:}" ) $L --no-cpu-baseline --no-host-mode --no-clustered-leg --no-legs $extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '$lib', round(d['sparse_attn_us_per_layer'],2), round(d['roofline']['avg_launch_us'],2))" 2>/dev/null || echo "$c $lib FAILED"
    done
  done
done
