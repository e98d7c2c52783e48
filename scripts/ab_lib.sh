V=$1; shift
for c in "$@"; do
  for rep in 1 2; do
    for lib in base $V; do
      if [ $lib = base ]; then L=""; else L="--lib magicpig_amd/lib/variants/$V/libmagicpig_hip.so"; fi
      set -- $c
      python bench.py --config ${c%%:*} $( [ "${c#*:}" != "$c" ] && echo "--data ${c#*:}" ) $L --no-cpu-baseline --no-host-mode --no-clustered-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '$lib', round(d['sparse_attn_us_per_layer'],2), round(d['roofline']['avg_launch_us'],2))"
    done
  done
done
