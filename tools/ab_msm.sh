#!/bin/bash
# A/B of two library builds inside ONE gpurun call (the pool's GPUs differ by up to 10 %): alternating runs
#   tools/ab_msm.sh build/libzkhip_old.so [rounds]     (B = the in-tree build)
A=$1; R=${2:-3}
for i in $(seq $R); do
  for L in "$A" ""; do
    ZKHIP_LIB=${L:+$(pwd)/$L} python bench.py --no-cpu --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${L:-in-tree}'.ljust(28), round(d['value']/1e8,3), {k:round(v,3) for k,v in d['msm_phase_ms'].items()})"
  done
done
