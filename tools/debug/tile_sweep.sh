#!/bin/bash
# headline d_msm (2^20, window table) against the tile length of k_accum_tiles (entries per lane): wave-quantisation of the launch
for t in "$@"; do
  v=$(ZKHIP_TUNE=msm_tile=$t python bench.py --no-cpu --no-extra --no-e2e --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4e  %.3f ms  accum %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))")
  echo "msm_tile=$t  $v"
done
