#!/bin/bash
# HBM-side traffic of ONE call of each sumcheck-family primitive at 2^N (rocprofv3 --pmc, one counter per pass):
#   tools/sc_pmc.sh <tag> <log2 n>   -> gpurun_out/<tag>_sc_pmc_2pN.csv
# per call = sum of the counter over every zk:: kernel of the run / number of calls; FETCH_SIZE is reported
# raw AND doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B for wide coalesced reads).
set -u
TAG=$1; N=$2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RES=$OUT/${TAG}_sc_pmc_2p$N.csv
echo "primitive,log2n,algorithmic_bytes,FETCH_SIZE_bytes_raw,FETCH_SIZE_bytes_x2,WRITE_SIZE_bytes,traffic_over_algorithmic" > $RES
for MODE in product plain fold open; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/scp_$C
    SC_MODE=$MODE rocprofv3 --pmc $C -f csv -d /tmp/scp_$C -o pmc -- python $REPO/tools/sc_time.py $N > /tmp/scp_$C.out 2>/tmp/scp_$C.err
  done
  python - "$MODE" "$N" >> $RES <<'PY'
import csv, glob, sys
mode, n = sys.argv[1], int(sys.argv[2])
def total(c):
    f = glob.glob(f"/tmp/scp_{c}/**/*counter_collection.csv", recursive=True)[0]
    calls = int([l for l in open(f"/tmp/scp_{c}.out") if l.startswith("calls")][0].split()[2])
    kb = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("zk::k_pass", "zk::k_local", "zk::k_fold_flat", "zk::k_plain_flat")))
    return kb * 1024.0 / calls
fe, wr = total("FETCH_SIZE"), total("WRITE_SIZE")
alg = {"product": 64, "plain": 32, "fold": 32, "open": 64}[mode] * (1 << n)
print(f"{mode},{n},{alg},{fe:.0f},{2*fe:.0f},{wr:.0f},{(2*fe+wr)/alg:.3f}")
PY
done
cat $RES
