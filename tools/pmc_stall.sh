#!/bin/bash
# Where the waves of a fill kernel spend their cycles (GPU box, through gpurun): SQ wave / wait / active counters for
# the C4 DP shape (wide kernel) and C2 (narrow kernel).  usage: bash tools/pmc_stall.sh <tag> -> gpurun_out/<tag>/
V=${1:-stall}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
for w in c4 c2; do
  if [ $w = c4 ]; then A="--monomers 64 --reads 256"; else A=""; fi
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
    n=$(echo $set | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${w}_$n -o p -- python3 $R/bench.py $A --steps 2 --warmup 1 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/${w}_$n.log 2>&1)
  done
done
python3 - <<PY
import csv, glob, collections
for w in ("c4", "c2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s_*/**/*counter_collection.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fill" in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(w, k)
        for c, v in sorted(d.items()):
            v.sort()
            print("   %-22s %.4g  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
