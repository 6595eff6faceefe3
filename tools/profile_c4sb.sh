#!/bin/bash
# C4 --second-best leg: bench line, rocprofv3 kernel stats, PMC passes of the identity kernel.
# usage (GPU box): bash tools/profile_c4sb.sh <outdir>
O=$PWD/${1:-gpurun_out/c4sb}
mkdir -p "$O"
R=$PWD
export TMPDIR=/tmp
timeout 600 python bench.py --config c4-second-best --steps 5 --warmup 2 > $O/bench_c4sb.json 2> $O/bench_c4sb.err
tail -c 600 $O/bench_c4sb.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --config c4-second-best --steps 3 --warmup 1 > $O/stats.log 2>&1)
for c in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -o p -- python3 $R/bench.py --config c4-second-best --steps 2 --warmup 1 > $O/pmc_$n.log 2>&1)
done
find $O -name "*.csv" | wc -l
