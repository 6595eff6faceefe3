#!/bin/bash
# Stream modes of the batch pipeline on C2, repeated: step time and the fill's HIP-event span per step.
# usage: bash tools/mode_ab.sh <tag> [repeats]
V=${1:-modeab}; N=${2:-3}
O=gpurun_out/$V
mkdir -p $O
for i in $(seq 1 $N); do
  for m in 2 1 0; do
    timeout 300 python bench.py --steps 20 --no-cpu-baseline --timed-only --pipe-mode $m > $O/m${m}_$i.json 2> $O/m${m}_$i.err
    python3 - $O/m${m}_$i.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernel_ms_per_step"]
print("mode %s: step %.2f ms  %.3f Gbp/s | spans: fill %.2f trace %.2f | hbm_notional %.3f" % (
    sys.argv[2], d["ms_per_step"], d["value"] / 1e9, k["fill"], k["traceback"], d["roofline"]["hbm_notional"]["frac"]))
PY
  done
done
