#!/bin/bash
# Pipelined C2 step (bench.py, no CPU baseline) under developer switches; one JSON line each, reduced to ms per step.
# usage: bash tools/step_ab.sh <tag> "<ENV=VAL ...>" "<ENV=VAL ...>" ...   ("-" = no switch)
V=$1; shift
O=gpurun_out/$V
mkdir -p $O
n=0
for cfg in "$@"; do
  n=$((n+1))
  if [ "$cfg" = "-" ]; then cfg=""; fi
  env $cfg timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/b$n.json 2> $O/b$n.err
  python3 - "$O/b$n.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["device_resident"]["kernel_ms_per_step"]
    print("%-45s step %.2f ms  %.3f Gbp/s | alone: fill %.2f trace %.2f | mode0 %.2f ms" % (
        sys.argv[2] or "(default)", d["ms_per_step"], d["value"] / 1e9, r["fill"], r["traceback"], d["other_pipe_mode"]["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
