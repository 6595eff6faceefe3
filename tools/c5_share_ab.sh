#!/bin/bash
# A/B of the 25-Mb share of BASELINE config 5 (what a rank gets at 8 GPUs) under different cuts into device batches
# (SD_BATCH_ROWS = row budget per batch; the share holds 27.5 M rows).  Usage: tools/c5_share_ab.sh <outdir> [rows ...]
out=${1:-gpurun_out/c5ab}; shift
mkdir -p $out
for rows in default "$@"; do
  if [ "$rows" = default ]; then unset SD_BATCH_ROWS; else export SD_BATCH_ROWS=$rows; fi
  SD_TIMING=1 python bench.py --config c5 --scaling strong --seq-len 25000000 --steps 10 --warmup 2 --no-cpu-baseline \
      > $out/rows_$rows.json 2> $out/rows_$rows.err
  echo "rows=$rows $(python -c "import json,sys; j=json.loads(open('$out/rows_$rows.json').read().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3))")"
  grep "sd timing" $out/rows_$rows.err | tail -2
done
