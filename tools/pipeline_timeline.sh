#!/bin/bash
# Device / host timeline of the C2 stream with two and with three batches in flight (SD_TIMELINE=1: Pipeline::pop_fetch prints,
# per batch, begin / end of fill, traceback and compaction on the device clock and the host's enqueue / fetch times, ms since
# the pipeline's reference event).  usage: tools/pipeline_timeline.sh > profiles/r05_pipeline_timeline.txt   (on the GPU box)
for cfg in "2 1" "3 2"; do
  set -- $cfg
  echo "== SD_PIPE_SLOTS=$1 SD_BENCH_DEPTH=$2: python bench.py --steps 20 --warmup 4 --no-cpu-baseline --timed-only"
  SD_PIPE_SLOTS=$1 SD_BENCH_DEPTH=$2 SD_TIMELINE=1 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --timed-only > /tmp/tl_$1.json 2> /tmp/tl_$1.err
  python -c "import json; j=json.loads(open('/tmp/tl_$1.json').read().splitlines()[0]); print('ms_per_step', round(j['ms_per_step'],3))"
  grep "sd timeline" /tmp/tl_$1.err | sed -n 8,19p
done
