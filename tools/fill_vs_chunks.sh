#!/bin/bash
# Fill time against the number of chunks of a launch (GPU box): bench.py --reads N for N around the multiples of the
# 4096 resident waves (409.6 reads of 10 chunks) -> how much the last, ragged round of chunks costs.
# usage: bash tools/fill_vs_chunks.sh [monomers] > profiles/...
M=${1:-12}
for n in 100 200 300 400 410 420 450 500 600 700 800 819 830 900 1000 1100 1200 1229 1240 1300 1600 1640 2000; do
  timeout 300 python bench.py --no-cpu-baseline --monomers $M --reads $n --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; c=d['config']
print('reads %5d chunks %6d  fill alone %7.3f ms  = %.3f us per chunk  traceback alone %6.3f ms  step %7.3f ms' % ($n, c['chunks_per_gpu'], r['isolated_avg_launch_ms'], 1e3*r['isolated_avg_launch_ms']/c['chunks_per_gpu'], d['device_resident']['kernel_ms_per_step']['traceback'], d['ms_per_step']))"
done
echo "# uniform chunks of 5000 rows, multiples of the 4096 resident waves"
for n in 2048 4096 8192 12288 16384; do
  timeout 300 python bench.py --no-cpu-baseline --monomers $M --reads $n --read-len 5000 --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; c=d['config']
print('chunks %6d x 5000 rows  fill alone %7.3f ms  traceback alone %6.3f ms' % (c['chunks_per_gpu'], r['isolated_avg_launch_ms'], d['device_resident']['kernel_ms_per_step']['traceback']))"
done
