#!/bin/bash
# What 8 ranks ask of the host, measured on a 1-GPU box: `bench.py --gpus 8` with every rank on GPU 0
# (SD_BENCH_SHARE_GPU=1: gloo barrier / reductions on the host -- the data path has no collective either way).
# The GPU is shared 8 ways, so bp/s means nothing here; what counts is process_cpu_ms per step and rank, and that the
# launcher / barrier / reduction path of N > 1 runs.  usage (GPU box): bash tools/share_gpu_8ranks.sh <outdir> [reads per rank]
O=$PWD/${1:-gpurun_out/share8}; R=${2:-250}
mkdir -p "$O"
for n in 2 8; do
  SD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus $n --steps 6 --warmup 2 --reads $R --no-cpu-baseline --timed-only \
      > $O/share_gpus$n.json 2> $O/share_gpus$n.err
  python3 - "$O/share_gpus$n.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
h = j["host_ms_per_step"]
print("n_gpus=%d reads/rank=%d  ms/step %.1f  rank0: process_cpu_ms/step %.1f (pack %.1f, wait %.1f, assemble %.1f)  quota %s CPUs" % (
    j["n_gpus"], j["config"]["reads_per_gpu"], j["ms_per_step"], h["process_cpu_ms"], h["pack_upload_enqueue"], h["wait_for_device"],
    h["d2h_assemble"], h["cpu_quota_cores"]))
PY
done
SD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --scaling strong --config c3 --reads-total 600 --steps 6 --warmup 4 > $O/strong_c3_gpus2.json 2> $O/strong_c3.err
SD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --scaling strong --config c5 --seq-len 20000000 --steps 6 --warmup 4 > $O/strong_c5_gpus2.json 2> $O/strong_c5.err
timeout 900 python bench.py --scaling strong --config c3 --reads-total 2000 --steps 6 --warmup 4 > $O/strong_c3_gpus1.json 2>> $O/strong_c3.err
timeout 900 python bench.py --scaling strong --config c5 --seq-len 50000000 --steps 6 --warmup 4 > $O/strong_c5_gpus1.json 2>> $O/strong_c5.err
for f in $O/strong_*.json; do python3 -c "
import json,sys; j=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], j['n_gpus'], j['scaling'], '%.2f Gbp/s' % (j['value']/1e9), '%.1f ms/step' % j['ms_per_step'], j['config']['share'])"; done
for e in $O/*.err; do tail -n 3 $e; done
