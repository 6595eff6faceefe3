# Round-6 multi-GPU rehearsal on a 1-GPU box (every rank on GPU 0: what is checked is the launcher / barrier / reduction path,
# the CPU time a rank asks of the host, rows identical to one process, clean exit codes -- not bp/s):
#   weak C2 with 8 ranks x 250 reads x 2 host threads; strong C3 / C5 with 2 ranks; the config-5 job from records to file with
#   8 and 2 processes, every rank its own range against the rank-0 gather.
# usage (GPU box): bash tools/rehearse_r06.sh <outdir>
O=$PWD/${1:-gpurun_out/rehearse}; mkdir -p "$O"
bash tools/share_gpu_8ranks.sh "${1:-gpurun_out/rehearse}" 250 > $O/share8.txt 2>&1
tail -n 12 $O/share8.txt
P=29711
for n in 8 2; do
  P=$((P+1))   # (the tool runs both forms itself: every rank its own range, and the rank-0 gather)
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $P \
      tools/c5_ranks_files.py 200000000 > $O/c5_files_${n}.json 2> $O/c5_files_${n}.err
  echo "$n processes: rc=$? $(tail -c 500 $O/c5_files_${n}.json)"
done
# N = 1 form of the SCALE command == the BENCH form
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29741 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --timed-only > $O/scale_form_n1.json 2>/dev/null
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --timed-only > $O/bench_form_n1.json 2>/dev/null
python - <<PY
import json
a=json.loads(open("$O/scale_form_n1.json").read().strip().splitlines()[-1]); b=json.loads(open("$O/bench_form_n1.json").read().strip().splitlines()[-1])
print("N=1 through torch.distributed.run: %.3f ms/step; plain bench.py: %.3f ms/step" % (a["ms_per_step"], b["ms_per_step"]))
PY
