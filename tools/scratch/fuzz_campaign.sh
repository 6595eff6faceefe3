# The three fuzz tools over fresh seeds in one GPU call; summaries are appended under gpurun_out/fz/ (copy them to profiles/).
# usage (GPU box): bash tools/scratch/fuzz_campaign.sh [first seed of fuzz_gpu, default 9901] [seeds, default 8]
S0=${1:-9901}; N=${2:-8}
mkdir -p gpurun_out/fz
for ((s=S0; s<S0+N; s++)); do timeout 400 python tools/fuzz_gpu.py 500 $s gpurun_out/fz/fuzz.txt > gpurun_out/fz/f$s.log 2>&1; tail -1 gpurun_out/fz/f$s.log; done
for ((s=S0+10; s<S0+13; s++)); do timeout 300 python tools/fuzz_final.py 120 $s gpurun_out/fz/fuzz_final.txt > gpurun_out/fz/ff$s.log 2>&1; tail -1 gpurun_out/fz/ff$s.log; done
for ((s=S0%100+30; s<S0%100+34; s++)); do timeout 300 python tools/fuzz_stream.py 50 $s gpurun_out/fz/fuzz_stream.txt > gpurun_out/fz/fs$s.log 2>&1; tail -1 gpurun_out/fz/fs$s.log; done
