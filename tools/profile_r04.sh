#!/bin/bash
# Round-4 profile pass (GPU box, through gpurun): bench lines, rocprofv3 kernel stats in both stream modes, VALU and
# HBM PMC passes for the C2 kernels, the secondary workloads.  usage: bash tools/profile_r04.sh <tag>  -> gpurun_out/<tag>/...
V=${1:-r04p}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" >> $O/bench_c2.err
tail -c 300 $O/bench_c2.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline --timed-only > $O/stats_c2.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2_mode0 -o c2 -- python3 $R/bench.py --no-cpu-baseline --timed-only --pipe-mode 0 > $O/stats_c2_mode0.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/pmc_valu_c2.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/pmc_$c.log 2>&1); done
timeout 300 python bench.py --host-threads 2 --steps 20 --no-cpu-baseline > $O/bench_c2_host2.json 2> $O/bench_c2_host2.err
timeout 600 python bench.py --monomers 64 --reads 256 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python bench.py --config c4-second-best --steps 5 --warmup 2 > $O/bench_c4_second_best.json 2> $O/bench_c4sb.err
timeout 600 python bench.py --scaling strong --config c5 --seq-len 200000000 --steps 3 --warmup 1 > $O/bench_c5_200mb.json 2> $O/bench_c5.err
timeout 600 python bench.py --scaling strong --config c5 --seq-len 25000000 --steps 5 --warmup 1 > $O/bench_c5_25mb.json 2> $O/bench_c5b.err
timeout 600 python bench.py --scaling strong --config c3 --reads-total 2000 --steps 3 --warmup 1 > $O/bench_c3_2000.json 2> $O/bench_c3.err
find $O -name "*.csv" | wc -l
