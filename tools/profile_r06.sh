# Round-6 profile pass (run on the GPU box through gpurun): the driver's bench form, rocprofv3 kernel stats in both stream
# modes, VALU / HBM counters of the C2 kernels AND of the C4 wide fill (separate --pmc passes), optionally the rest
# (strong-scaling lines of config 5, C4 --second-best with its kernel stats, two host threads per rank).
# usage: bash tools/profile_r06.sh <tag> [pmc|all]      -> gpurun_out/<tag>/...
V=${1:-r06}
WHAT=${2:-all}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/pmc_valu_c2.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/pmc_$c.log 2>&1); done
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c4 -o p -- python3 $R/bench.py --monomers 64 --reads 256 --steps 3 --warmup 1 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/pmc_valu_c4.log 2>&1)
find $O -name "*.csv" | wc -l
[ "$WHAT" = pmc ] && exit 0
python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 200 $O/bench_c2.json; echo
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --timed-only --host-threads 2 > $O/bench_c2_host_threads_2.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --timed-only > $O/stats_c2.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2_mode0 -o c2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --timed-only --pipe-mode 0 > $O/stats_c2_mode0.log 2>&1)
python bench.py --monomers 64 --reads 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c4_dp.json 2>/dev/null
python bench.py --config c5 --scaling strong --seq-len 25000000 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5_25mb.json 2>/dev/null
python bench.py --config c5 --scaling strong --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c5_200mb.json 2>/dev/null
python bench.py --config c4-second-best --steps 6 --warmup 2 > $O/bench_c4_second_best.json 2> $O/bench_c4_second_best.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4sb -o c4 -- python3 $R/bench.py --config c4-second-best --steps 4 --warmup 1 --no-cpu-baseline > $O/stats_c4sb.log 2>&1)
find $O -name "*.csv" | wc -l
SD_TIMING=1 python bench.py --config c4-second-best --steps 2 --warmup 3 --no-cpu-baseline 2>&1 | grep "sd timing" | grep -v "FASTA\|chunk table\|pipeline from\|batch plan\|identities on\|of which" | tail -14 > $O/c4_second_best_timeline.txt
for s in 9601 9602 9603; do python tools/fuzz_gpu.py 500 $s $O/fuzz.txt > /dev/null 2>&1; done
python tools/fuzz_final.py 80 9611 $O/fuzz_final.txt > /dev/null 2>&1
for s in 21 22; do python tools/fuzz_stream.py 50 $s $O/fuzz_stream.txt > /dev/null 2>&1; done
cat $O/fuzz*.txt
