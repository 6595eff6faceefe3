# Round-2 profile pass (run on the GPU box through gpurun): bench lines + rocprofv3 kernel stats for C2
# (narrow fill) and the C4 shape (wide fill, 64 monomers x 256 reads), VALU PMC for every kernel, HBM
# traffic PMC for C2, the --ed_thr filter kernels, the NW identity kernel, CLI and host-stage timings.
# usage: bash tools/profile_r02.sh <tag>      -> gpurun_out/<tag>/...
V=${1:-r02}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 300 $O/bench_c2.json
python bench.py --monomers 64 --reads 256 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --ed-thr 20 --steps 5 --no-cpu-baseline > $O/bench_c2_edthr20.json 2> $O/bench_c2_edthr20.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline --timed-only > $O/stats_c2.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o c4 -- python3 $R/bench.py --monomers 64 --reads 256 --steps 5 --no-cpu-baseline --timed-only > $O/stats_c4.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_edthr -o e -- python3 $R/bench.py --ed-thr 20 --steps 5 --no-cpu-baseline --timed-only > $O/stats_edthr.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nw -o n -- python3 $R/tools/nw_bench.py 64000 12 > $O/stats_nw.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only > $O/pmc_valu_c2.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c4 -o p -- python3 $R/bench.py --monomers 64 --reads 256 --steps 3 --warmup 1 --no-cpu-baseline --timed-only > $O/pmc_valu_c4.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_edthr -o p -- python3 $R/bench.py --ed-thr 20 --steps 3 --warmup 1 --no-cpu-baseline --timed-only > $O/pmc_valu_edthr.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only > $O/pmc_$c.log 2>&1); done
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_nw_$c -o p -- python3 $R/tools/nw_bench.py 64000 12 > $O/pmc_nw_$c.log 2>&1); done
python tools/cli_bench.py 1000 32 12 1 80 > $O/cli_50mbp_light.txt 2>&1
python tools/cli_bench.py 1000 32 12 1 80 --second-best > $O/cli_50mbp_second_best.txt 2>&1
python tools/cli_bench.py 1000 64 12 10 0 > $O/cli_500mbp_light.txt 2>&1
python tools/cli_bench.py 256 32 64 1 80 --second-best > $O/cli_c4_second_best.txt 2>&1
python -m pytest tests/test_host_cpu.py -q -k eight -s > $O/host_stage_rates.txt 2>&1
python tools/nw_bench.py 64000 12 > $O/nw_bench.txt 2>&1
find $O -name "*.csv" | wc -l
