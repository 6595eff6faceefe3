# Round-2 profile pass (run on the GPU box through gpurun): bench lines + rocprofv3 kernel stats for C2
# (narrow fill) and the C4 shape (wide fill, 64 monomers x 256 reads), VALU PMC for every kernel.
# usage: bash tools/profile_r02.sh <tag>      -> gpurun_out/<tag>/...
V=${1:-r02}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 400 $O/bench_c2.json
python bench.py --monomers 64 --reads 256 --steps 5 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
tail -c 400 $O/bench_c4.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline > $O/stats_c2.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o c4 -- python3 $R/bench.py --monomers 64 --reads 256 --steps 5 --no-cpu-baseline > $O/stats_c4.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_valu_c2.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_c4 -o p -- python3 $R/bench.py --monomers 64 --reads 256 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_valu_c4.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_c2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_${c}_c2.log 2>&1); done
find $O -name "*.csv" | head -40
