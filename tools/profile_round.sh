export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/v6
python bench.py > gpurun_out/v6/bench.json 2> gpurun_out/v6/bench.err
tail -c 600 gpurun_out/v6/bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v6/stats -o v6 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/v6/stats.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/v6/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/v6/pmc_$c.log 2>&1); done
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/v6/pmc_SQ_INSTS_VALU -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/v6/pmc_valu.log 2>&1)
find gpurun_out/v6 -name "*.csv" | head -20
