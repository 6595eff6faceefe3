V=${1:-v7}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/$V
python bench.py > gpurun_out/$V/bench.json 2> gpurun_out/$V/bench.err
tail -c 600 gpurun_out/$V/bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$V/stats -o $V -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/$V/stats.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$V/pmc_$c.log 2>&1); done
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmc_SQ_INSTS_VALU -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$V/pmc_valu.log 2>&1)
find gpurun_out/$V -name "*.csv" | head -20
