# Counters of the identity kernels of a C4 --second-best step (round 6: the pruned homopolymer pass): SQ_INSTS_VALU /
# GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE in separate rocprofv3 --pmc passes over `bench.py --config c4-second-best`.
# usage (GPU box): bash tools/ident_pmc.sh <tag>   -> gpurun_out/<tag>/ident_pmc_*;  then tools/reduce_ident_pmc.py locally
V=${1:-r06}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
ST=3; WU=1
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/ident_pmc_valu -o p -- python3 $R/bench.py --config c4-second-best --steps $ST --warmup $WU --no-cpu-baseline > $O/ident_pmc_valu.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/ident_pmc_$c -o p -- python3 $R/bench.py --config c4-second-best --steps $ST --warmup $WU --no-cpu-baseline > $O/ident_pmc_$c.log 2>&1); done
echo "$ST $WU" > $O/ident_pmc_steps.txt
find $O -name "*counter_collection.csv" | grep ident_pmc | wc -l
