#!/bin/bash
# Kernel time + HBM traffic of the identity kernel (tools/nw_bench.py workload): rocprofv3 --kernel-trace --stats,
# then FETCH_SIZE / WRITE_SIZE in their own passes.  usage (on the GPU box): tools/prof_nw.sh <outdir> [blocks] [monomers]
out=${1:-gpurun_out/nwprof}; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o nw -- python3 tools/nw_bench.py "$@" > "$out/trace.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv" && cut -c1-160 "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -o nw -- python3 tools/nw_bench.py "$@" > "$out/pmc_$c.log" 2>&1
  f=$(find "$out/pmc_$c" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/pmc_$c.csv" && grep -c sd_nw "$out/pmc_$c.csv"
done
