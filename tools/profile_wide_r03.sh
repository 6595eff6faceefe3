#!/bin/bash
# Round 3, after the one-address LDS reads of the wide / multi-wave fills: bench lines of the 64-, 128- and
# 400-monomer shapes and the SQ counters of the fills (tools/pmc_stall.sh).  usage: bash tools/profile_wide_r03.sh <tag>
V=${1:-wide}
O=gpurun_out/$V
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --monomers 64 --reads 256 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 300 python bench.py --no-cpu-baseline --monomers 128 --reads 64 --steps 5 --warmup 2 > $O/bench_m128.json 2> $O/bench_m128.err
timeout 300 python bench.py --no-cpu-baseline --monomers 128 --reads 256 --steps 5 --warmup 2 > $O/bench_m128_256reads.json 2> $O/bench_m128_256.err
timeout 300 python bench.py --no-cpu-baseline --monomers 400 --reads 64 --steps 5 --warmup 2 > $O/bench_m400.json 2> $O/bench_m400.err
python3 - <<PY
import json
for w in ("c4", "m128", "m128_256reads", "m400"):
    d = json.loads(open("$O/bench_%s.json" % w).read().strip().split("\n")[-1]); r = d["roofline"]
    print("%-14s %.3f Gbp/s %.2f ms/step; fill alone %.2f ms = %.2f Tcell/s" % (w, d["value"] / 1e9, d["ms_per_step"],
          r["isolated_avg_launch_ms"], d["config"]["rows_per_gpu"] * d["config"]["sum_template_len"] / r["isolated_avg_launch_ms"] / 1e9))
PY
bash tools/pmc_stall.sh $V/stall > $O/pmc_stall.txt 2>&1
tail -40 $O/pmc_stall.txt
