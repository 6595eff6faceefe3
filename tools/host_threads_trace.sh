#!/bin/bash
# Kernel timeline of the C2 step with few host threads: where does the device wait?  usage: tools/host_threads_trace.sh <outdir> [threads ...]
out=${1:-gpurun_out/ht}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in "$@"; do
  SD_BENCH_STEP_TIMES=1 python3 bench.py --host-threads $t --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$t -- python3 bench.py --host-threads $t --steps 12 --warmup 3 --no-cpu-baseline > $out/prof_$t.json 2> $out/prof_$t.err
  f=$(find $out/trace_$t -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $out/gaps_$t.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows)
fills = [e for e in ev if "sd_fast_fill" in e[2]]
tr = [e for e in ev if "sd_fast_trace" in e[2]]
print("fills", len(fills), "traces", len(tr))
for a, b in zip(fills[-14:], fills[-13:]):
    print("fill start->next start %.3f ms, fill dur %.3f, gap end->next start %.3f" % ((b[0]-a[0])/1e6, (a[1]-a[0])/1e6, (b[0]-a[1])/1e6))
# busy union over the last 10 steps
lo = fills[-11][0]; hi = fills[-1][0]
segs = sorted((max(s, lo), min(e, hi)) for s, e, _ in ev if e > lo and s < hi)
busy, cur_s, cur_e = 0, None, None
for s, e in segs:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.3f ms per step, device busy %.3f ms per step (any kernel)" % ((hi-lo)/1e7, busy/1e7))
PY
  rm -rf $out/trace_$t
  echo "threads=$t $(python3 -c "import json; j=json.loads(open('$out/bench_$t.json').read().splitlines()[-1]); print(j['ms_per_step'], j['host_ms_per_step'])")"
  cat $out/gaps_$t.txt | tail -16
done
