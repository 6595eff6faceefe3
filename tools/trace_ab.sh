#!/bin/bash
# A/B of the two traceback forms on C2 (GPU box, through gpurun): kernel times under the developer knobs of
# sd_fast_trace_pk (SD_TRACE_MARGIN: when block B is skipped; SD_TRACE_BPC: workgroups per CU) and SQ counters of
# both forms.  usage: bash tools/trace_ab.sh <tag>  -> gpurun_out/<tag>/
V=${1:-trab}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
{
echo "== default"; timeout 120 python tools/kbench.py 1000 3
echo "== v1"; SD_TRACE=1 timeout 120 python tools/kbench.py 1000 3
for m in -1000 0 3 6 10 1000; do echo "== margin $m"; SD_TRACE_MARGIN=$m timeout 120 python tools/kbench.py 1000 3; done
for b in 2 3 4 5; do echo "== bpc $b"; SD_TRACE_BPC=$b timeout 120 python tools/kbench.py 1000 3; done
} > $O/kbench.txt 2>&1
for form in pk v1; do
  if [ $form = v1 ]; then export SD_TRACE=1; else unset SD_TRACE; fi
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
    n=$(echo $set | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${form}_$n -o p -- python3 $R/tools/kbench.py 1000 2 > $O/${form}_$n.log 2>&1)
  done
done
unset SD_TRACE
python3 - <<PY
import csv, glob, collections
for form in ("pk", "v1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s_*/**/*counter_collection.csv" % form, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "trace" in k:
                acc[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(form, k)
        for c, v in sorted(d.items()):
            v.sort()
            print("   %-22s %.4g  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
