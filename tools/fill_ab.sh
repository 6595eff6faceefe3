#!/bin/bash
# A/B of the narrow fill's carry scan (ds_bpermute form against the DPP form, SD_FILL_DPP_SCAN=1) on C2: kernel times
# and SQ_INSTS_VALU of the fill.  usage: bash tools/fill_ab.sh <tag>  -> gpurun_out/<tag>/
V=${1:-fillab}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
{
echo "== bpermute scan"; timeout 120 python tools/kbench.py 1000 4
echo "== dpp scan"; SD_FILL_DPP_SCAN=1 timeout 120 python tools/kbench.py 1000 4
echo "== bpermute scan, trace v1"; SD_TRACE=1 timeout 120 python tools/kbench.py 1000 4
} > $O/kbench.txt 2>&1
for form in bp dpp; do
  if [ $form = dpp ]; then export SD_FILL_DPP_SCAN=1; else unset SD_FILL_DPP_SCAN; fi
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/${form}_pmc -o p -- python3 $R/tools/kbench.py 1000 2 > $O/${form}_pmc.log 2>&1)
done
unset SD_FILL_DPP_SCAN
python3 - <<PY
import csv, glob, collections
for form in ("bp", "dpp"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s_pmc/**/*counter_collection.csv" % form, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fill" in k or "trace" in k:
                acc[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(form, k)
        for c, v in sorted(d.items()):
            v.sort()
            print("   %-22s %.4g  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
