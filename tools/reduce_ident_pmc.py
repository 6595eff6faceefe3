#!/usr/bin/env python3
"""Reduce tools/ident_pmc.sh to profiles/r06_ident_pmc.json (developer tool, no GPU needed): per identity kernel the
wave-instructions and HBM bytes PER STEP of bench.py --config c4-second-best (FETCH_SIZE doubled: the gfx950 correction for
wide per-lane reads, MI355X_MICROARCH.md HBM section; KiB -> bytes).  usage: reduce_ident_pmc.py gpurun_out/<tag> out.json"""
import csv, glob, json, os, re, sys
root, dst = sys.argv[1], sys.argv[2]
steps, warm = map(int, open(os.path.join(root, "ident_pmc_steps.txt")).read().split())
jobs = steps + max(warm, 1)


def sums(d, ctr):
    acc = {}
    for f in glob.glob(os.path.join(root, d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr or "sd_ident" not in r["Kernel_Name"]:
                continue
            k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("sd::", "")
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    return acc


valu, act = sums("ident_pmc_valu", "SQ_INSTS_VALU"), sums("ident_pmc_valu", "GRBM_GUI_ACTIVE")
fe, wr = sums("ident_pmc_FETCH_SIZE", "FETCH_SIZE"), sums("ident_pmc_WRITE_SIZE", "WRITE_SIZE")
out = {"workload": "bench.py --config c4-second-best (64 monomers x 256 reads x 50 kb): identity kernels of ONE step = one sd_run_files job",
       "method": __doc__.split("\n")[1:4], "jobs_profiled": jobs, "kernels": {}}
for k in sorted(valu):
    out["kernels"][k] = {"SQ_INSTS_VALU_per_step": valu[k] / jobs, "GRBM_GUI_ACTIVE_per_step": act.get(k, 0.0) / jobs,
                         "hbm_bytes_per_step": (2 * 1024.0 * fe.get(k, 0.0) + 1024.0 * wr.get(k, 0.0)) / jobs}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out["kernels"], indent=1))
