#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/profile_r02.sh to one JSON (developer tool, no GPU needed).

usage: reduce_pmc.py <gpurun_out/<tag>> <out.json> [fill_traffic.json to rewrite]

Per pass directory (pmc_valu_<workload>, pmc_FETCH_SIZE, pmc_WRITE_SIZE, pmc_nw_<counter>) and kernel:
the median counter value over the launches seen.  VALU issue-slot fraction of a launch =
SQ_INSTS_VALU x 4.1 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): measured on this GPU (tools/ubench_issue.hip,
profiles/r03_ubench_issue.txt) a SIMD issues one packed-f16 / DPP / SDWA / 3-operand wave64 instruction per
4.07-4.20 cycles at any occupancy; only plain v_add_f32/u32, v_mov, v_max_f16, and/or/xor reach 2.2-2.3, and
not while packed ops are in the stream.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts half
of the bytes of wide streaming reads on gfx950 and is reported raw here (fill_traffic.json doubles it)."""
import csv
import glob
import json
import os
import re
import sys


def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    return k.replace("sd::", "")


def medians(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    out = {}
    for (k, c), v in acc.items():
        v.sort()
        out.setdefault(k, {})[c] = (v[len(v) // 2], len(v))
    return out


def main():
    root, dst = sys.argv[1], sys.argv[2]
    res = {"method": __doc__.split("\n\n")[2].replace("\n", " "), "valu": {}, "hbm_counters": {}}
    for d in sorted(glob.glob(os.path.join(root, "pmc_valu_*"))):
        wl = os.path.basename(d)[len("pmc_valu_"):]
        for k, c in medians(d).items():
            if "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c or not k.startswith("sd_"):
                continue
            insts, n = c["SQ_INSTS_VALU"]
            act = c["GRBM_GUI_ACTIVE"][0]
            cyc = act / 8.0
            res["valu"]["%s:%s" % (wl, k)] = {"SQ_INSTS_VALU_per_launch": insts, "GRBM_GUI_ACTIVE_per_launch": act,
                                              "cycles": cyc, "launches_seen": n,
                                              "valu_issue_frac": round(insts * 4.1 / (cyc * 1024.0), 4)}
    for tag, pat in (("c2", "pmc_%s"), ("nw", "pmc_nw_%s")):
        per = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(root, pat % ctr)
            if not os.path.isdir(d):
                continue
            for k, c in medians(d).items():
                if ctr in c and k.startswith("sd_"):
                    per.setdefault(k, {})[ctr + "_KiB_median"] = c[ctr][0]
        for k, v in per.items():
            res["hbm_counters"]["%s:%s" % (tag, k)] = v
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(res["valu"]), "valu entries,", len(res["hbm_counters"]), "hbm entries")
    if len(sys.argv) > 3:
        # the committed figures bench.py quotes for the dominant kernel (C2 fill)
        ft = json.load(open(sys.argv[3]))
        fill = [k for k in res["valu"] if k.startswith("c2:sd_fast_fill<")]
        hb = [k for k in res["hbm_counters"] if k.startswith("c2:sd_fast_fill<")]
        if fill and hb:
            v, h = res["valu"][fill[0]], res["hbm_counters"][hb[0]]
            ft["SQ_INSTS_VALU_per_launch"] = v["SQ_INSTS_VALU_per_launch"]
            ft["GRBM_GUI_ACTIVE_per_launch"] = v["GRBM_GUI_ACTIVE_per_launch"]
            ft["FETCH_SIZE_KiB_median"] = h["FETCH_SIZE_KiB_median"]
            ft["WRITE_SIZE_KiB_median"] = h["WRITE_SIZE_KiB_median"]
            ft["fetch_bytes_corrected_x2"] = 2 * 1024.0 * h["FETCH_SIZE_KiB_median"]
            ft["write_bytes"] = 1024.0 * h["WRITE_SIZE_KiB_median"]
            ft["hbm_bytes_per_launch"] = ft["fetch_bytes_corrected_x2"] + ft["write_bytes"]
            ft["kernel_instance"] = fill[0].split(":", 1)[1]
            # the traceback of the same workload (the other kernel of a C2 step): for the step's combined issue fraction
            tr = [k for k in res["valu"] if k.startswith("c2:sd_fast_trace")]
            if tr:
                ft["traceback_SQ_INSTS_VALU_per_launch"] = res["valu"][tr[0]]["SQ_INSTS_VALU_per_launch"]
                ft["traceback_kernel_instance"] = tr[0].split(":", 1)[1]
            # the build these counters belong to: bench.py withholds them when the kernel sources have changed since
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            ft["kernel_sources_sha256"] = bench.kernel_source_hashes()
            json.dump(ft, open(sys.argv[3], "w"), indent=1, sort_keys=True)
            print("updated", sys.argv[3])


if __name__ == "__main__":
    main()
