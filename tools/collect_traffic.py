#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/fill_traffic.json.

usage: collect_traffic.py <dir with pmc_FETCH_SIZE/, pmc_WRITE_SIZE/ [, pmc_SQ_INSTS_VALU/] rocprof outputs> <bench json>
(the optional third pass, --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE, adds the VALU issue-slot figures)
HBM bytes per launch of the fill kernel, following MI355X_MICROARCH.md (HBM section):
  * counters are collected in separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2);
  * both are in KiB;
  * on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads, so
    it is doubled (the fill reads almost nothing from HBM, so this correction is immaterial here);
  * WRITE_SIZE is exact for 16-B-per-lane streaming stores; the fill's checkpoint stores are
    4 B per lane (256 B per wave-instruction), a width the guide lists as uncalibrated.
"""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel_sub):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s under %s" % (counter, kernel_sub, d))
    vals.sort()
    return vals[len(vals) // 2], len(vals)


def main():
    root, bench_json = sys.argv[1], sys.argv[2]
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    kern = bench["roofline"]["kernel"]
    fetch, nf = per_launch(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE", kern)
    write, nw = per_launch(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE", kern)
    extra = {}
    if os.path.isdir(os.path.join(root, "pmc_SQ_INSTS_VALU")):
        v, _ = per_launch(os.path.join(root, "pmc_SQ_INSTS_VALU"), "SQ_INSTS_VALU", kern)
        g, _ = per_launch(os.path.join(root, "pmc_SQ_INSTS_VALU"), "GRBM_GUI_ACTIVE", kern)
        extra = {"SQ_INSTS_VALU_per_launch": v, "GRBM_GUI_ACTIVE_per_launch": g,
                 "cells": bench["config"].get("cell_arithmetic", "").split(" ")[0]}
    out = {
        "kernel": kern,
        "kernel_family": bench["config"]["kernel_family"],
        "workload_rows": bench["config"]["rows_per_gpu"],
        "FETCH_SIZE_KiB_median": fetch, "WRITE_SIZE_KiB_median": write, "launches_seen": [nf, nw],
        "fetch_bytes_corrected_x2": 2 * fetch * 1024,
        "write_bytes": write * 1024,
        "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace; "
                  "KiB -> bytes; FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)",
    }
    out.update(extra)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["kernel_sources_sha256"] = bench.kernel_source_hashes()   # bench.py withholds the counters when the kernel sources change
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "fill_traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
