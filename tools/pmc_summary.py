#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs of tools/pmc.sh: per kernel (template arguments dropped, so the launches of a
chain are averaged together), the mean counter value per dispatch, the mean duration under the profiler, and
`hbm_bytes_per_launch` = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- the gfx950 correction of MI355X_MICROARCH.md (HBM):
FETCH_SIZE tallies 64 B per 128-B request of a 16-B-per-lane streaming read, WRITE_SIZE is exact for 16-B stores.
The file is stamped with the hash of the kernel sources it was collected on (bench.kernel_source_hash): bench.py quotes
`roofline.traffic` from a summary only when that hash matches the tree it runs in.
usage: pmc_summary.py <dir with p1/, p2/, ...>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

KEEP = ("flow_fused", "flow_range", "flow_bwd", "coupling_mfma", "coupling_bwd", "coupling_wide", "wide_gw", "backward_reduce", "cond_flow", "cond_gw",
        "cond_gh", "maf_", "to_interval")


def base(name):
    name = name.replace("void tnf::", "").replace("tnf::", "")
    return re.split(r"[<(]", name)[0]


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    variants = defaultdict(set)
    for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[base(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            variants[base(row["Kernel_Name"])].add(row["Kernel_Name"].split("(")[0].replace("void tnf::", ""))
    for f in glob.glob(os.path.join(out, "p*", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[base(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    res = {"kernel_source_sha16": kernel_source_hash(),
           "correction": "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 FETCH_SIZE counts 64 B per "
                         "128-B request of a 16-B-per-lane streaming read (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact"}
    for name, ctr in acc.items():
        if not any(t in name for t in KEEP):
            continue
        # steady-state dispatches only: drop the first (cold) one of each counter
        r = {k: (sum(v[1:]) / (len(v) - 1) if len(v) > 1 else v[0]) for k, v in ctr.items()}
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            r["hbm_bytes_per_launch"] = int(round((2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024))
        d = dur.get(name, [])
        if d:
            r["avg_us_under_pmc"] = sum(d) / len(d)
            r["dispatches"] = len(d)
        r["instantiations"] = sorted(variants[name])
        res[name] = r
    print(json.dumps(res, indent=1, sort_keys=True))
    json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
