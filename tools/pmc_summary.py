#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void tnf::", "").replace("tnf::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(out, "p*", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void tnf::", "").replace("tnf::", "")
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
res = {}
for name, ctr in acc.items():
    if not any(t in name for t in ("flow_fused", "flow_bwd", "coupling_mfma", "coupling_bwd", "cond_flow", "cond_gw", "maf_kernel", "to_interval")):
        continue
    # steady-state dispatches only: drop the first (cold) one of each counter
    res[name] = {k: sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0] for k, v in ctr.items()}
    d = dur.get(name, [])
    if d:
        res[name]["avg_us_under_pmc"] = sum(d) / len(d)
        res[name]["dispatches"] = len(d)
print(json.dumps(res, indent=1, sort_keys=True))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
