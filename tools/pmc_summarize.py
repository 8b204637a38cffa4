#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_profile.sh) per kernel: mean counter value per dispatch."""
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        short = "k_render_ctr" if "k_render_ctr" in k else "k_resolve" if "k_resolve" in k else "k_render_ref" if "k_render_ref" in k else None
        if short is None:
            continue
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    d["_dispatches"] = max(len(v) for v in acc[k].values())
print(json.dumps(out, indent=1, sort_keys=True))
