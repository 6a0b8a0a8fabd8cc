#!/usr/bin/env python3
"""Per-kernel averages of every counter collected by profiles/pmc.sh <tag> (sums over XCD rows per dispatch)."""
import csv, glob, json, os, sys
from collections import defaultdict
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(float)
    for r in csv.DictReader(open(f)):
        acc[(r["Dispatch_Id"], r["Kernel_Name"][:48], r["Counter_Name"])] += float(r["Counter_Value"])
    per = defaultdict(list)
    for (d, k, c), v in acc.items():
        per[(k, c)].append(v)
    for (k, c), v in per.items():
        res[k][c] = sum(v) / len(v)
        res[k]["dispatches"] = len(v)
keep = {k: v for k, v in res.items() if "gdpt" in k or "gp::" in k}
json.dump(keep, open(os.path.join(root, "gpurun_out", f"{tag}_pmc_summary.json"), "w"), indent=1)
for k, v in keep.items():
    if "render" in k:
        print(k); [print(f"   {c:32s} {x:16.1f}") for c, x in sorted(v.items())]
