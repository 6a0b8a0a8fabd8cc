#!/usr/bin/env python3
"""Condenses the rocprofv3 output of profiles/collect.sh into the two files committed under profiles/:
<tag>_kernel_stats.csv (verbatim copy of the --stats summary) and <tag>_hbm_traffic.json (per-kernel
average FETCH_SIZE / WRITE_SIZE per dispatch, plus the render kernel's HBM bytes per launch)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


ks = one(f"{tag}_stats/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
b = os.path.join(out, f"{tag}_bench_under_rocprof.json")
if os.path.exists(b):
    lines = [l for l in open(b) if l.startswith("{")]
    if lines:
        open(os.path.join(root, "profiles", f"{tag}_bench_under_rocprof.json"), "w").write(lines[-1])

per = defaultdict(lambda: defaultdict(list))
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = one(f"{tag}_{sub}/**/*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(float)   # (dispatch, kernel) -> summed over XCD rows
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        acc[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
    for (_, k), v in acc.items():
        per[k[:60]][ctr].append(v)
kern = {}
for k, d in per.items():
    kern[k] = {f"{c}_KB_avg": sum(v) / len(v) for c, v in d.items()}
    kern[k]["dispatches"] = max(len(v) for v in d.values())
render = [k for k in kern if "gdpt_render_persistent" in k or "gdpt_render_phases" in k]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around "
               "`python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline` on MI355X (profiles/collect.sh); units KB per "
               "dispatch as reported; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests "
               "at 64 B, MI355X_MICROARCH.md HBM section; the factor is calibrated for 16-B/lane streams, so the read side "
               "is an upper-bound estimate)",
       "kernels": kern}
if render:
    # the dominant one = the render kernel with the most traffic
    rk = max(render, key=lambda k: kern[k].get("WRITE_SIZE_KB_avg", 0) + kern[k].get("FETCH_SIZE_KB_avg", 0))
    res["render_kernel"] = rk
    res["render_traffic_bytes_per_launch"] = (2 * kern[rk].get("FETCH_SIZE_KB_avg", 0) + kern[rk].get("WRITE_SIZE_KB_avg", 0)) * 1024
json.dump(res, open(os.path.join(root, "profiles", f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "note"}, indent=1)[:3000])
