#!/usr/bin/env python
"""Condense rocprofv3 --pmc counter_collection.csv files into per-kernel per-launch means (timed launches only).

usage: summarize.py KERNEL LAST_N out.json dir1 [dir2 ...] [--meta key=value ...]
       (each dir holds bench_counter_collection.csv; --meta pairs are stored at the top level, ints where they parse)"""
import csv, json, os, sys
args = sys.argv[1:]
meta = {}
if "--meta" in args:
    i = args.index("--meta")
    for kv in args[i + 1:]:
        k, v = kv.split("=", 1)
        meta[k] = int(v) if v.lstrip("-").isdigit() else v
    args = args[:i]
kernel, last_n, out = args[0], int(args[1]), args[2]
res = {"kernel": kernel, "launches_averaged": last_n, **meta, "counters": {}}
for d in args[3:]:
    path = d + "/bench_counter_collection.csv"
    if not os.path.exists(path):
        continue
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"] == kernel:
            acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in acc.items():
        v = v[-last_n:]
        res["counters"][k] = {"mean_per_launch": sum(x for x, _ in v) / len(v), "mean_duration_us": sum(t for _, t in v) / len(v) / 1e3, "launches": len(v)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
