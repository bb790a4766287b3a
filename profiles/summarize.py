#!/usr/bin/env python
"""Condense rocprofv3 --pmc counter_collection.csv files into per-kernel per-launch means (timed launches only).

usage: summarize.py KERNEL LAST_N out.json dir1 [dir2 ...]   (each dir holds bench_counter_collection.csv)"""
import csv, json, sys
kernel, last_n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
res = {"kernel": kernel, "launches_averaged": last_n, "counters": {}}
for d in sys.argv[4:]:
    acc = {}
    for r in csv.DictReader(open(d + "/bench_counter_collection.csv")):
        if r["Kernel_Name"] == kernel:
            acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in acc.items():
        v = v[-last_n:]
        res["counters"][k] = {"mean_per_launch": sum(x for x, _ in v) / len(v), "mean_duration_us": sum(t for _, t in v) / len(v) / 1e3}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
