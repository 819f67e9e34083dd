"""Reduces the FETCH_SIZE / WRITE_SIZE passes of tools/config_traffic.sh to bytes per launch of one kernel.
usage: config_traffic.py <dir> <prefix> <kernel substring> <points key of the bench json>"""
import csv
import glob
import json
import os
import sys

root, prefix, kernel, key = sys.argv[1:5]


def per_launch(counter):
    per = {}
    for path in glob.glob(os.path.join(root, f"{prefix}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r.get("Kernel_Name", "") and r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    xs = sorted(per.values())
    xs = xs[len(xs) // 4:] if len(xs) >= 4 else xs          # drop the small launches of the warm-up / first frame
    return (sum(xs) / len(xs) * 1024.0, len(xs)) if xs else (None, 0)     # counters are in KB


fetch, n_f = per_launch("FETCH_SIZE")
write, n_w = per_launch("WRITE_SIZE")
bench = json.loads(open(os.path.join(root, f"{prefix}_FETCH_SIZE.json")).read().strip().splitlines()[-1])
print(json.dumps({
    "kernel": kernel, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "points_per_launch": bench[key],
    "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/config_traffic.sh), mean of "
              f"{n_f} / {n_w} dispatches",
    "correction": "factor 1.0, as for the fp32 field kernel (profiles/r2/field_traffic.json): isolated 64-B sector reads",
}, indent=1))
