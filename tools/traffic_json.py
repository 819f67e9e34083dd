"""Reduces the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh to bytes per launch of the field kernel."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def per_launch(counter):
    per = {}
    for path in glob.glob(os.path.join(root, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "field_kernel" in r.get("Kernel_Name", "") and r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    xs = list(per.values())
    return (sum(xs) / len(xs) * 1024.0, len(xs)) if xs else (None, 0)     # counters are in KB


fetch, n_f = per_launch("FETCH_SIZE")
write, n_w = per_launch("WRITE_SIZE")
bench = json.load(open(os.path.join(root, "pmc_FETCH_SIZE.json")))
print(json.dumps({
    "kernel": "field_kernel<1, false>", "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
    "points_per_launch": bench["quadrature_points_per_frame"],
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 5 "
              f"--warmup 1 --no-cpu-baseline`, mean of {n_f} / {n_w} dispatches (tools/profile_bench.sh)",
    "correction": "factor 1.0: tools/calib_fetch.hip (16.78 M 8-byte loads from distinct 64-B sectors of a 2 GiB table) "
                  "reads FETCH_SIZE = 64 B per load = TCC_EA0_RDREQ x 64 B exactly; the L2 fills 64-B sectors on this "
                  "access pattern, so the guide's x2 correction for wide streaming reads does not apply to this gather",
}, indent=1))
