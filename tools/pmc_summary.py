"""Aggregates rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")[:60]
        acc[name][r["Counter_Name"]].append((r.get("Dispatch_Id"), float(r["Counter_Value"])))
for name, ctrs in sorted(acc.items()):
    if not any(k in name for k in ("field_kernel", "bvh8_", "raster_kernel", "pack_samples", "derive_properties", "deform", "texture_shade",
                                   "grid_backward", "pack_tiles", "composite_tiles", "select_nearest")):
        continue
    print(name)
    for c, vals in sorted(ctrs.items()):
        per = defaultdict(float)
        for did, v in vals:
            per[did] += v
        xs = list(per.values())
        print(f"   {c:42s} mean/dispatch {sum(xs) / len(xs):.6g}   (dispatches {len(xs)})")
