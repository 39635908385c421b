"""rocprofv3 --kernel-trace csv -> median / min duration per (kernel, grid size): separates the single-pair launches of a
matcher kernel from its batched ones.  usage: python tools/trace_by_grid.py <dir-or-csv> [name-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if sub not in name:
            continue
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        acc[(name[:60], grid // max(wg, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for (name, wgs), v in sorted(acc.items()):
    v = np.array(v)
    print("%-60s workgroups %6d  calls %5d  median %8.1f us  min %8.1f  p90 %8.1f" % (name, wgs, len(v), np.median(v), v.min(), np.percentile(v, 90)))
