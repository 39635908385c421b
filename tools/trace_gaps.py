"""Kernel-to-kernel gaps of one bench step from a rocprofv3 --kernel-trace csv (newest under the given directory)."""
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
f = max(fs, key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "") in
        ("k_resize", "k_pyr_group", "k_fast", "k_tree", "k_desc", "k_stereo_prep", "k_stereo", "k_stereo_cut")]
seq = rows[-40:-14]
prev = None; tot_gap = 0; t0 = None
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = None if prev is None else (s - prev) / 1000
    if g is not None: tot_gap += g
    print(r["Kernel_Name"][:28].ljust(28), "%8.1f us" % ((e - s) / 1000), "gap", "-" if g is None else "%.1f" % g)
    prev = e
print("sum of gaps over", len(seq), "kernels: %.1f us" % tot_gap)
