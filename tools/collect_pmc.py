#!/usr/bin/env python3
"""Collect HBM-side traffic of every kernel with rocprofv3 PMC passes and write profiles/<tag>_traffic.json.

Run ON THE GPU BOX (e.g. `gpurun -- python3 tools/collect_pmc.py r01`).  FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes (they do not fit one pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots),
each with --kernel-trace only.  rocprofv3 is given `python3 bench.py ...` directly after `--` (no shell,
env or launcher hop).  Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch.
gfx950 correction: FETCH_SIZE tallies the L2's 128-byte memory-side read requests at 64 bytes.  Calibrated on known byte
counts in the access patterns these kernels use (tools/ubench/fetch_calib.hip, profiles/r03_fetch_calibration.json: a 512 MiB
stream read with global_load_dword, global_load_dwordx4, global_load_lds_dword and global_load_lds_dwordx4 reports 0.500 x the
bytes in every case; WRITE_SIZE reports 1.000 x for dword and dwordx4 stores), so FETCH_SIZE is DOUBLED here; the raw counters are
stored beside the corrected bytes.  (Round 2 took the raw value for dword loads: its "k_resize 156 MB = algorithmic" was half the truth.)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_sha: which kernel sources these counters belong to)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
workload = sys.argv[3] if len(sys.argv) > 3 else "stereo1000"   # euroc_bow: BASELINE config 3 (k_bow<0>, k_vocab_*, k_bow_build)
out_dir = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{workload}")
res = collections.defaultdict(dict)
FETCH_FACTOR = 2.0     # profiles/r03_fetch_calibration.json: FETCH_SIZE = 0.500 x the bytes for every streaming read pattern in use
env = dict(os.environ, TMPDIR="/tmp")
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out_dir, counter)
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                           "python3", os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--cpu-frames", "0", "--extras", "0", "--no-verify",
                           "--batch", str(batch), "--workload", workload], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()   # "void k_fast<48, 40>(...)" -> k_fast
        acc[name].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k.startswith("k_"):
            res[k][counter + "_KiB_per_launch"] = sum(v) / len(v)
            res[k]["launches_sampled"] = len(v)
for k in res:
    res[k]["fetch_bytes_per_launch_corrected"] = int(1024 * FETCH_FACTOR * res[k].get("FETCH_SIZE_KiB_per_launch", 0))
    res[k]["write_bytes_per_launch"] = int(1024 * res[k].get("WRITE_SIZE_KiB_per_launch", 0))
    res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch_corrected"] + res[k]["write_bytes_per_launch"]
doc = {"tag": tag, "workload": workload, "csrc_sha": bench.csrc_sha(), "frames_per_step": batch, "images_per_launch": 2 * batch if workload.startswith("stereo") else batch,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace; mean over dispatches; "
                 "FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests at 64 bytes; calibrated on known byte counts for dword, dwordx4, "
                 "lds_dword and lds_dwordx4 loads, profiles/r03_fetch_calibration.json), WRITE_SIZE as reported",
       "kernels": res}
path = os.path.join(ROOT, "gpurun_out", f"{tag}_traffic.json" if workload == "stereo1000" else f"{tag}_traffic_{workload}.json")
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in res.items()}))
