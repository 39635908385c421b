#!/usr/bin/env python3
"""Instruction-mix / stall counters of every kernel (rocprofv3 --pmc, a few SQ counters per pass).
Run ON THE GPU BOX: `gpurun -- python3 tools/collect_sq.py r01`.  Writes gpurun_out/<tag>_sq.json
(per-launch means, summed over XCDs/SEs as rocprofv3 reports them)."""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_sha: which kernel sources these counters belong to)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
PASSES = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES"],
    ["SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
]
res = collections.defaultdict(dict)
env = dict(os.environ, TMPDIR="/tmp")
for i, counters in enumerate(PASSES):
    d = os.path.join(ROOT, "gpurun_out", f"sq_{tag}", f"pass{i}")
    rc = subprocess.call(["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                          "python3", os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-frames", "0", "--extras", "0", "--no-verify"],
                         cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if rc != 0 or not files:
        print("pass", i, "failed rc", rc); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if k.startswith("k_"):
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
    print("pass", i, "done", flush=True)
json.dump({"csrc_sha": bench.csrc_sha(), "kernels": res}, open(os.path.join(ROOT, "gpurun_out", f"{tag}_sq.json"), "w"), indent=1)
for k, v in res.items():
    w = v.get("SQ_WAVES", 1)
    print(k, {c: round(x / w, 1) for c, x in v.items()})
