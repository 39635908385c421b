#!/usr/bin/env python3
"""Copies the measurement summaries of gpurun_out/final_<tag>/ (tools/final_meas.sh) into profiles/ (tracked) and merges the SQ
counters with the static instruction mix (tools/isa_mix.py) into profiles/<tag>_sq_counters.json, the file bench.py's
roofline.issue reads.  Runs in the build container after the gpurun call:  python tools/publish_profiles.py r02"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"final_{tag}")
dst = os.path.join(ROOT, "profiles")


def last_json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


for name, out in (("bench_default.json", f"{tag}_z_bench_default.json"), ("bench_under_rocprof.json", f"{tag}_z_bench_under_rocprof.json"),
                  ("stereo2000.json", f"{tag}_z_bench_stereo2000.json"), ("fhd4000.json", f"{tag}_z_bench_fhd4000.json"),
                  ("euroc_bow.json", f"{tag}_z_bench_euroc_bow.json")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        json.dump(last_json_line(p), open(os.path.join(dst, out), "w"), indent=1)
p2 = os.path.join(src, "bench_2ranks_gloo.json")
if os.path.exists(p2) and os.path.getsize(p2):
    json.dump(last_json_line(p2), open(os.path.join(dst, f"{tag}_z_bench_2ranks_gloo.json"), "w"), indent=1)
stats = glob.glob(os.path.join(src, "prof", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, f"{tag}_z_kernel_stats.csv"))   # the newest run
for name in (f"{tag}_traffic.json", f"{tag}_traffic_euroc_bow.json"):
    tr = os.path.join(ROOT, "gpurun_out", name)
    if os.path.exists(tr):
        shutil.copy(tr, os.path.join(dst, name))
others = glob.glob(os.path.join(src, "prof_others", "*", "*kernel_stats.csv"))
if others:
    shutil.copy(max(others, key=os.path.getmtime), os.path.join(dst, f"{tag}_other_kernels_stats.csv"))
for name, out in (("b1_chain.txt", f"{tag}_b1_launch_chain.txt"), ("matcher_kernels.txt", f"{tag}_matcher_kernels.txt"), ("matchers.json", f"{tag}_z_matchers.json"),
                  ("match_stamps.txt", f"{tag}_match_stamps.txt"), ("hostfed_c.txt", f"{tag}_hostfed_c.txt"), ("pcie_bw.txt", f"{tag}_pcie_and_hostfed_batched.txt"),
                  ("fast_pair_ab.txt", f"{tag}_fast_pair_ab.txt"), ("sq_bow.txt", f"{tag}_sq_bow.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, out))
sq = os.path.join(ROOT, "gpurun_out", f"{tag}_sq.json")
if os.path.exists(sq):
    mix = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py")]))
    kern = json.load(open(sq))
    sha = kern.get("csrc_sha") if "kernels" in kern else None
    kern = kern.get("kernels", kern)
    # launch durations of the plain kernel trace of the same build (no counters attached): the denominator of the counter-only
    # VALU-busy figure in bench.py
    if stats:
        import csv
        for r in csv.DictReader(open(os.path.join(dst, f"{tag}_z_kernel_stats.csv"))):
            k = r["Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
            if k in kern:
                kern[k]["launch_ms"] = float(r["AverageNs"]) / 1e6
    cyc = {}
    for k in kern:
        cands = [v for n, v in mix["kernels"].items() if n.split("<")[0] == k]
        if cands:
            cyc[k] = cands[0]["cycles_per_valu_inst"]
    doc = {"note": "rocprofv3 --pmc SQ counters, 4 passes (tools/collect_sq.py), per-launch means summed over XCDs/SEs as rocprofv3 reports them; "
                   "default bench workload (512 images per launch).  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count in units of 4 cycles. "
                   "issue_model: wave64 VALU issue cost per SIMD by opcode class measured with tools/ubench/op_cost.hip on the same GPU "
                   "(profiles/%s_op_cost.txt), weighted with each kernel's static instruction mix (tools/isa_mix.py)." % tag,
           "images_per_launch": 512, "csrc_sha": sha, "kernels": kern,
           "issue_model": {"simds": 1024, "clock_ghz": 2.4, "fast_class_cycles": mix["fast_cycles"], "slow_class_cycles": mix["slow_cycles"],
                           "cycles_per_valu_inst": cyc, "static_mix": mix["kernels"]}}
    json.dump(doc, open(os.path.join(dst, f"{tag}_sq_counters.json"), "w"), indent=1)
# (round 3 also copied gpurun_out/{op_cost.txt, issue_rate.txt, hostfed_c.log} -- scratch files of WHATEVER round last wrote them -- over the
# tagged names: that is how profiles/r03_hostfed_c.txt came to hold round 2's numbers.  Only files of gpurun_out/final_<tag>/ are published now.)
print("published:", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
