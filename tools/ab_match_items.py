"""Single-pair matcher calls with their work items by value (k_match_v, default) against items read from mapped host memory (k_match):
python tools/ab_match_items.py  ->  median us per call of tools/matcher_bench.measure for both, interleaved twice"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from tools import matcher_bench as mb
pkg = ge.load_pkg()
out = {}
for rnd in range(2):
    for name, in_mem in (("items_in_memory", True), ("items_by_value", False)):
        pkg.orbx.debug_set_match_items(in_mem)
        r = mb.measure(pkg, None, reps=300, cpu_reps=0)
        out.setdefault(name, []).append({"host_pointers": r["gpu"], "resident": r.get("gpu_resident")})
pkg.orbx.debug_set_match_items(False)
print(json.dumps(out, indent=1))
