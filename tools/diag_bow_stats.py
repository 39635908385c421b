"""Work statistics of k_bow on the euroc_bow bench data (diagnostic build: python -c "import __graft_entry__ as g; g.build_diag()").
Run on the GPU box: python tools/diag_bow_stats.py"""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ORBX_SO"] = os.path.join(ROOT, "diag", "liborbx_diag.so")
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--workload", "euroc_bow", "--steps", "2", "--warmup", "0", "--cpu-frames", "0"]
import bench
import __graft_entry__ as ge
L = ge.load_pkg().lib()
bench.main()
out = (C.c_ulonglong * 8)()
L.orbx_diag_bow_stats(out, 1)
wg = max(out[4], 1)
print("phase-1 cycles/wg", out[6] / max(out[4], 1), "phase-2 cycles/wg", out[7] / max(out[4], 1))
print("workgroups", out[4], "entries/wg", out[0] / wg, "rows/wg", out[5] / wg, "rounds/wg", out[1] / wg, "passes/wg", out[2] / wg, "fallback nodes/wg", out[3] / wg)
