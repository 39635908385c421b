"""The multi-rank path of bench.py on real hardware: two ranks (one process each, torch.distributed.run) on the ONE GPU of the
test box, `gloo` as the rendezvous backend (ORBX_BENCH_BACKEND=gloo: the driver's 8-GPU run uses "nccl" = RCCL; the data path has
no collective either way).  Checks the contract fields of the JSON line: whole-job aggregate over both ranks,
exactly K timed steps, weak scaling, the oracle verification of rank 0's last step and the KITTI-2000 leg (BASELINE config 5)
that every N-GPU line carries."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, ORBX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8", "--distinct", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["config"]["frames_per_step_per_gpu"] == 8
    # whole-job aggregate: 2 ranks x 8 frames x 4 steps over the slowest rank's time
    assert abs(d["value"] - 2 * 8 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-3
    assert d["verified"]["bit_exact"] is True and d["verified"]["frames"] == 8
    k2 = d["config"]["kitti2000_frames_per_s"]
    assert k2["value"] > 0 and k2["keypoints_per_image"] > 1900
    assert d["cpu_baseline"] is None and "host_fed" not in d["config"]      # rank-0, N = 1 legs only
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
