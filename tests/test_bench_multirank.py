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
    _check_rank_evidence(d, 2)


def _check_rank_evidence(d, n):
    """what makes an N-rank line self-evidencing: the ranks that met, the card each held, each rank's own rate,
    and both north-star shapes (1241x376 and 640x480) as barrier-aligned whole-job legs"""
    c = d["config"]
    assert c["ranks_seen"] == n and len(c["rank_devices"]) == n and len(c["per_rank_frames_per_s"]) == n
    assert all(len(x.split()) >= 3 for x in c["rank_devices"])            # "<pci bus id> <uuid> <arch>"
    assert all(x.split()[-1].startswith("gfx950") for x in c["rank_devices"])
    assert min(c["per_rank_frames_per_s"]) * n >= d["value"] * 0.999          # value = n * B * K / (slowest rank's time + the closing barrier)
    t = c["tum640_frames_per_s"]
    assert t["value"] > 0 and len(t["per_rank_frames_per_s"]) == n and 900 < t["keypoints_per_image"] < 1100
    assert t["verified"]["bit_exact"] is True
    assert c["kitti2000_frames_per_s"]["value"] > 0


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher on the command line and WORLD_SIZE unset: the parent starts the two ranks as child
    processes (before any GPU call of its own), relays rank 0's one JSON line and the exit code"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ORBX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8", "--distinct", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["config"]["backend"] == "gloo"
    assert d["verified"]["bit_exact"] is True
    _check_rank_evidence(d, 2)


def test_parent_never_touches_the_gpu(tmp_path):
    """CPU: with a stand-in `torch.distributed.run` first on PYTHONPATH (it shadows torch altogether: an `import torch` + any
    torch.cuda call in the parent would raise), `bench.py --gpus 3` must spawn it with the same arguments, relay its one JSON line
    and its exit code, and exit -- without importing torch or loading liborbx.so itself"""
    pk = tmp_path / "torch" / "distributed"
    pk.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pk / "__init__.py").write_text("")
    (pk / "run.py").write_text(
        "import json, os, sys\n"
        "a = sys.argv[1:]\n"
        "assert '--nproc-per-node' in a and a[a.index('--nproc-per-node') + 1] == '3', a\n"
        "assert a[a.index('--master-addr') + 1] == '127.0.0.1'\n"
        "i = [k for k, x in enumerate(a) if x.endswith('bench.py')][0]\n"
        "print('rank chatter')\n"
        "print(json.dumps({'launcher_args': a[:i], 'bench_args': a[i + 1:], 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}))\n"
        "sys.exit(int(os.environ.get('FAKE_RC', '0')))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path) + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--batch", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "rank chatter" in out.stdout
    d = json.loads(lines[0])
    assert d["bench_args"] == ["--gpus", "3", "--steps", "2", "--warmup", "1", "--batch", "4"] and d["ipc"] == "0"
    # the launcher's failure is the parent's failure
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(env, FAKE_RC="7"), cwd=str(tmp_path))
    assert out.returncode == 7
    # under a launcher (WORLD_SIZE set) the same command does NOT spawn: it goes on to import torch (the empty stand-in here) and fails there
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="3", RANK="0"), cwd=str(tmp_path))
    assert out.returncode != 0 and "launcher_args" not in out.stdout


@pytest.mark.gpu
def test_nccl_branch_runs_with_one_rank():
    """the "nccl" (= RCCL) branch of an N-GPU run on the one GPU of the test box: ORBX_BENCH_FORCE_DIST=1 makes a ONE-rank process group, so
    init_process_group("nccl", device_id=...), the barriers, the MAX / SUM all-reduces and the all-gathers on device tensors all execute through
    RCCL (two ranks cannot share a card under nccl; the 2-rank runs above use gloo)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ORBX_BENCH_BACKEND")}
    env.update(ORBX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8", "--distinct", "4", "--extras", "0", "--cpu-frames", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["backend"] == "nccl" and c["ranks_seen"] == 1 and len(c["rank_devices"]) == 1 and len(c["per_rank_frames_per_s"]) == 1
    assert d["verified"]["bit_exact"] is True
