#!/usr/bin/env python3
"""bench.py — frames/s of ORB extract + match on MI355X (BASELINE.json metric).

Workload (config.workload): KITTI-shape stereo, 1241x376, ORBextractor.nFeatures=1000 per eye,
8 levels, scale 1.2, FAST 20/7.  One FRAME = one stereo pair = extract(left) + extract(right) +
ComputeStereoMatches.  A step = one batch of `--batch` frames (2*batch images in one set of
launches); inputs are synthetic (tools/synth.py) and already resident in HBM when timing starts.

`python bench.py --gpus N --steps K --warmup W`.  N > 1: one rank per GPU, independent camera streams per GPU, RCCL ("nccl") only for
the start / stop barrier and the MAX-reduction of the elapsed time -> "scaling": "weak".  The ranks come either from a launcher
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE is set) or, when the command line has no
launcher, from bench.py itself: the process becomes the PARENT of that same launcher command before it touches the GPU (launch_ranks),
relays rank 0's one JSON line and returns the launcher's exit code.  Every line carries what makes it self-evidencing: config.ranks_seen
(SUM-all-reduce of ones), config.rank_devices (PCI bus id / UUID / arch of the card each rank held), config.per_rank_frames_per_s, and
both north-star shapes as barrier-aligned whole-job legs (config.kitti2000_frames_per_s, config.tum640_frames_per_s).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (largest share of the HIP-event
time measured live on the launch stream during the timed region) by its ALGORITHMIC bytes
(DESIGN.md section "Kernels") and, under `roofline.issue`, by its measured VALU instruction count against
the chip's wave-instruction issue rate (the bound these integer kernels actually run into);
`cpu_baseline` times the CPU oracle (oracle/, a port: the reference itself needs OpenCV and cannot be
built) on a bounded sample of the same workload on rank 0, N=1.

Outside the timed region the headline run also (rank 0 / every rank as noted):
  * compares the LAST step's device outputs of every frame of the batch (keypoints, descriptors, uRight, depth) with the
    CPU oracle's results for the distinct synthetic pairs -> "verified"; a mismatch makes the exit code non-zero;
  * sweeps the batch size (frames per launch set) -> config.batch_sweep_frames_per_s;
  * measures the host-fed rates -> config.host_fed (and config.host_fed_batched_frames_per_s / host_fed_single_stream_frames_per_s at
    the top level): `batched` = pinned host frames -> H2D on a copy stream under the previous batch's kernels -> D2H of all results,
    four buffer sets going round (HostFedPipeline); `single_stream_c_abi` = one camera stream through orbx_extract_stereo_submit / _wait
    from a plain C client (examples/stereo_stream.c) with its latency distribution.  The headline `value` excludes PCIe;
  * times the KITTI 2000-feature workload (BASELINE configs 2 / 5) on every rank with its own barrier-aligned window ->
    config.kitti2000_frames_per_s (whole-job aggregate), so that an N-GPU line also carries BASELINE config 5;
  * runs short legs of the other BASELINE configs in the same process (N = 1, rank 0) -> config.other_configs: `tum640`
    (config 1: 640x480 mono @1000, CPU oracle beside the GPU rate), `euroc_bow` (config 3), `fhd4000` (config 4) and
    `euroc_track` (extract + ComputeBoW + SearchByBoW(previous keyframe, frame)), each with its own `verified`
    block (every frame of its last step against the oracle; BoW: every keyframe x 4 frames of the search rows) and a
    CPU-oracle figure on one thread (mono: src/Frame.cc:205 runs one extraction);
  * times the three north-star matchers per call next to the CPU oracle -> config.matchers (tools/matcher_bench.py).
`roofline.traffic` and `roofline.issue` come from committed rocprofv3 counter files (profiles/*_traffic.json, *_sq_counters.json),
not from this run: each is stamped with its source file and the hash of the kernel sources it was collected on, and nulled /
marked stale when orb-slam2_amd/csrc has changed since.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS = 1241, 376, 1000, 8   # headline workload; --workload may rebind W/H/NFEAT
WORKLOADS = {
    # name: (w, h, nfeatures, kind, default frames per step, metric string)
    "stereo1000": (1241, 376, 1000, "stereo", 256, "frames/sec ORB extract+match, 1241x376 @1000 feats; bit-exact vs CPU"),
    "stereo2000": (1241, 376, 2000, "stereo", 128, "frames/sec ORB extract + L/R stereo match, KITTI 1241x376 @2000 feats (BASELINE config 2)"),
    "euroc_bow": (752, 480, 1000, "bow", 32, "frames/sec ORB extract + SearchByBoW vs 500-KF map, EuRoC 752x480 @1000 feats (BASELINE config 3)"),
    "fhd4000": (1920, 1080, 4000, "mono", 64, "frames/sec ORB extract, 1920x1080 @4000 feats (BASELINE config 4)"),
    "tum640": (640, 480, 1000, "mono", 256, "frames/sec ORB extract, TUM 640x480 mono @1000 feats (BASELINE config 1)"),
    "euroc_track": (752, 480, 1000, "bow", 64, "frames/sec ORB extract + ComputeBoW + SearchByBoW(previous keyframe, frame), EuRoC 752x480 @1000 feats (SURVEY 8d)"),
}
BF, FX = 386.1448, 718.856          # reference Examples/Stereo/KITTI00-02.yaml:8,25
MIN_Z = BF / FX                     # mb = mbf/fx (src/Frame.cc:118)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def level_pixels(w, h, nlevels=NLEVELS, sf=1.2):
    """P_l per level with the reference's float arithmetic (src/ORBextractor.cc:436-461,1353)"""
    s = np.float32(1.0); out = []
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        out.append(int(np.rint(np.float32(w) * inv)) * int(np.rint(np.float32(h) * inv)))
        s = np.float32(np.float64(s) * np.float64(np.float32(sf)))
    return out


def algorithmic_bytes(stage, n_images, n_pairs, nkp_avg):
    """bytes one launch set of `stage` must move for n_images images (Kernels)"""
    P = level_pixels(W, H)
    if stage == "resize":      # read level l-1, write level l, l = 1..7 (7 launches per step)
        return n_images * (sum(P[:-1]) + sum(P[1:]))
    if stage == "fast":        # every level read once; candidates out (4 B each) not counted
        return n_images * sum(P)
    if stage == "tree":        # candidates in (4 B), kept keypoints out (4 B): ~3x quota
        return n_images * 4 * (4 * NFEAT)
    if stage == "desc":        # 43x43 patch in, 28 B keypoint + 32 B descriptor out
        return n_images * nkp_avg * (43 * 43 + 60)
    if stage == "stereo":      # SURVEY 8d B_stereo
        return n_pairs * (60 * 2 * nkp_avg + 8 * nkp_avg + nkp_avg * (121 + 21 * 11))
    return n_pairs * nkp_avg * 12


def _oracle_frame(oL, oR, left, right):
    """one stereo frame the way the reference runs it: L/R extraction on two threads (src/Frame.cc:82-85), then the matcher"""
    from oracle import oracle_py
    res = {}
    tl = threading.Thread(target=lambda: res.__setitem__("l", oL.extract(left)))
    tr = threading.Thread(target=lambda: res.__setitem__("r", oR.extract(right)))
    tl.start(); tr.start(); tl.join(); tr.join()
    ur, dp = oracle_py.stereo_match(oL, oR, res["l"][0], res["l"][1], res["r"][0], res["r"][1], BF, MIN_Z)
    return res["l"][0], res["l"][1], res["r"][0], res["r"][1], ur, dp


def oracle_results(pairs):
    """CPU oracle outputs for the distinct synthetic pairs: the checker of the `verified` block"""
    from oracle import oracle_py
    oL, oR = oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7), oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7)
    return [_oracle_frame(oL, oR, l, r) for l, r in pairs]


def cpu_baseline(frames, all_core_frames=4):
    """the CPU oracle driven like the reference: stereo = 2 threads (L/R), src/Frame.cc:82-85"""
    from oracle import oracle_py
    oL, oR = oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7), oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7)
    times = []
    for left, right in frames:
        t0 = time.perf_counter()
        _oracle_frame(oL, oR, left, right)
        times.append(time.perf_counter() - t0)
    times = np.array(times)
    out = {"value": round(float(len(times) / times.sum()), 3), "unit": "frames/s", "cores": 2, "kind": "port",
           "sample": f"{len(times)} stereo frames {W}x{H} @{NFEAT} feats through oracle/liborb_oracle.so: a scalar C port "
                     "(-O3 -march=native; no SIMD FAST / GaussianBlur / resize as OpenCV has them, so OpenCV itself would be several "
                     f"times faster), L/R extraction on 2 threads, median {np.median(times) * 1e3:.1f} ms/frame",
           "host_cpus": os.cpu_count()}
    # all cores, one independent camera stream per core (each stream extracts L then R on its own core)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, cores)               # every core the process may run on (one stream per core, core count stated)

    def run_streams(nthreads):
        done = [0] * nthreads

        def stream(ci):
            a, b = oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7), oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7)
            for i in range(all_core_frames):
                left, right = frames[(ci + i) % len(frames)]
                kl, dl = a.extract(left); kr, dr = b.extract(right)
                oracle_py.stereo_match(a, b, kl, dl, kr, dr, BF, MIN_Z)
                done[ci] += 1
        ths = [threading.Thread(target=stream, args=(ci,)) for ci in range(nthreads)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        return round(sum(done) / (time.perf_counter() - t0), 2)
    # one stream per core the affinity mask names; a container may grant fewer core-seconds than the mask suggests (a one-GPU box
    # of the pool: 256 in the mask), so smaller thread counts are tried too and the best is the figure
    sweep = {str(n): run_streams(n) for n in sorted({cores, min(cores, 64), min(cores, 16)})}
    best = max(sweep, key=sweep.get)
    out["all_cores"] = {"value": sweep[best], "unit": "frames/s", "cores": int(best), "affinity_cores": cores, "frames_per_s_by_threads": sweep,
                        "sample": f"independent streams, one thread each in the C oracle (every core of the affinity mask, and 64 / 16 threads), {all_core_frames} stereo frames per stream; value = the best"}
    return out


def cpu_baseline_mono(images, bow=None):
    """the CPU oracle on one thread (the reference extracts a monocular frame on the calling thread, src/Frame.cc:205); with
    `bow`: + Frame::ComputeBoW + SearchByBoW against every keyframe of the map (oracle vocabulary / matcher)"""
    from oracle import oracle_py
    o = oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7)
    times = []
    for img in images:
        t0 = time.perf_counter()
        k, d = o.extract(img)
        if bow is not None:
            t = bow["ovoc"].transform(d, 4)
            q = dict(desc=d, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"], flag=np.zeros(len(d), np.uint8), angle=k["angle"].copy())
            for kf in bow["kfs"]:
                oracle_py.search_by_bow_kf_f(kf, q, 0.75, True)
        times.append(time.perf_counter() - t0)
    times = np.array(times)
    what = "extract" if bow is None else f"extract + ComputeBoW + SearchByBoW against {len(bow['kfs'])} keyframes"
    return {"value": round(float(len(times) / times.sum()), 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{len(times)} frames {W}x{H} @{NFEAT} feats, {what}, through oracle/liborb_oracle.so on one thread, median {np.median(times) * 1e3:.1f} ms/frame",
            "host_cpus": os.cpu_count()}


def csrc_sha():
    """hash of the kernel sources: counter files under profiles/ say which sources they were collected on"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "orb-slam2_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def newest_profile(pattern):
    """newest committed counter file profiles/r??_<pattern> (by round tag) -> (path relative to the repo, document) or (None, None)"""
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    if not fs:
        return None, None
    return os.path.relpath(fs[-1], ROOT), json.load(open(fs[-1]))


class StereoRig:
    """device-resident batch of B stereo frames on one handle: the launch set of one bench step"""

    def __init__(self, pkg, torch, dev, local, w, h, nfeat, B, pairs, stream=None):
        self.pkg, self.torch, self.B, self.w, self.h = pkg, torch, B, w, h
        self.pitch = (w + 63) // 64 * 64
        host = np.zeros((2 * B, h, self.pitch), np.uint8)
        for i in range(B):
            host[i, :, :w] = pairs[i % len(pairs)][0]
            host[B + i, :, :w] = pairs[i % len(pairs)][1]
        self.imgs = torch.from_numpy(host).to(dev)
        self.ex = pkg.ORBextractor(nfeat, 1.2, NLEVELS, 20, 7, device=local, max_size=(w, h), max_batch=2 * B)
        self.cap = cap = self.ex.max_keypoints(w, h)
        self.kps = torch.zeros((2 * B, cap, 7), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((2 * B, cap, 32), dtype=torch.uint8, device=dev)
        self.nout = torch.zeros(2 * B, dtype=torch.int32, device=dev)
        self.ur = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        self.dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        self.stream = stream or torch.cuda.Stream(device=dev)
        self.sp = self.stream.cuda_stream

    def step(self, imgs=None):
        B, cap = self.B, self.cap
        src = self.imgs if imgs is None else imgs
        self.ex.extract_batch_device(src.data_ptr(), self.h * self.pitch, self.pitch, 2 * B, self.w, self.h, self.kps.data_ptr(),
                                     self.desc.data_ptr(), cap, self.nout.data_ptr(), self.sp)
        self.pkg.orbx.stereo_match_batch_device(self.ex, 0, self.ex, B, B, self.kps.data_ptr(), self.desc.data_ptr(), self.nout.data_ptr(),
                                                self.kps[B:].data_ptr(), self.desc[B:].data_ptr(), self.nout[B:].data_ptr(), cap,
                                                BF, MIN_Z, self.ur.data_ptr(), self.dp.data_ptr(), self.sp,
                                                row_table=self.pkg.orbx.ROWTAB_OF_EXTRACTION)   # kps[B:] is what the launch above wrote, untouched

    def verify(self, expect, npairs):
        """every frame of the batch (frame i holds pair i % npairs) against the oracle's outputs for that pair, byte for byte"""
        self.stream.synchronize()
        B, cap = self.B, self.cap
        n = self.nout.cpu().numpy()
        k = self.kps.cpu().numpy().view(np.uint8).reshape(2 * B, cap, 28)
        d = self.desc.cpu().numpy(); ur = self.ur.cpu().numpy(); dp = self.dp.cpu().numpy()
        bad = []
        for i in range(B):
            kL, dL, kR, dR, our, odp = expect[i % npairs]
            nl, nr = int(n[i]), int(n[B + i])
            ok = (nl == len(kL) and nr == len(kR) and k[i, :nl].tobytes() == kL.tobytes() and d[i, :nl].tobytes() == dL.tobytes() and
                  k[B + i, :nr].tobytes() == kR.tobytes() and d[B + i, :nr].tobytes() == dR.tobytes() and
                  ur[i, :nl].tobytes() == our.tobytes() and dp[i, :nl].tobytes() == odp.tobytes())
            if not ok:
                bad.append(i)
        return {"frames": B, "distinct_pairs": npairs, "checked": "keypoints (28 B each), descriptors, uRight, depth of every frame of the last step",
                "against": "oracle/liborb_oracle.so", "bit_exact": not bad, "mismatching_frames": bad[:8]}


class MonoRig:
    """device-resident batch of B monocular frames on one handle (BASELINE config 1 shape as a leg of every N-GPU line)"""

    def __init__(self, pkg, torch, dev, local, w, h, nfeat, B, images):
        self.B, self.w, self.h = B, w, h
        self.pitch = (w + 63) // 64 * 64
        host = np.zeros((B, h, self.pitch), np.uint8)
        for i in range(B):
            host[i, :, :w] = images[i % len(images)]
        self.imgs = torch.from_numpy(host).to(dev)
        self.ex = pkg.ORBextractor(nfeat, 1.2, NLEVELS, 20, 7, device=local, max_size=(w, h), max_batch=B)
        self.cap = cap = self.ex.max_keypoints(w, h)
        self.kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        self.nout = torch.zeros(B, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.sp = self.stream.cuda_stream

    def step(self):
        self.ex.extract_batch_device(self.imgs.data_ptr(), self.h * self.pitch, self.pitch, self.B, self.w, self.h, self.kps.data_ptr(),
                                     self.desc.data_ptr(), self.cap, self.nout.data_ptr(), self.sp)

    def verify(self, images):
        """every frame of the batch against the oracle's extraction of its image, byte for byte"""
        from oracle import oracle_py
        self.stream.synchronize()
        orc = oracle_py.Oracle(self.ex.nfeatures, 1.2, NLEVELS, 20, 7)
        ref = [orc.extract(m) for m in images]
        n = self.nout.cpu().numpy(); k = self.kps.cpu().numpy().view(np.uint8).reshape(self.B, self.cap, 28); d = self.desc.cpu().numpy()
        bad = [i for i in range(self.B) if not (int(n[i]) == len(ref[i % len(images)][0]) and k[i, :n[i]].tobytes() == ref[i % len(images)][0].tobytes() and
                                                d[i, :n[i]].tobytes() == ref[i % len(images)][1].tobytes())]
        return {"frames": self.B, "distinct_images": len(images), "against": "oracle/liborb_oracle.so", "bit_exact": not bad, "mismatching_frames": bad[:8]}


class HostFedPipeline:
    """Batches of B stereo frames that START IN PINNED HOST MEMORY and END THERE: H2D of batch k+1 on a copy stream while batch k computes,
    D2H of batch k-1's results on a second copy stream.  `depth` buffer sets (device images + workspaces + device / pinned result blocks) go
    round: with two, the back-pressure waits (an upload into buffer j waits for the compute that last read it, a compute into result block j
    waits for the download that last read it) sit on the critical path and the link idles a quarter of the time (42 k frames/s against 56 k
    with the forward dependencies alone, tools/hostfed_batched_diag.py); with four they refer to work that finished a batch time ago.  All
    compute runs on ONE stream (it is serial anyway): compute + two copy streams stay within the four hardware queues HIP multiplexes its
    streams onto by default (GPU_MAX_HW_QUEUES), where a fifth stream would share a queue with a copy stream and serialise behind it.
    One FRAME crosses PCIe as 2 images in and n / keypoints / descriptors / uRight / depth out."""

    def __init__(self, pkg, torch, dev, local, w, h, nfeat, B, depth=4):
        self.torch, self.B, self.depth = torch, B, depth
        self.compute = torch.cuda.Stream(device=dev)
        blank = [(np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8))]
        self.rigs = [StereoRig(pkg, torch, dev, local, w, h, nfeat, B, blank, stream=self.compute) for _ in range(depth)]
        self.h_in = [torch.zeros(r.imgs.shape, dtype=torch.uint8).pin_memory() for r in self.rigs]
        self.h_out = [[torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in (r.nout, r.kps, r.desc, r.ur, r.dp)] for r in self.rigs]
        self.cin, self.cout = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.ev_in = [torch.cuda.Event() for _ in self.rigs]; self.ev_done = [torch.cuda.Event() for _ in self.rigs]; self.ev_out = [torch.cuda.Event() for _ in self.rigs]
        for j in range(depth):
            self.ev_done[j].record(self.compute); self.ev_out[j].record(self.cout)
        self.k = 0

    def host_images(self, j):
        """the pinned [2B, h, pitch] block of buffer set j (left eyes first): the caller writes the next batch here"""
        return self.h_in[j]

    def submit(self):
        """enqueue upload -> extraction + stereo match -> download of the batch in buffer set k % depth; returns that index"""
        torch, j = self.torch, self.k % self.depth
        r = self.rigs[j]
        with torch.cuda.stream(self.cin):
            self.cin.wait_event(self.ev_done[j])          # the batch that last used these device images has been computed
            r.imgs.copy_(self.h_in[j], non_blocking=True)
            self.ev_in[j].record(self.cin)
        self.compute.wait_event(self.ev_in[j])
        self.compute.wait_event(self.ev_out[j])           # its previous results have left the device result block
        r.step()
        self.ev_done[j].record(self.compute)
        with torch.cuda.stream(self.cout):
            self.cout.wait_event(self.ev_done[j])
            for ht, dt in zip(self.h_out[j], (r.nout, r.kps, r.desc, r.ur, r.dp)):
                ht.copy_(dt, non_blocking=True)
            self.ev_out[j].record(self.cout)
        self.k += 1
        return j

    def results(self, j):
        """(n[2B], keypoints[2B, cap, 28 bytes], descriptors[2B, cap, 32], uRight[B, cap], depth[B, cap]) of buffer set j, on the host, once its download is done"""
        self.ev_out[j].synchronize()
        n, k, d, ur, dp = (t.numpy() for t in self.h_out[j])
        return n, k.view(np.uint8).reshape(k.shape[0], k.shape[1], 28), d, ur, dp

    def bytes_per_frame(self):
        return int(self.h_in[0].numel() / self.B), int(sum(t.numel() * t.element_size() for t in self.h_out[0]) / self.B)


def host_fed_batched(pkg, torch, dev, local, pairs, B, steps, depth=4):
    """frames/s of the batched host-fed path (HostFedPipeline) over `steps` batches of B frames, inputs in pinned host memory"""
    pipe = HostFedPipeline(pkg, torch, dev, local, W, H, NFEAT, B, depth)
    for j in range(depth):
        hi = pipe.host_images(j).numpy()
        for i in range(B):
            hi[i, :, :W] = pairs[i % len(pairs)][0]; hi[B + i, :, :W] = pairs[i % len(pairs)][1]
    for i in range(2 * depth):
        pipe.submit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.submit()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    in_b, out_b = pipe.bytes_per_frame()
    return {"frames_per_s": round(B * steps / el, 1), "frames_per_batch": B, "buffer_sets": depth, "batches": steps,
            "h2d_bytes_per_frame": in_b, "d2h_bytes_per_frame": out_b,
            "pcie_gbytes_per_s": round((in_b + out_b) * B * steps / el / 1e9, 2),
            "h2d_gbytes_per_s": round(in_b * B * steps / el / 1e9, 2)}


def host_fed_c_client(streams, frames=1500):
    """one camera stream (or several) through the pipelined host-pointer C ABI, measured by the plain C client"""
    import subprocess
    exe = os.path.join(ROOT, "examples", "stereo_stream")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--streams", str(streams), "--frames", str(frames), "--nfeat", str(NFEAT)], capture_output=True, text=True, timeout=180)
        return json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else {"error": (out.stderr or out.stdout)[-200:]}
    except Exception as exc:    # measurement leg only: never fail the bench line over it
        return {"error": str(exc)[:200]}


def run_workload(ctx, args):
    """one workload: warm-up, stage breakdown, timed region, verification, roofline -> the dict of its JSON line"""
    global W, H, NFEAT
    W, H, NFEAT, kind, def_batch, metric = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = def_batch
    torch, pkg, st, dev, local, rank, world, dist_dev = (ctx[k] for k in ("torch", "pkg", "st", "dev", "local", "rank", "world", "dist_dev"))
    from tools import synth

    B = args.batch
    stream_id = st.owned_streams(rank, world)[0]      # one camera stream per GPU
    seed0 = st.stream_seed(stream_id)
    pitch = (W + 63) // 64 * 64
    stereo = kind == "stereo"
    NI = 2 * B if stereo else B                        # images per step
    rig = None
    if stereo:
        pairs = [synth.stereo_pair(seed0 + i, W, H)[:2] for i in range(args.distinct)]
        rig = StereoRig(pkg, torch, dev, local, W, H, NFEAT, B, pairs)
        ex, imgs, kps, desc, nout, ur, stream, cap = rig.ex, rig.imgs, rig.kps, rig.desc, rig.nout, rig.ur, rig.stream, rig.cap
    else:
        pairs = None
        host = np.zeros((NI, H, pitch), np.uint8)
        monos = [synth.image(seed0 + i, W, H, nshapes=max(400, W * H // 311)) for i in range(args.distinct)]
        for i in range(B):
            host[i, :, :W] = monos[i % len(monos)]
        imgs = torch.from_numpy(host).to(dev)
        ex = pkg.ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local, max_size=(W, H), max_batch=NI)
        cap = ex.max_keypoints(W, H)
        kps = torch.zeros((NI, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.zeros((NI, cap, 32), dtype=torch.uint8, device=dev)
        nout = torch.zeros(NI, dtype=torch.int32, device=dev)
        ur = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream

    def extract():
        ex.extract_batch_device(imgs.data_ptr(), H * pitch, pitch, NI, W, H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), sp)

    bow, bow_cpu = None, None
    NKF = args.keyframes if args.keyframes > 0 else (1 if args.workload == "euroc_track" else 500)
    if kind == "bow":
        # BASELINE config 3: 500-keyframe synthetic map: keyframe descriptors = frame descriptors
        # with Bernoulli(0.08) bit flips, shuffled; hasGoodMP ~ Bernoulli(0.6); FeatureVector from a seeded
        # two-level slice of a k=10 vocabulary (levelsup=4 of L=6; the real ORBvoc.txt is absent).
        extract(); stream.synchronize()
        n0 = nout.cpu().numpy()
        rng = np.random.Generator(np.random.PCG64(seed0 + 7))
        k_h = kps.cpu().numpy().view(np.uint8).reshape(NI, cap, 28)
        d_h = desc.cpu().numpy()
        # synthetic k=10, L=6 vocabulary (1 111 110 nodes): levels 1-2 seeded from real descriptors so that
        # features spread over the <=100 level-2 nodes the FeatureVector uses (levelsup = 4), deeper nodes random
        d0 = d_h[0, :n0[0]]
        par, leaf, nd, wt = synth.vocab_tree(seed0 + 9, 10, 6, stop_frac=0.0, data=None)
        nd[:110] = synth.flip_bits(rng, d0[rng.integers(0, len(d0), 110)], 0.1)
        voc = pkg.ORBVocabulary(10, 6, par, leaf, nd, wt, device=local)
        frames_fs = []
        for i in range(min(B, args.distinct)):
            kp = np.frombuffer(k_h[i, :n0[i]].tobytes(), dtype=pkg.KP_DTYPE)
            dd = d_h[i, :n0[i]].copy()
            t = voc.transform(dd, 4)
            frames_fs.append(dict(desc=dd, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"],
                                  flag=np.zeros(len(dd), np.uint8), angle=kp["angle"].copy()))
        kfs = []
        base = frames_fs[0]
        for _ in range(NKF):
            perm = rng.permutation(len(base["desc"]))
            dk = synth.flip_bits(rng, base["desc"], 0.08)[perm]
            t = voc.transform(dk, 4)
            kfs.append(dict(desc=dk, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"],
                            flag=(rng.random(len(dk)) < 0.6).astype(np.uint8), angle=base["angle"][perm]))
        bow = {"db": pkg.BowDatabase(kfs, device=local), "voc": voc, "frames": frames_fs, "ms": 0.0, "tms": 0.0, "queries": 0, "matches": 0,
               "fr": pkg.BowFrames(NI, cap, device=local),
               "d_match": torch.zeros((NI, NKF, cap), dtype=torch.int32, device=dev), "d_nm": torch.zeros((NI, NKF), dtype=torch.int32, device=dev),
               # compact result form (orbx_bowdb_search_batch_device_compact): per (frame, keyframe) the (frame feature, keyframe feature) pairs
               "cap_pairs": cap, "d_pairs": torch.zeros((NI, NKF, cap, 2), dtype=torch.int32, device=dev), "compact": not args.bow_dense,   # (capacity = every feature: a list is never cut; only the matches are written)
               "kfs": kfs, "vocab_arrays": (par, leaf, nd, wt),
               "ev": [], "host_path": args.bow_host_path, "kf_feats": int(sum(len(k_["desc"]) for k_ in kfs)),
               "kf_list": int(sum(len(k_["feat"]) for k_ in kfs)), "kf_nodes": int(sum(len(k_["node_id"]) for k_ in kfs))}

    def step():
        if bow is not None and prof_on[0]:
            ex.profile_enable(True)   # BoW work sits between two extractions on this stream: start a fresh event chain
        if stereo:
            rig.step()
            return
        extract()
        if bow is not None and not bow["host_path"]:
            # device-resident chain: Frame::ComputeBoW for the batch, then every keyframe against every frame, all on `sp`
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
            bow["fr"].transform(bow["voc"], kps.data_ptr(), desc.data_ptr(), nout.data_ptr(), B, 4, sp)
            e1.record(stream)
            if bow["compact"]:
                bow["fr"].search_compact(bow["db"], B, bow["d_pairs"].data_ptr(), bow["cap_pairs"], bow["d_nm"].data_ptr(), 0.75, True, sp)
            else:
                bow["fr"].search(bow["db"], B, bow["d_match"].data_ptr(), bow["d_nm"].data_ptr(), 0.75, True, sp)
            e2.record(stream)
            bow["ev"].append((e0, e1, e2))
            bow["queries"] += B
        elif bow is not None:
            tq = time.perf_counter()
            for i in range(B):                      # Frame::ComputeBoW, then the Tracking::Relocalization loop over keyframes
                fr = bow["frames"][i % len(bow["frames"])]
                tt = time.perf_counter()
                t = bow["voc"].transform(fr["desc"], 4)
                bow["tms"] += (time.perf_counter() - tt) * 1e3
                q = dict(fr); q["node_id"], q["node_off"], q["feat"] = t["fv_node_id"], t["fv_node_off"], t["fv_feat"]
                m, n = bow["db"].search(q, 0.75, True)
                bow["matches"] += int(n.sum())
            bow["ms"] += (time.perf_counter() - tq) * 1e3
            bow["queries"] += B

    prof_on = [False]

    def local_sync():
        stream.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    local_sync()
    if bow is not None:
        bow["ms"], bow["tms"], bow["queries"], bow["matches"] = 0.0, 0.0, 0, 0
        bow["ev"] = []
    # Stage breakdown: a few UNTIMED steps with events at every stage boundary (`roofline.stage_ms_per_step`).  An event between
    # two kernels is a barrier packet, ~6 us of idle GPU each (13 launches per step: 3-4 % of a step), so the TIMED region only
    # brackets the kernel whose duration the roofline divides by -- the dominant stage of that breakdown -- live, on its stream.
    ex.profile_read(reset=True)
    prof_on[0] = True
    ex.profile_stages(0xFFFFFFFF)
    ex.profile_enable(True)
    n_prof = max(2, min(10, args.steps))
    for _ in range(n_prof):
        step()
    local_sync()
    prof_all = ex.profile_read(reset=True)
    stage_all_ms = {k: v[0] / n_prof for k, v in prof_all.items()}
    dom_stage = max(stage_all_ms, key=stage_all_ms.get)
    ex.profile_stages(1 << pkg.orbx.STAGES.index(dom_stage))
    tdet = {}
    elapsed = st.timed_steps(step, args.steps, local_sync, world, device=dist_dev, detail=tdet)   # barrier + sync both sides, MAX over ranks
    ex.profile_enable(False)
    prof = ex.profile_read(reset=True)
    ex.profile_stages(0xFFFFFFFF)
    ex.sync(sp)

    n_h = nout.cpu().numpy()
    nkp_avg = float(n_h.mean())
    matched = float((ur.cpu().numpy() >= 0).sum() / B) if stereo else 0.0
    value = st.aggregate_rate(B, args.steps, world, elapsed)

    # ---- outside the timed region: the last step's outputs against the CPU oracle (rank 0)
    verified, expect = None, None
    if stereo and rank == 0 and not args.no_verify:
        expect = oracle_results(pairs)
        verified = rig.verify(expect, len(pairs))

    if not stereo and rank == 0 and not args.no_verify:
        # every frame of the batch against the oracle's extraction of its image; BoW: + the search rows of every keyframe x 4
        # frames against the oracle's ComputeBoW + SearchByBoW (the keyframes' FeatureVectors re-derived by the oracle too)
        from oracle import oracle_py
        orc = oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7)
        ref = [orc.extract(m) for m in monos]
        k_all = kps.cpu().numpy().view(np.uint8).reshape(NI, cap, 28); d_all = desc.cpu().numpy()
        badf = [i for i in range(B) if not (int(n_h[i]) == len(ref[i % len(monos)][0]) and
                                            k_all[i, :n_h[i]].tobytes() == ref[i % len(monos)][0].tobytes() and
                                            d_all[i, :n_h[i]].tobytes() == ref[i % len(monos)][1].tobytes())]
        verified = {"frames": B, "distinct_images": len(monos), "checked": "keypoints (28 B each) and descriptors of every frame of the last step",
                    "against": "oracle/liborb_oracle.so", "bit_exact": not badf, "mismatching_frames": badf[:8]}
        if bow is not None:
            par, leaf, nd, wt = bow["vocab_arrays"]
            ovoc = oracle_py.Vocabulary(10, 6, par, leaf, nd, wt)
            bow_cpu = {"ovoc": ovoc, "kfs": bow["kfs"]}
            if not bow["host_path"]:
                nkc, nfc = NKF, min(B, 4)          # EVERY keyframe of the map (config 3: all 500) x the first four frames of the launch
                # both result forms of the last launch's inputs: the timed one as it stands, the other from one more (untimed) launch
                dn_c = dn_d = None
                if bow["compact"]:
                    dn_c = bow["d_nm"][:nfc, :nkc].cpu().numpy().copy(); dpairs = bow["d_pairs"][:nfc, :nkc].cpu().numpy()
                    bow["fr"].search(bow["db"], B, bow["d_match"].data_ptr(), bow["d_nm"].data_ptr(), 0.75, True, sp); stream.synchronize()
                    dn_d = bow["d_nm"][:nfc, :nkc].cpu().numpy().copy()
                else:
                    dn_d = bow["d_nm"][:nfc, :nkc].cpu().numpy().copy()
                    bow["fr"].search_compact(bow["db"], B, bow["d_pairs"].data_ptr(), bow["cap_pairs"], bow["d_nm"].data_ptr(), 0.75, True, sp); stream.synchronize()
                    dn_c = bow["d_nm"][:nfc, :nkc].cpu().numpy().copy(); dpairs = bow["d_pairs"][:nfc, :nkc].cpu().numpy()
                dm = bow["d_match"][:nfc, :nkc].cpu().numpy(); dn = dn_d
                badp = []
                same_fv = []
                for kq in range(nkc):               # the keyframes' FeatureVectors re-derived by the oracle, once each
                    kf = bow["kfs"][kq]
                    tk = ovoc.transform(kf["desc"], 4)
                    same_fv.append(np.array_equal(tk["fv_node_id"], kf["node_id"]) and np.array_equal(tk["fv_node_off"], kf["node_off"]) and np.array_equal(tk["fv_feat"], kf["feat"]))
                for f in range(nfc):
                    ok_, od_ = ref[f % len(monos)]
                    t = ovoc.transform(od_, 4)
                    q = dict(desc=od_, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"], flag=np.zeros(len(od_), np.uint8), angle=ok_["angle"].copy())
                    for kq in range(nkc):
                        exp, en = oracle_py.search_by_bow_kf_f(bow["kfs"][kq], q, 0.75, True)
                        slots = np.nonzero(exp >= 0)[0]
                        nc_ = int(dn_c[f, kq])
                        ok_compact = nc_ == en and nc_ <= bow["cap_pairs"] and np.array_equal(dpairs[f, kq, :nc_, 0], slots) and np.array_equal(dpairs[f, kq, :nc_, 1], exp[slots])
                        if not (same_fv[kq] and en == int(dn[f, kq]) and np.array_equal(dm[f, kq, :len(exp)], exp) and ok_compact):
                            badp.append((f, kq))
                verified["bow"] = {"pairs_checked": nfc * nkc, "what": f"search results of keyframes 0..{nkc - 1} x frames 0..{nfc - 1} of the last launch in BOTH forms (dense rows and compact (frame feature, keyframe feature) lists), "
                                                                      "and those keyframes' FeatureVectors, against oracle ComputeBoW + SearchByBoW", "timed_form": "compact" if bow["compact"] else "dense",
                                       "mismatching_pairs": badp[:8]}
                verified["bit_exact"] = verified["bit_exact"] and not badp

    # dominant kernel + roofline (per launch: total stage time / launches; resize = 7 launches per step)
    stage_ms = {k: v[0] for k, v in prof.items()}     # timed region: only the dominant stage has events
    dom = dom_stage
    launches = max(prof[dom][1], 1)
    avg_ms = stage_ms[dom] / launches
    per_step_launches = launches / args.steps
    bytes_per_launch = algorithmic_bytes(dom, NI, B, nkp_avg) / per_step_launches
    bow_dev, bow_models = None, None
    if bow is not None and bow["ev"]:
        t_tr = sum(a.elapsed_time(b_) for a, b_, _ in bow["ev"]); t_se = sum(b_.elapsed_time(c) for _, b_, c in bow["ev"])
        bow_dev = {"transform_ms_per_step": round(t_tr / len(bow["ev"]), 4), "search_ms_per_step": round(t_se / len(bow["ev"]), 4)}
        bow["matches"] = int(bow["d_nm"].sum().item()) * len(bow["ev"])
        if t_se > stage_ms[dom]:   # the search launch dominates: one launch per step
            dom, avg_ms, per_step_launches = "bow<0>", t_se / len(bow["ev"]), 1.0
            # COMPULSORY bytes of one launch (the map side is read once per query batch): the 500-keyframe map
            # (list-order descriptors 32 B + flag 1 B per listed feature, node ids 4 B + offsets 4 B per node, angles 4 B per
            # feature) once + the B frames' sides once + one int32 match row per (keyframe, frame) pair out.  The round-1 model
            # (SURVEY's B_bow per pair x 16 000 pairs) re-counted the frame side 500x and the map side B x: kept as
            # `per_pair_model_bytes` for comparison.
            frame_side = B * nkp_avg * (32 + 1 + 4 + 4) + B * 100 * 8
            map_side = bow["kf_list"] * 33 + bow["kf_nodes"] * 8 + bow["kf_feats"] * 4
            # results out: dense = one int32 row per pair; compact = 8 bytes per match + the count
            out_bytes = NKF * B * 4 * nkp_avg if not bow["compact"] else 8 * bow["matches"] / len(bow["ev"]) + 4 * NKF * B
            bytes_per_launch = map_side + frame_side + out_bytes
            bow_models = {"compulsory_bytes": int(bytes_per_launch), "of_which_results_out": int(out_bytes), "result_form": "compact" if bow["compact"] else "dense",
                          "per_pair_model_bytes": int(NKF * B * (32 + 4 + 1 + 4) * 2 * nkp_avg + NKF * B * 4 * nkp_avg)}
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch from the rocprofv3 PMC passes (tools/collect_pmc.py): a committed file, not this run -- stamped with its
    # source and nulled when the kernel sources have changed since it was collected
    traffic, traffic_src = None, None
    sha_now = csrc_sha()
    tpath, tj = newest_profile("traffic.json" if args.workload == "stereo1000" else "traffic_" + args.workload + ".json")
    if tj:
        kern = tj.get("kernels", {}).get("k_" + dom.split("<")[0])
        fresh = tj.get("csrc_sha") == sha_now
        traffic_src = f"{tpath}@csrc {tj.get('csrc_sha', 'unstamped')}" + ("" if fresh else f" -- STALE (sources now {sha_now}): not reported")
        if kern and tj.get("images_per_launch") and fresh:
            traffic = int(kern["hbm_bytes_per_launch"] * NI / tj["images_per_launch"])
    step_ms = {k: round(v, 4) for k, v in stage_all_ms.items()}
    step_ms[dom_stage] = round(stage_ms[dom_stage] / args.steps, 4)
    roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "stage_ms_per_step": step_ms,
                "stage_ms_source": f"{dom_stage}: HIP events inside the timed region; other stages: {n_prof} untimed steps with events at every stage boundary"}
    if stereo:   # every stage of the step against the same HBM roof: algorithmic bytes of the stage / its time per step
        roofline["stage_hbm_frac"] = {k: round(algorithmic_bytes(k if k != "stereo_cut" else "cut", NI, B, nkp_avg) / max(v, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)
                                      for k, v in step_ms.items() if v > 0}
    if bow_models:
        roofline["byte_models"] = bow_models
    ipath, im = newest_profile("sq_counters.json")
    if im and args.workload == "stereo1000" and ("k_" + dom) in im.get("kernels", {}):
        kk = im["kernels"]["k_" + dom]
        if im.get("csrc_sha") != sha_now:
            roofline["issue"] = {"stale": True, "source": f"{ipath}@csrc {im.get('csrc_sha', 'unstamped')}; sources now {sha_now}: not reported"}
        else:
            per_image = kk["SQ_INSTS_VALU"] / im["images_per_launch"]
            valu = per_image * NI / per_step_launches
            cyc = im["issue_model"]["cycles_per_valu_inst"]["k_" + dom]
            peak = im["issue_model"]["simds"] * im["issue_model"]["clock_ghz"] * 1e9 / cyc
            roofline["issue"] = {"valu_insts": int(valu), "cycles_per_valu_inst": cyc, "peak_wave_insts_per_s": round(peak, 0),
                                 "frac": round(valu / (avg_ms * 1e-3) / peak, 4), "source": f"{ipath}@csrc {im['csrc_sha']}",
                                 "note": "MODEL-DERIVED: wave64 VALU instructions per launch (rocprofv3 SQ_INSTS_VALU from the committed counter file, not this run) / "
                                         "this run's launch time, against 1024 SIMDs x clock / mix-weighted cycles per instruction measured by "
                                         "tools/ubench/op_cost.hip (DESIGN.md section 5)"}
            if roofline["issue"]["frac"] > 1.0:
                roofline["issue"]["reading"] = ("above 1: the kernel issues its VALU instructions faster than the price list says is possible, i.e. the mix-weighted cost "
                                                "(measured on micro-kernels at 8 waves per SIMD) overstates this kernel's by that much; read it as 'the vector ALUs are the "
                                                "bound', not as a utilisation")
            if "SQ_ACTIVE_INST_VALU" in kk and kk.get("launch_ms"):
                # counter-only figure (no price list): SIMD-cycles the vector ALUs were busy / SIMD-cycles of the launch, both from the counter run
                roofline["issue"]["valu_busy_frac_counters"] = round(kk["SQ_ACTIVE_INST_VALU"] * 4 / (im["issue_model"]["simds"] * kk["launch_ms"] * 1e-3 * im["issue_model"]["clock_ghz"] * 1e9), 4)

    desc_txt = {"stereo": f"KITTI-shape stereo {W}x{H}, nFeatures={NFEAT}/eye, 8 levels, FAST 20/7: "
                          "extract(L)+extract(R)+ComputeStereoMatches; 1 frame = 1 stereo pair; images RESIDENT IN HBM when the timed region "
                          "starts (kernel throughput: excludes PCIe; fed from pinned host memory the same path runs at config.host_fed_batched_frames_per_s "
                          "-- the PCIe Gen5 x16 link, 0.96 MB per frame -- and one camera stream at config.host_fed_single_stream_frames_per_s)",
                "mono": f"mono {W}x{H}, nFeatures={NFEAT}, 8 levels, FAST 20/7: extract only",
                "bow": f"mono {W}x{H}, nFeatures={NFEAT}: extract + ComputeBoW (synthetic k=10 L=6 vocabulary on device) + SearchByBoW(ratio 0.75, "
                       f"checkOri) of every frame against a device-resident {NKF}-keyframe synthetic map; device-resident chain "
                       "(orbx_bow_transform_batch_device -> orbx_bowdb_search_batch_device[_compact]), matches stay in HBM as " +
                       ("dense match rows" if args.bow_dense else "compact (frame feature, keyframe feature) lists, what Tracking::Relocalization hands to its PnP solver")}[kind]
    out = {"metric": metric, "value": round(value, 2),
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": desc_txt, "name": args.workload, "csrc_sha": sha_now,
                      "frames_per_step_per_gpu": B, "images_per_step_per_gpu": NI,
                      "keypoints_per_image": round(nkp_avg, 1), "stereo_matches_per_frame": round(matched, 1),
                      "parallelism": f"{world} independent camera-stream batches, one per GPU"},
           "roofline": roofline}
    if args.workload == "stereo1000" or world > 1:
        # what makes an N-rank line self-evidencing: how many ranks met (SUM of ones), which card each held, each rank's own rate
        out["config"]["ranks_seen"] = st.ranks_seen(world, dist_dev)
        out["config"]["rank_devices"] = st.gather_strings(pkg.orbx.device_identity(local), world, dist_dev)
        out["config"]["per_rank_frames_per_s"] = [round(B * args.steps / t, 1) for t in tdet["per_rank"]]
        out["config"]["backend"] = ctx.get("backend", "none")
    if verified is not None:
        out["verified"] = verified
    if bow is not None and bow["queries"]:
        if bow_dev:
            out["config"]["bow_device_chain"] = bow_dev
        else:
            out["config"]["bow_ms_per_query_frame_incl_host"] = round(bow["ms"] / bow["queries"], 4)
            out["config"]["of_which_bow_transform_ms"] = round(bow["tms"] / bow["queries"], 4)
        out["config"]["bow_matches_per_query_frame"] = round(bow["matches"] / bow["queries"], 1)
        out["config"]["keyframes"] = NKF

    # ---- extra legs, all outside the timed headline region
    if stereo and args.extras and args.workload == "stereo1000":
        # BASELINE configs 2 / 5 on every rank: KITTI 2000 features per eye, its own barrier-aligned window
        del rig, imgs, kps, desc, nout, ur
        torch.cuda.empty_cache()
        B2 = max(B // 2, 1)
        pairs2 = pairs[:4]
        rig2 = StereoRig(pkg, torch, dev, local, W, H, 2000, B2, pairs2)
        for _ in range(2):
            rig2.step()

        def sync2():
            rig2.stream.synchronize(); torch.cuda.synchronize()
        k2 = max(5, args.steps // 3)
        el2 = st.timed_steps(rig2.step, k2, sync2, world, device=dist_dev)
        out["config"]["kitti2000_frames_per_s"] = {"value": round(st.aggregate_rate(B2, k2, world, el2), 1), "frames_per_step_per_gpu": B2, "steps": k2,
                                                   "keypoints_per_image": round(float(rig2.nout.float().mean().item()), 1),
                                                   "what": "BASELINE configs 2 / 5: 1241x376 @2000 feats per eye, extract L+R + stereo match, whole-job aggregate over n_gpus"}
        del rig2
        torch.cuda.empty_cache()
        # BASELINE config 1 shape (north_star: "synthetic 640x480 / 1241x376 ... at 1, 2, 4 and 8 GPUs") on every rank, its own barrier-aligned window
        monos = [synth.image(seed0 + 50 + i, 640, 480) for i in range(4)]
        rig3 = MonoRig(pkg, torch, dev, local, 640, 480, 1000, B, monos)
        for _ in range(2):
            rig3.step()

        def sync3():
            rig3.stream.synchronize(); torch.cuda.synchronize()
        k3 = max(5, args.steps // 3)
        d3 = {}
        el3 = st.timed_steps(rig3.step, k3, sync3, world, device=dist_dev, detail=d3)
        out["config"]["tum640_frames_per_s"] = {"value": round(st.aggregate_rate(B, k3, world, el3), 1), "frames_per_step_per_gpu": B, "steps": k3,
                                                "keypoints_per_image": round(float(rig3.nout.float().mean().item()), 1),
                                                "per_rank_frames_per_s": [round(B * k3 / t, 1) for t in d3["per_rank"]],
                                                "what": "BASELINE config 1 shape: 640x480 mono @1000 feats, extract only, whole-job aggregate over n_gpus"}
        if rank == 0 and not args.no_verify:
            out["config"]["tum640_frames_per_s"]["verified"] = rig3.verify(monos)
            if not out["config"]["tum640_frames_per_s"]["verified"]["bit_exact"]:
                out["verified"] = dict(out.get("verified") or {}, bit_exact=False, tum640_leg_mismatch=True)
        del rig3
        torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            sweep = {}
            for bs in (1, 8, 64, 256):
                r = StereoRig(pkg, torch, dev, local, W, H, NFEAT, bs, pairs)
                for _ in range(6):
                    r.step()
                r.stream.synchronize()
                ks = max(20, min(200, 2048 // bs))
                t0 = time.perf_counter()
                for _ in range(ks):
                    r.step()
                r.stream.synchronize()
                sweep[str(bs)] = round(bs * ks / (time.perf_counter() - t0), 1)
                del r
            out["config"]["batch_sweep_frames_per_s"] = sweep
            # the same 256 frames as two half-batches on two handles / streams, submitted round-robin: the launch chains of the halves overlap
            # (one half's latency-bound quadtree and stereo kernels run beside the other's FAST) -- a deployment option, not the headline
            halves = [StereoRig(pkg, torch, dev, local, W, H, NFEAT, B // 2, pairs) for i in range(2)] if B >= 2 else []
            if halves:
                for _ in range(6):
                    for r in halves:
                        r.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    for r in halves:
                        r.step()
                torch.cuda.synchronize()
                out["config"]["two_half_batches_on_two_streams_frames_per_s"] = round(2 * (B // 2) * 30 / (time.perf_counter() - t0), 1)
                del halves
            # B = 1 is the reference's operating point (one stereo frame at a time, Examples/Stereo/stereo_kitti.cc:68-117): its time is the
            # chain of dependent launches of one frame
            out["config"]["single_frame"] = {"us_per_frame": round(1e6 / sweep["1"], 1), "kernel_launches_per_stereo_frame": 6,
                                             "launches": ["k_pyr_group (levels 1-2)", "k_pyr_group (levels 3-7)", "k_fast", "k_tree", "k_desc (+ stereo row table)",
                                                          "k_stereo (+ median cut)"],
                                             "trace": "profiles/r03_b1_launch_chain.txt (rocprofv3 --kernel-trace, tools/b1_chain.sh)"}
            torch.cuda.empty_cache()
            hf = {"batched": host_fed_batched(pkg, torch, dev, local, pairs, 128, 30)}
            torch.cuda.synchronize()
            hf["single_stream_c_abi"] = host_fed_c_client(1)
            hf["four_streams_c_abi"] = host_fed_c_client(4)
            out["config"]["host_fed"] = hf
            # the rates a deployment can actually be fed at, at the top level of config (the headline `value` is kernel throughput on resident images)
            out["config"]["host_fed_batched_frames_per_s"] = hf["batched"]["frames_per_s"]
            out["config"]["host_fed_batched_pcie_gbytes_per_s"] = hf["batched"]["pcie_gbytes_per_s"]
            if isinstance(hf["single_stream_c_abi"], dict) and "frames_per_s" in hf["single_stream_c_abi"]:
                out["config"]["host_fed_single_stream_frames_per_s"] = hf["single_stream_c_abi"]["frames_per_s"]
    if rank == 0 and world == 1 and args.cpu_frames > 0 and stereo:
        out["cpu_baseline"] = cpu_baseline([pairs[i % len(pairs)] for i in range(args.cpu_frames)])
    elif rank == 0 and world == 1 and args.cpu_frames > 0:
        if bow is not None and bow_cpu is None:
            from oracle import oracle_py
            bow_cpu = {"ovoc": oracle_py.Vocabulary(10, 6, *bow["vocab_arrays"]), "kfs": bow["kfs"]}
        out["cpu_baseline"] = cpu_baseline_mono([monos[i % len(monos)] for i in range(args.cpu_frames)], bow_cpu)
    elif rank == 0:
        out["cpu_baseline"] = None
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process becomes the PARENT of `python -m torch.distributed.run
    --nproc-per-node N bench.py <same arguments>` (one rank per GPU), relays the ranks' output -- rank 0's one JSON line on stdout --
    and returns the launcher's exit code.  The parent never imports torch, never loads liborbx.so and never execs: nothing here
    touches the GPU, the ranks are ordinary child processes (one process per GPU)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in proc.stdout:                                  # stderr is inherited; stdout is relayed as it comes
        if line.startswith("{"):
            lines += 1
        sys.stdout.write(line); sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, saw {lines}", file=sys.stderr)
        return 4
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0 = the workload's default)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="stereo1000",
                    help="stereo1000 is the BASELINE.json headline metric; the others are the remaining single-GPU configs")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames of the CPU-baseline sample (0 = skip; default 240 stereo / 6 otherwise)")
    ap.add_argument("--bow-host-path", action="store_true", help="euroc_bow: per-frame host-pointer ComputeBoW + search (round-1 form)")
    ap.add_argument("--bow-dense", action="store_true", help="BoW workloads: time the dense match rows instead of the compact (frame feature, keyframe feature) lists")
    ap.add_argument("--keyframes", type=int, default=0, help="BoW workloads: keyframes of the synthetic map (0 = 500; euroc_track: 1)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic stereo pairs / images (tiled to the batch)")
    ap.add_argument("--extras", type=int, default=1, help="0 = only the timed headline region (profiling runs): no batch sweep, host-fed, other-config or matcher legs")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the last step's outputs")
    args = ap.parse_args()
    if args.cpu_frames < 0:
        args.cpu_frames = 240 if WORKLOADS[args.workload][3] == "stereo" else 6
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))     # no launcher on the command line: start the ranks ourselves, before anything touches the GPU

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    pkg = ge.load_pkg()
    st = pkg.streams
    # ORBX_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", 0))
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # "nccl" is RCCL on ROCm; no process group for a single process -- unless ORBX_BENCH_FORCE_DIST=1 asks for a one-rank group, so that the very
    # calls an N-GPU run makes (init with device_id, barrier, MAX / SUM all-reduce and the all-gathers on device tensors) execute through RCCL on a one-GPU box
    force_dist = os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"
    rank, world = st.init(backend, device_id=dev if backend == "nccl" else None, force=force_dist)
    assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE={world})"
    pkg.lib()  # fails loudly if liborbx.so is missing: there is no fallback path
    ctx = {"torch": torch, "pkg": pkg, "st": st, "dev": dev, "local": local, "rank": rank, "world": world,
           "dist_dev": dev if backend == "nccl" else None, "backend": backend if (world > 1 or force_dist) else "none"}
    out = run_workload(ctx, args)
    bad = out.get("verified") is not None and not out["verified"]["bit_exact"]

    if rank == 0 and world == 1 and args.extras and args.workload == "stereo1000":
        # ---- the other BASELINE configs, short legs in the same process (outside the headline window)
        others = {}
        for name, steps, batch, distinct, cpuf, nkf in (("tum640", 10, 0, 4, 6, 0), ("euroc_bow", 6, 0, 4, 3, 0),
                                                        ("fhd4000", 6, 0, 2, 3, 0), ("euroc_track", 8, 0, 4, 4, 1)):
            a2 = argparse.Namespace(**vars(args))
            a2.workload, a2.steps, a2.warmup, a2.batch, a2.distinct, a2.cpu_frames, a2.extras, a2.keyframes = name, steps, 2, batch, distinct, cpuf, 0, nkf
            torch.cuda.empty_cache()
            o = run_workload(ctx, a2)
            r = o["roofline"]
            others[name] = {"metric": o["metric"], "value": o["value"], "unit": o["unit"], "steps": steps, "ms_per_step": o["ms_per_step"],
                            "frames_per_step": o["config"]["frames_per_step_per_gpu"], "keypoints_per_image": o["config"]["keypoints_per_image"],
                            "verified": o.get("verified"), "cpu_baseline": o.get("cpu_baseline"),
                            "roofline": {k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "traffic_source") if k in r}}
            for k in ("bow_device_chain", "bow_matches_per_query_frame", "keyframes"):
                if k in o["config"]:
                    others[name][k] = o["config"][k]
            bad = bad or (o.get("verified") is not None and not o["verified"]["bit_exact"])
        out["config"]["other_configs"] = others
        # ---- the three north-star matchers, one pair per call, next to the CPU oracle
        try:
            from oracle import oracle_py
            from tools import matcher_bench
            out["config"]["matchers"] = matcher_bench.measure(pkg, oracle_py, reps=100, cpu_reps=30, device=local)
            bad = bad or out["config"]["matchers"].get("verified") is False
        except Exception as exc:    # measurement leg only
            out["config"]["matchers"] = {"error": str(exc)[:300]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
