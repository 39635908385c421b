#!/usr/bin/env python3
"""bench.py — frames/s of ORB extract + match on MI355X (BASELINE.json metric).

Workload (config.workload): KITTI-shape stereo, 1241x376, ORBextractor.nFeatures=1000 per eye,
8 levels, scale 1.2, FAST 20/7.  One FRAME = one stereo pair = extract(left) + extract(right) +
ComputeStereoMatches.  A step = one batch of `--batch` frames (2*batch images in one set of
launches); inputs are synthetic (tools/synth.py) and already resident in HBM when timing starts.

`python bench.py --gpus N --steps K --warmup W`; for N>1 launch under torch.distributed.run (one
rank per GPU, independent camera streams per GPU, RCCL only for the start/stop barrier and the
MAX-reduction of the elapsed time) -> "scaling": "weak".

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (largest share of the HIP-event
time measured live on the launch stream during the timed region) by its ALGORITHMIC bytes
(DESIGN.md section "Kernels"); `cpu_baseline` times the CPU oracle (oracle/, a port: the reference
itself needs OpenCV and cannot be built) on a bounded sample of the same workload on rank 0, N=1.
"""
import argparse
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


def cpu_baseline(frames):
    """the CPU oracle driven like the reference: stereo = 2 threads (L/R), src/Frame.cc:82-85"""
    from oracle import oracle_py
    oL, oR = oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7), oracle_py.Oracle(NFEAT, 1.2, NLEVELS, 20, 7)
    times = []
    for left, right in frames:
        res = {}
        t0 = time.perf_counter()
        tl = threading.Thread(target=lambda: res.__setitem__("l", oL.extract(left)))
        tr = threading.Thread(target=lambda: res.__setitem__("r", oR.extract(right)))
        tl.start(); tr.start(); tl.join(); tr.join()
        oracle_py.stereo_match(oL, oR, res["l"][0], res["l"][1], res["r"][0], res["r"][1], BF, MIN_Z)
        times.append(time.perf_counter() - t0)
    times = np.array(times)
    return {"value": round(float(len(times) / times.sum()), 3), "unit": "frames/s", "cores": 2, "kind": "port",
            "sample": f"{len(times)} stereo frames {W}x{H} @{NFEAT} feats through oracle/liborb_oracle.so "
                      f"(-O3 -march=native), L/R extraction on 2 threads, median {np.median(times) * 1e3:.1f} ms/frame",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0 = the workload's default)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="stereo1000",
                    help="stereo1000 is the BASELINE.json headline metric; the others are the remaining single-GPU configs")
    ap.add_argument("--cpu-frames", type=int, default=240, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--bow-host-path", action="store_true", help="euroc_bow: per-frame host-pointer ComputeBoW + search (round-1 form)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic stereo pairs (tiled to the batch)")
    args = ap.parse_args()
    global W, H, NFEAT
    W, H, NFEAT, kind, def_batch, metric = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = def_batch

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
    rank, world = st.init(backend, device_id=dev if backend == "nccl" else None)   # "nccl" is RCCL on ROCm; no-op for a single process
    assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE={world})"

    pkg.lib()  # fails loudly if liborbx.so is missing: there is no fallback path
    from tools import synth

    B = args.batch
    stream_id = st.owned_streams(rank, world)[0]      # one camera stream per GPU
    seed0 = st.stream_seed(stream_id)
    pitch = (W + 63) // 64 * 64
    stereo = kind == "stereo"
    NI = 2 * B if stereo else B                        # images per step
    host = np.zeros((NI, H, pitch), np.uint8)
    if stereo:
        pairs = [synth.stereo_pair(seed0 + i, W, H)[:2] for i in range(args.distinct)]
        for i in range(B):
            host[i, :, :W] = pairs[i % len(pairs)][0]
            host[B + i, :, :W] = pairs[i % len(pairs)][1]
    else:
        pairs = None
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
    dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    orbx = pkg.orbx

    def extract():
        ex.extract_batch_device(imgs.data_ptr(), H * pitch, pitch, NI, W, H, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), sp)

    bow = None
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
        for _ in range(500):
            perm = rng.permutation(len(base["desc"]))
            dk = synth.flip_bits(rng, base["desc"], 0.08)[perm]
            t = voc.transform(dk, 4)
            kfs.append(dict(desc=dk, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"],
                            flag=(rng.random(len(dk)) < 0.6).astype(np.uint8), angle=base["angle"][perm]))
        bow = {"db": pkg.BowDatabase(kfs, device=local), "voc": voc, "frames": frames_fs, "ms": 0.0, "tms": 0.0, "queries": 0, "matches": 0,
               "fr": pkg.BowFrames(NI, cap, device=local),
               "d_match": torch.zeros((NI, 500, cap), dtype=torch.int32, device=dev), "d_nm": torch.zeros((NI, 500), dtype=torch.int32, device=dev),
               "ev": [], "host_path": args.bow_host_path}

    def step():
        if bow is not None and prof_on[0]:
            ex.profile_enable(True)   # BoW work sits between two extractions on this stream: start a fresh event chain
        extract()
        if stereo:
            orbx.stereo_match_batch_device(ex, 0, ex, B, B, kps.data_ptr(), desc.data_ptr(), nout.data_ptr(),
                                           kps[B:].data_ptr(), desc[B:].data_ptr(), nout[B:].data_ptr(), cap,
                                           BF, MIN_Z, ur.data_ptr(), dp.data_ptr(), sp)
        elif bow is not None and not bow["host_path"]:
            # device-resident chain: Frame::ComputeBoW for the batch, then every keyframe against every frame, all on `sp`
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
            bow["fr"].transform(bow["voc"], kps.data_ptr(), desc.data_ptr(), nout.data_ptr(), B, 4, sp)
            e1.record(stream)
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
    ex.profile_read(reset=True)
    prof_on[0] = True
    ex.profile_enable(True)           # HIP events on the launch stream, around every kernel of the timed steps
    elapsed = st.timed_steps(step, args.steps, local_sync, world, device=dev if backend == "nccl" else None)   # barrier + sync both sides, MAX over ranks
    ex.profile_enable(False)
    prof = ex.profile_read(reset=True)
    ex.sync(sp)

    n_h = nout.cpu().numpy()
    nkp_avg = float(n_h.mean())
    matched = float((ur.cpu().numpy() >= 0).sum() / B) if stereo else 0.0
    value = st.aggregate_rate(B, args.steps, world, elapsed)

    # dominant kernel + roofline (per launch: total stage time / launches; resize = 7 launches per step)
    stage_ms = {k: v[0] for k, v in prof.items()}
    dom = max(stage_ms, key=stage_ms.get)
    launches = max(prof[dom][1], 1)
    avg_ms = stage_ms[dom] / launches
    per_step_launches = launches / args.steps
    bytes_per_launch = algorithmic_bytes(dom, NI, B, nkp_avg) / per_step_launches
    bow_dev = None
    if bow is not None and bow["ev"]:
        t_tr = sum(a.elapsed_time(b_) for a, b_, _ in bow["ev"]); t_se = sum(b_.elapsed_time(c) for _, b_, c in bow["ev"])
        bow_dev = {"transform_ms_per_step": round(t_tr / len(bow["ev"]), 4), "search_ms_per_step": round(t_se / len(bow["ev"]), 4)}
        bow["matches"] = int(bow["d_nm"].sum().item()) * len(bow["ev"])
        if t_se > stage_ms[dom]:   # the search launch dominates: one launch per step, B_bow of SURVEY.md 8d per (keyframe, frame) pair
            dom, avg_ms, per_step_launches = "bow<0>", t_se / len(bow["ev"]), 1.0
            bytes_per_launch = 500 * B * (32 + 4 + 1 + 4) * 2 * nkp_avg + 500 * B * 4 * nkp_avg
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None   # HBM bytes per launch from the rocprofv3 PMC passes (tools/collect_pmc.py), same workload only
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if args.workload == "stereo1000" and os.path.exists(tpath):
        tj = json.load(open(tpath))
        kern = tj.get("kernels", {}).get("k_" + dom)
        if kern and tj.get("images_per_launch"):
            traffic = int(kern["hbm_bytes_per_launch"] * NI / tj["images_per_launch"])
    roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "stage_ms_per_step": {k: round(v / args.steps, 4) for k, v in stage_ms.items()}}

    desc_txt = {"stereo": f"KITTI-shape stereo {W}x{H}, nFeatures={NFEAT}/eye, 8 levels, FAST 20/7: "
                          "extract(L)+extract(R)+ComputeStereoMatches; 1 frame = 1 stereo pair",
                "mono": f"mono {W}x{H}, nFeatures={NFEAT}, 8 levels, FAST 20/7: extract only",
                "bow": f"mono {W}x{H}, nFeatures={NFEAT}: extract + ComputeBoW (synthetic k=10 L=6 vocabulary on device) + SearchByBoW(ratio 0.75, "
                       "checkOri) of every frame against a device-resident 500-keyframe synthetic map; device-resident chain "
                       "(orbx_bow_transform_batch_device -> orbx_bowdb_search_batch_device), matches stay in HBM"}[kind]
    out = {"metric": metric, "value": round(value, 2),
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": desc_txt, "name": args.workload,
                      "frames_per_step_per_gpu": B, "images_per_step_per_gpu": NI,
                      "keypoints_per_image": round(nkp_avg, 1), "stereo_matches_per_frame": round(matched, 1),
                      "parallelism": f"{world} independent camera-stream batches, one per GPU"},
           "roofline": roofline}
    if bow is not None and bow["queries"]:
        if bow_dev:
            out["config"]["bow_device_chain"] = bow_dev
        else:
            out["config"]["bow_ms_per_query_frame_incl_host"] = round(bow["ms"] / bow["queries"], 4)
            out["config"]["of_which_bow_transform_ms"] = round(bow["tms"] / bow["queries"], 4)
        out["config"]["bow_matches_per_query_frame"] = round(bow["matches"] / bow["queries"], 1)
    if rank == 0 and world == 1 and args.cpu_frames > 0 and stereo:
        out["cpu_baseline"] = cpu_baseline([pairs[i % len(pairs)] for i in range(args.cpu_frames)])
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
