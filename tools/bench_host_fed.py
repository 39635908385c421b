#!/usr/bin/env python3
"""Host-fed throughput of the drop-in stereo path: frames live in HOST memory, every frame is uploaded, processed
(extract L + extract R + ComputeStereoMatches) and its keypoints / descriptors / uRight / depth downloaded.
  --streams S   S extractor handles, one Python thread each (ctypes releases the GIL), each keeps --depth frames in flight
  --pinned      frame buffers in page-locked memory (orbx_pinned_alloc): no staging copy inside submit
Prints one JSON line.  bench.py imports run() for its `config.host_fed` leg."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BF, FX = 386.1448, 718.856


def run(pkg, w=1241, h=376, nfeat=1000, streams=2, depth=None, frames=600, pinned=True, sync_call=False, device=0, seed=1000):
    from tools import synth
    depth = depth or pkg.orbx.pipeline_depth()
    pairs = [synth.stereo_pair(seed + i, w, h)[:2] for i in range(4)]
    if pinned:
        pp = []
        for l, r in pairs:
            a, b = pkg.orbx.pinned_array((h, w)), pkg.orbx.pinned_array((h, w))
            a[:] = l; b[:] = r
            pp.append((a, b))
        pairs = pp
    exs = [pkg.ORBextractor(nfeat, 1.2, 8, 20, 7, device=device, max_size=(w, h), max_batch=2) for _ in range(streams)]
    matched = [0] * streams
    lat = [[] for _ in range(streams)]

    def worker(si):
        ex = exs[si]
        if sync_call:
            for i in range(frames):
                t0 = time.perf_counter()
                out = ex.extract_stereo(pairs[i % 4][0], pairs[i % 4][1], BF, BF / FX)
                lat[si].append(time.perf_counter() - t0)
                matched[si] += int((out[4] >= 0).sum())
            return
        q = []
        for i in range(frames + depth):
            if i >= depth:
                t, t0 = q.pop(0)
                out = ex.extract_stereo_wait(t, copy=False)
                lat[si].append(time.perf_counter() - t0)
                matched[si] += int((out[4] >= 0).sum())
            if i < frames:
                q.append((ex.extract_stereo_submit(pairs[i % 4][0], pairs[i % 4][1], BF, BF / FX), time.perf_counter()))

    for si in range(streams):          # warm-up: geometry tables, workspaces, pipeline slots
        exs[si].extract_stereo(np.asarray(pairs[0][0]), np.asarray(pairs[0][1]), BF, BF / FX)
        t = exs[si].extract_stereo_submit(pairs[0][0], pairs[0][1], BF, BF / FX); exs[si].extract_stereo_wait(t)
    ths = [threading.Thread(target=worker, args=(si,)) for si in range(streams)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    el = time.perf_counter() - t0
    all_lat = np.concatenate([np.array(x) for x in lat]) * 1e6
    return {"frames_per_s": round(streams * frames / el, 1), "streams": streams, "frames_in_flight_per_stream": 1 if sync_call else depth,
            "pinned_frame_buffers": bool(pinned), "frames": streams * frames, "latency_us_median": round(float(np.median(all_lat)), 1),
            "latency_us_p95": round(float(np.percentile(all_lat, 95)), 1), "stereo_matches_per_frame": round(sum(matched) / (streams * frames), 1),
            "h2d_bytes_per_frame": 2 * w * h, "form": "orbx_extract_stereo (synchronous)" if sync_call else "orbx_extract_stereo_submit/_wait"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--nfeat", type=int, default=1000)
    ap.add_argument("--pageable", action="store_true")
    ap.add_argument("--sync", action="store_true", help="the synchronous one-call form, one frame at a time per stream")
    a = ap.parse_args()
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    print(json.dumps(run(pkg, nfeat=a.nfeat, streams=a.streams, depth=a.depth or None, frames=a.frames, pinned=not a.pageable, sync_call=a.sync)))


if __name__ == "__main__":
    main()
