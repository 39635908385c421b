"""Raw PCIe rates of the box (pinned host memory <-> HBM) with the copy engines: what bounds every host-fed figure.  python tools/pcie_bw.py"""
import time, torch, os
dev = torch.device("cuda", 0)
for mb in (16, 64, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device=dev)
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run(fn, reps=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return n * reps / (time.perf_counter() - t0) / 1e9
    def h2d():
        with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    def d2h():
        with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
    def both():
        h2d(); d2h()
    print(f"{mb} MiB: H2D {run(h2d):.1f} GB/s, D2H {run(d2h):.1f} GB/s, both at once {run(both):.1f} GB/s each")
print("cpus", len(os.sched_getaffinity(0)))
