"""Per-phase s_memtime cycle shares of k_fast from the diagnostic build.
Build first: python -c "import __graft_entry__ as g; g.build_diag()"; run on the GPU box."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ORBX_SO"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diag", "liborbx_diag.so")  # built with -DORBX_DIAG
import numpy as np, torch
import __graft_entry__ as ge
from tools import synth
pkg = ge.load_pkg(); L = pkg.lib()
W,H,B = 1241,376,(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
pairs=[synth.stereo_pair(1000+i,W,H)[:2] for i in range(4)]
pitch=1280; host=np.zeros((2*B,H,pitch),np.uint8)
for i in range(B): host[i,:,:W]=pairs[i%4][0]; host[B+i,:,:W]=pairs[i%4][1]
imgs=torch.from_numpy(host).cuda()
ex=pkg.ORBextractor(1000,1.2,8,20,7,device=0,max_size=(W,H),max_batch=2*B)
cap=ex.max_keypoints(W,H)
kps=torch.zeros((2*B,cap,7),device='cuda'); desc=torch.zeros((2*B,cap,32),dtype=torch.uint8,device='cuda'); n=torch.zeros(2*B,dtype=torch.int32,device='cuda')
def step(): ex.extract_batch_device(imgs.data_ptr(),H*pitch,pitch,2*B,W,H,kps.data_ptr(),desc.data_ptr(),cap,n.data_ptr(),None)
for _ in range(3): step()
ex.sync()
out=(C.c_ulonglong*8)()
L.orbx_diag_fast_stamps(out,1)
for _ in range(5): step()
ex.sync()
L.orbx_diag_fast_stamps(out,1)
w=out[7]
names={0:"load+zero",1:"pretest",5:"list",6:"score",2:"barrier",3:"nms",4:"emit"}
tot=sum(out[i] for i in names)
print("k_fast form", ex.debug_fast_form(), "waves",w, "avg cycles/wave", tot/w)
for i in (0,1,5,6,2,3,4): print(f"  {names[i]:10s} {out[i]/w:9.1f} cycles  {100*out[i]/tot:5.1f}%")
if os.environ.get("ORBX_DIAG_FAST_ONLY"): sys.exit(0)

L.orbx_diag_desc_stamps(out,1)
for _ in range(5): step()
ex.sync()
L.orbx_diag_desc_stamps(out,1)
w=out[7]
names=["sync after load","orient","sincos + rowpass","(unused)","sample+store","prologue (levels, counts, kp)","patch load"]
tot=sum(out[i] for i in range(7))
print("k_desc waves",w, "avg cycles/wave", tot/w)
for i in (5,6,0,1,2,3,4): print(f"  {names[i]:30s} {out[i]/w:9.1f} cycles  {100*out[i]/tot:5.1f}%")

L.orbx_diag_tree_stamps(out,1)
for _ in range(5): step()
ex.sync()
L.orbx_diag_tree_stamps(out,1)
w=out[7]
names=["counts + prefix + gather","roots + first classification","sweep: order / scans","sweep: apply","sweep: relabel + classify","final relabel, best per leaf, output"]
tot=sum(out[i] for i in range(6))
print("k_tree level-0 workgroups",w, "avg cycles (s_memtime, 100 MHz) per workgroup", tot/w, "phase-2 sweeps per workgroup", out[6]/w)
for i,nm in enumerate(names): print(f"  {nm:40s} {out[i]/w:9.1f}  {100*out[i]/tot:5.1f}%")

# timeline of the level-0 tree of image 0 (last launch): tag, table size, cycles since the previous entry
tl = (C.c_uint32 * 1024)()
L.orbx_diag_tree_timeline(tl)
step(); ex.sync()
L.orbx_diag_tree_timeline(tl)
tl = np.frombuffer(tl, dtype=np.uint32).reshape(256, 4)
names = {2: "  roots: counters zeroed, barrier", 3: "  roots: points counted into 4-bit fields", 4: "  roots: fields summed over the wave, added", 5: "  roots: barrier",
         6: "  first classification: points classified", 7: "  first classification: fields summed, added",
         0: "gather done (n points)", 1: "roots + first classification done (m)", 10: "sweep: top barrier passed", 11: "  phase-1 order/scan done (new m)",
         12: "  phase-2 order/scan done (new m)", 13: "  apply done", 14: "  relabel + classify done", 20: "output done"}
for r in tl:
    if r[3]: print("%-44s m=%5d  +%6d cycles" % (names.get(int(r[0]), str(r[0])), r[1], r[2]))
