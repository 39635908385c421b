"""Experiment: split a step into half batches on several HIP streams / extractor handles (results
log): overlapping latency-bound kernels of one half with compute-bound kernels of the other.
Run on the GPU box: python tools/bench_multistream.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from tools import synth
pkg = ge.load_pkg(); orbx = pkg.orbx
W,H,NF = 1241,376,1000
BF=386.1448; MINZ=BF/718.856
def setup(B, seed):
    pairs=[synth.stereo_pair(seed+i,W,H)[:2] for i in range(4)]
    pitch=1280; host=np.zeros((2*B,H,pitch),np.uint8)
    for i in range(B): host[i,:,:W]=pairs[i%4][0]; host[B+i,:,:W]=pairs[i%4][1]
    d={}
    d['imgs']=torch.from_numpy(host).cuda(); d['B']=B; d['pitch']=pitch
    d['ex']=pkg.ORBextractor(NF,1.2,8,20,7,device=0,max_size=(W,H),max_batch=2*B)
    cap=d['ex'].max_keypoints(W,H); d['cap']=cap
    d['kps']=torch.zeros((2*B,cap,7),device='cuda'); d['desc']=torch.zeros((2*B,cap,32),dtype=torch.uint8,device='cuda'); d['n']=torch.zeros(2*B,dtype=torch.int32,device='cuda')
    d['ur']=torch.zeros((B,cap),device='cuda'); d['dp']=torch.zeros((B,cap),device='cuda')
    d['stream']=torch.cuda.Stream()
    return d
def step(d):
    B=d['B']; sp=d['stream'].cuda_stream; cap=d['cap']
    d['ex'].extract_batch_device(d['imgs'].data_ptr(),H*d['pitch'],d['pitch'],2*B,W,H,d['kps'].data_ptr(),d['desc'].data_ptr(),cap,d['n'].data_ptr(),sp)
    orbx.stereo_match_batch_device(d['ex'],0,d['ex'],B,B,d['kps'].data_ptr(),d['desc'].data_ptr(),d['n'].data_ptr(),d['kps'][B:].data_ptr(),d['desc'][B:].data_ptr(),d['n'][B:].data_ptr(),cap,BF,MINZ,d['ur'].data_ptr(),d['dp'].data_ptr(),sp,row_table=orbx.ROWTAB_OF_EXTRACTION)
for nstreams, B in ((1,256),(2,128),(4,64),(4,128),(8,64)):
    ds=[setup(B,1000+10*i) for i in range(nstreams)]
    for _ in range(3):
        for d in ds: step(d)
    torch.cuda.synchronize()
    K=30; t0=time.perf_counter()
    for _ in range(K):
        for d in ds: step(d)
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    print(nstreams,B,"frames/s",round(nstreams*B*K/el,1), "ms per", nstreams*B, "frames:", round(el/K*1e3,3))
    del ds
