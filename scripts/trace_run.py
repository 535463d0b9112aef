import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
B=int(sys.argv[1]); til=int(sys.argv[2])
buf=torch.zeros(8*(B//(64//til)+4), dtype=torch.int64, device='cuda')
os.environ["OPS_AMD_TRACE_PTR"]=str(buf.data_ptr())
import bench, openpystruct_amd as oa
inp=bench.synth_inputs(B,0,torch.device('cuda'),'trajectory')
out=oa.beam_solve(**inp, tiling=til)
for _ in range(5): oa.beam_solve(**inp, tiling=til, out=out)
torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
oa.beam_solve(**inp, tiling=til, out=out); torch.cuda.synchronize()
t=buf.cpu().numpy().reshape(-1,8); t=t[t[:,0]>0]
t0=t[:,0].min(); ts=(t[:,:4]-t0)/100.0  # 100 MHz -> us
print("waves",len(t)); 
for k,name in enumerate(["start","loaded","computed","end"]):
    print(name, "min %.2f p10 %.2f med %.2f p90 %.2f max %.2f us"%(ts[:,k].min(), np.percentile(ts[:,k],10), np.median(ts[:,k]), np.percentile(ts[:,k],90), ts[:,k].max()))
ex=(t[:,5:8]-t[:,0:1])/100.0
for k,nm in enumerate(["loads issued","table written","all loads landed"]): print("since entry:",nm,"min %.2f med %.2f max %.2f"%(ex[:,k].min(),np.median(ex[:,k]),ex[:,k].max()))
print("since entry: staged+barrier med %.2f"%np.median(ts[:,1]-ts[:,0]))
print("dur load med %.2f compute med %.2f store med %.2f"%(np.median(ts[:,1]-ts[:,0]), np.median(ts[:,2]-ts[:,1]), np.median(ts[:,3]-ts[:,2])))
hw=t[:,4]; simd=(hw>>4)&3; wid=hw&15
import collections
print("wave_id hist", sorted(collections.Counter(wid.tolist()).items()))
for w in sorted(set(wid.tolist())):
    m=wid==w; print("slot",w,"n",m.sum(),"computed med %.2f end med %.2f"%(np.median(ts[m,2]),np.median(ts[m,3])))
