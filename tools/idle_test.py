import sys, importlib, time, numpy as np, torch
sys.path.insert(0, '.')
PKG="demo-learned-point-cloud-compression_amd"
runtime=importlib.import_module(PKG+".runtime"); sparse=importlib.import_module(PKG+".sparse"); wl=importlib.import_module(PKG+".workloads")
rt=runtime.Runtime(0)
with rt:
    frame=wl.room(1_000_000, seed=0)
    pts=torch.from_numpy(frame["points"].astype(np.int32)).cuda()
    coords=torch.cat([torch.zeros((pts.shape[0],1),dtype=torch.int32,device="cuda"),pts],1).contiguous()
    keys=rt.morton_keys(coords); rt.sort_pairs(keys)
    cs1=sparse.CoordSet(rt,keys,1,1); cs2=cs1.down()[0]; cs4=cs2.down()[0]
    g=torch.Generator(device="cpu").manual_seed(0)
    cand2=cs4.up()
    keep=torch.sort(torch.randperm(cand2.n,generator=g)[:cs2.n]).values.to(torch.int32).cuda()
    pruned2=cand2.subset(keep); pn=pruned2.nbr27()
    gw=torch.Generator(device="cuda").manual_seed(1)
    w=(torch.randn((27,32,32),generator=gw,device="cuda")*0.05).contiguous(); b=torch.randn((32,),generator=gw,device="cuda").contiguous()
    hw=torch.randn((32,1),generator=gw,device="cuda").contiguous(); hb=torch.zeros((1,),device="cuda")
    x=torch.randn((8*pruned2.n,32),generator=gw,device="cuda").contiguous()
    rt.conv_prepare(w)
    def one():
        rt.timer_start(); rt.sparse_conv_head_up(x,pn,w,b,True,hw,hb); return rt.timer_stop_ms()
    for _ in range(3): one()
    for gap in (0.0, 0.001, 0.003, 0.010, 0.050):
        ts=[]
        for _ in range(10):
            rt.sync(); time.sleep(gap); ts.append(one())
        print("idle gap %.0f ms -> launch %.3f ms (min %.3f max %.3f)" % (gap*1e3, np.mean(ts), min(ts), max(ts)))
    # busy with light kernels in front
    ts=[]
    small=torch.zeros(1024,device="cuda")
    for _ in range(10):
        rt.sync()
        for _ in range(200): small.add_(1)
        ts.append(one())
    print("200 tiny kernels in front -> %.3f" % np.mean(ts))
rt.close()
