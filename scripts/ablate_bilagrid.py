import sys; sys.path.insert(0,'/root/repo')
import torch, statistics
from bilateral_driving_amd import _lib as L, harness as Hn
from bilateral_driving_amd.bilagrid import bilagrid_transform
dev='cuda'; H,W=1080,1920
rgb=torch.rand(H,W,3,device=dev,requires_grad=True); alpha=torch.rand(H,W,device=dev); sky=torch.rand(H,W,3,device=dev)
grids=[g[0:1].clone().requires_grad_(True) for g in Hn.make_grids(1,device=dev)]
wt=torch.randn(H,W,3,device=dev)
def run():
    out=bilagrid_transform(rgb,grids,Hn.FACTORS_3,alpha=alpha,sky=sky); (out*wt).sum().backward()
for m in (0,8,0,8):
    L.check(L.lib().bds_set_option(3,m),'opt')
    for _ in range(3): run()
    L.enable_timers(True)
    for _ in range(10): run()
    torch.cuda.synchronize(); t=L.timer_summary(); L.enable_timers(False)
    print('mask',m,{k:round(v[1],4) for k,v in t.items()})
L.check(L.lib().bds_set_option(3,0),'opt')
