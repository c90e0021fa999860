import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/vfm-registration_amd')
import numpy as np, torch
from oracle import oracle as orc
from vfmreg import ops
rng=np.random.default_rng(0)
for n,ext,vs,mode in ((1000000,200.0,1.0,"ds"),(1000000,200.0,1.0,"map"),(1500000,60.0,0.25,"ds"),(1200000,400.0,0.25,"ds")):
    pts=rng.uniform(-ext,ext,(n,3))*[1,1,0.2]
    x=torch.from_numpy(pts).cuda()
    t0=time.perf_counter()
    try:
        if mode=="ds":
            got,info=ops.voxel_robin(x,vs,return_info=True)
            ref=orc.voxel_robin(pts,vs)
        else:
            got,info=ops.voxel_robin(x,vs,20,reserve=False,hash_mul=ops.HASH_MAP,return_info=True)
            ref=orc.voxel_robin(pts,vs,20,False,orc.HASH_MUL_MAP)
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(n,ext,vs,mode,"voxels",info[1],"buckets",info[0],"maxdist",info[2],"equal",bool(np.array_equal(got.cpu().numpy(),ref)),"%.1f ms"%(dt*1e3),flush=True)
    except RuntimeError as e:
        print(n,ext,vs,mode,"refused:",str(e)[:160],flush=True)
