import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from neilpy_amd import _lib
lib=_lib.load()
other=C.CDLL(os.path.abspath(sys.argv[1]))
rng=np.random.default_rng(0)
ok=True
for dt,sfx in ((torch.float32,"f32"),(torch.float64,"f64")):
    Z=torch.from_numpy(rng.normal(0,1,(300,777)).cumsum(0)).to(dt).cuda()
    fa=getattr(lib,"smrf_disk_filter_"+sfx); fb=getattr(other,"smrf_disk_filter_"+sfx)
    fb.restype=fa.restype; fb.argtypes=fa.argtypes
    st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for r in list(range(1,65)):
        for dil in (0,1):
            a=torch.empty_like(Z); b=torch.empty_like(Z)
            assert fa(C.c_void_p(Z.data_ptr()),C.c_void_p(a.data_ptr()),300,777,777,0,300,0,300,r,dil,0,0,st)==0
            assert fb(C.c_void_p(Z.data_ptr()),C.c_void_p(b.data_ptr()),300,777,777,0,300,0,300,r,dil,0,0,st)==0
            if not torch.equal(a,b): ok=False; print("MISMATCH",sfx,r,dil)
print("variant equals current:",ok)
