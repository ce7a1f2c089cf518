import sys, time, numpy as np
sys.path.insert(0,'.')
import torch, neilpy_amd
s=np.load('tests/golden/samples.npz')
for name in ('samp11','samp24','samp61','samp53'):
    x,y,z=(s[name+'_'+k]/100.0 for k in 'xyz')
    neilpy_amd.smrf(x,y,z,1,18,.15,.5,1.25)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(3): out=neilpy_amd.smrf(x,y,z,1,18,.15,.5,1.25)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/3
    print(name, out[0].shape, "smrf %.1f ms"%(dt*1e3), neilpy_amd.last_stats['inpaint1']['itn'], neilpy_amd.last_stats['inpaint2']['itn'])
