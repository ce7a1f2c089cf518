import sys; sys.path.insert(0,'.')
import numpy as np, neilpy_amd
G=np.load('tests/golden/fda.npz')
for tag in [str(c) for c in G['cases']]:
    A=G[tag+'_in']; got=neilpy_amd.inpaint_nans_by_fda(A); st=neilpy_amd.last_stats['inpaint_fda']
    w=G[tag+'_out']
    d=np.abs(got-w); 
    print(tag,(st['istop'],st['itn']),tuple(int(v) for v in G[tag+'_lsqr']),'maxdiff',float(d.max()) if d.size else 0, 'argmax', np.unravel_index(d.argmax(),d.shape) if d.size else None, flush=True)
