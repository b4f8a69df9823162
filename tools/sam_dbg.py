import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
import oracle_lib as O, siggen
import t41_sdr_amd as T
L=2048
KW=dict(mode=8, FLoCut=-3000, FHiCut=3000)
nch,nfr=9,8
nco=siggen.nco_grid(nch,seed=22)
I,Q=siggen.make_am_carrier(nch,nfr*L,nco,seed=77)
def run(splits):
    rx=T.RxChain(nch,T.default_params(**KW),NCOFreq=nco)
    x,y=torch.from_numpy(I).cuda(),torch.from_numpy(Q).cuda(); pos=0
    for n in splits:
        rx.ProcessIQData(x[:,pos*L:(pos+n)*L].contiguous(),y[:,pos*L:(pos+n)*L].contiguous()); pos+=n
    return rx.state_records()
a=run([8]); b=run([3,1,4])
d=np.argwhere(a!=b)
print(len(d)); print(d[:40])
for c,i in d[:10]: print(c,i,a[c,i],b[c,i])
