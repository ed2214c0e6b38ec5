import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, oracle
from simulate_2048_amd import capi, ops, runtime
from oracle_backend import OracleLibrary
from test_expectimax_depth import mixed_boards
rng=np.random.default_rng(103); b=mixed_boards(rng,40)
g=ops.expectimax(torch.from_numpy(b).cuda(),3, rt=runtime.Runtime(capi.library(),'cuda:0'))
ref=oracle.expectimax_depth(b,3)
q=g['q'].cpu().numpy()
bad=np.argwhere(q!=ref['q'])
print('mismatches',len(bad), bad[:10])
for i,a in bad[:5]:
    print(i,a,repr(q[i,a]),repr(ref['q'][i,a]), q[i,a]-ref['q'][i,a])
lv=ops.expand(torch.from_numpy(b).cuda(), rt=runtime.Runtime(capi.library(),'cuda:0'))
kids=lv['children']
v=torch.zeros(kids.shape[0],dtype=torch.float64,device='cuda')
capi.library().expectimax_depth2_f64(0, torch.cuda.current_stream().cuda_stream, kids.shape[0], kids.data_ptr(), v.data_ptr())
d2=oracle.expectimax_depth2(kids.cpu().numpy())
want=np.where(d2['best']==255,0.0,d2['value'])
print('leaf mismatches', (v.cpu().numpy()!=want).sum(), 'of', len(want))
idx=np.flatnonzero(v.cpu().numpy()!=want)[:5]
for j in idx: print(j, repr(v.cpu().numpy()[j]), repr(want[j]), d2['terms'][j].tolist())
# back-up by hand for the first mismatching (i, a)
off=lv['offsets'].cpu().numpy(); ne=lv['n_empty'].cpu().numpy(); rw=lv['reward'].cpu().numpy(); cv=v.cpu().numpy()
for i,a in bad[:4]:
    at=int(off[i])+2*int(ne[i,:a].astype(int).sum())
    s2=s4=0.0
    for k in range(int(ne[i,a])):
        s2=s2+float(cv[at]); s4=s4+float(cv[at+1]); at+=2
    mix=0.9*s2+0.1*s4
    print('hand', i,a, repr(float(rw[i,a])+mix/float(ne[i,a])), 'gpu', repr(q[i,a]), 'oracle', repr(ref['q'][i,a]), 'ne', ne[i,a], 's2', repr(s2), 's4', repr(s4), 'mix', repr(mix), 'div', repr(mix/float(ne[i,a])))
    import math
    print('   fma-mix', repr(math.fma(0.9,s2,0.1*s4)) if hasattr(math,'fma') else None)
