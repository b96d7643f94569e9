import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, torch
import util as U
N,H,W,seed,deg,sig = 20000,128,128,5,1,(0.02,)
case = U.make_case(N,H,W,seed,deg=deg,sigma0=sig)
o,_ = U.run_oracle(case,'f32')
h,_ = U.run_hip(case)
nc_h = h['n_contrib'].view(np.uint32); nc_o=o['n_contrib']
bad = np.argwhere(nc_h!=nc_o)
print('mismatch', len(bad), 'of', H*W)
# per-tile stats
gx=(W+15)//16
from collections import Counter
c=Counter(); 
for y,x in bad: c[(y//16)*gx + x//16]+=1
lens = (o['ranges'][:,1]-o['ranges'][:,0])
print('tiles with mismatch', len(c), 'of', len(lens))
for t,cnt in list(c.items())[:10]:
    print(' tile',t,'len',lens[t],'bad px',cnt)
# sub-tile distribution
sub=Counter()
for y,x in bad: sub[((y%16)//8, (x%16)//8)]+=1
print('subtile dist', dict(sub))
for (y,x) in bad[:8]:
    print(' px',y,x,'hip',nc_h[y,x],'ora',nc_o[y,x],'Thip',h['final_T'][y,x],'Tora',o['final_T'][y,x])
# check oracle: for a bad pixel, list contributing entries and their extents
y,x = bad[0]
t=(y//16)*gx + x//16
r0,r1=o['ranges'][t]
ids=o['point_list'][r0:r1]
co=o['conic_opacity'][ids]; xy=o['xy'][ids]
dx=xy[:,0]-x; dy=xy[:,1]-y
power=-0.5*(co[:,0]*dx*dx+co[:,2]*dy*dy)-co[:,1]*dx*dy
alpha=np.minimum(0.99,co[:,3]*np.exp(power))
contrib=(power<=0)&(alpha>=1/255)
det=co[:,0]*co[:,2]-co[:,1]**2
tt=255*co[:,3]
tau2=2*np.log(np.maximum(tt,1.0000001))/det
hx=np.sqrt(tau2*co[:,2])*1.002+0.02; hy=np.sqrt(tau2*co[:,0])*1.002+0.02
sx0=(x//8)*8; sy0=(y//8)*8
ov=(tt>1)&(xy[:,0]+hx>=sx0)&(xy[:,0]-hx<=sx0+7)&(xy[:,1]+hy>=sy0)&(xy[:,1]-hy<=sy0+7)
print('pixel',y,x,'list len',len(ids),'contrib',contrib.sum(),'contrib but culled',(contrib&~ov).sum())
idx=np.nonzero(contrib&~ov)[0][:5]
for i in idx: print('  entry',i,'xy',xy[i],'co',co[i],'hx',hx[i],'hy',hy[i],'alpha',alpha[i],'det',det[i])
