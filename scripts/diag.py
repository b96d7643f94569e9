import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, torch
import util as U
for (N,H,W,seed,deg,sig) in [(20000,128,128,5,1,(0.02,)), (2000,64,64,7,0,(0.05,)), (10000,256,256,0,3,(0.0052,0.00065))]:
    case = U.make_case(N,H,W,seed,deg=deg,sigma0=sig)
    grads = U.rand_grads(case)
    o, og = U.run_oracle(case,'f32',grads)
    o64, og64 = U.run_oracle(case,'f64',grads)
    h, hg = U.run_hip(case, grads)
    print('case',N,H,W,'D',o['num_rendered'],'max list', (o['ranges'][:,1]-o['ranges'][:,0]).max(), 'max ncontrib', o['n_contrib'].max())
    for k in ('color','depth','alpha','final_T'):
        r64 = o64[k].reshape(h[k].shape)
        d_h = np.abs(h[k]-r64); d_o = np.abs(o[k].reshape(h[k].shape)-r64); d_ho=np.abs(h[k]-o[k].reshape(h[k].shape))
        print(f'  {k}: hip-f64 max {d_h.max():.3e} mean {d_h.mean():.3e} | f32-f64 max {d_o.max():.3e} mean {d_o.mean():.3e} | hip-f32 max {d_ho.max():.3e} frac>1e-4 {(d_ho>1e-4).mean():.4f} >1e-5 {(d_ho>1e-5).mean():.4f}')
    nc_h = h['n_contrib'].view(np.uint32); 
    print('  n_contrib mismatch hip-f32', (nc_h!=o['n_contrib']).mean(), 'f32-f64', (o['n_contrib']!=o64['n_contrib']).mean(), 'hip-f64', (nc_h!=o64['n_contrib']).mean())
    for k in ('means3D','means2D','shs','opacities','scales','rotations'):
        ref = og64[k].reshape(hg[k].shape)
        print(f'  grad {k}: hip rel {U.rel_inf(hg[k],ref):.3e}  f32 rel {U.rel_inf(og[k].reshape(hg[k].shape),ref):.3e}  max|ref| {np.abs(ref).max():.3e}')
