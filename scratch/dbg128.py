import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, oracle_lib
from pinocchio_amd import api, synth
n=128
radii = synth.radii_ladder(5)*(n/128.0); radii[-1]=0
dk = synth.make_density(n, seed=synth.SEED)
x,y = synth.invgrow_table("lcdm"); g=synth.growth_multipliers()
o=oracle_lib.Oracle(n,0); o.set_density(dk); o.set_invgrow(x,y); o.set_growth(g)
f=api.Fmax(n); f.set_density(dk); f.set_invgrow(x,y); f.set_growth(g)
tv=f.sweep(radii); p=f.products()
tvo=o.compute_fmax(radii, do_lpt=False); po=o.products()
d=np.abs(p["Fmax"].astype(float)-po["Fmax"].astype(float))
ulp=np.spacing(np.maximum(np.abs(po["Fmax"]),1).astype(np.float32)).astype(float)
bad=np.argwhere(d>2*ulp)
print("nbad",len(bad), "rmax mismatch", (p["Rmax"]!=po["Rmax"]).sum())
for b in bad[:8]:
    b=tuple(b); print(b, p["Fmax"][b], po["Fmax"][b], p["Rmax"][b], po["Rmax"][b])
# per radius
for ir,rs in enumerate(radii):
    ho=o.second_derivatives(rs); f.compute_second_derivatives(rs)
    hg=[f.second_derivative(i) for i in range(6)]
    for b in bad[:4]:
        b=tuple(b)
        dg=np.array([h[b] for h in hg]); do=np.array([h[b] for h in ho])
        Fg=f.collapse_cells(dg[None,:])[0]; Fg_o=f.collapse_cells(do[None,:])[0]
        Fo=o.inverse_collapse_time(do)[0]; Fo_g=o.inverse_collapse_time(dg)[0]
        print(ir, rs, b, "hess maxdiff", np.abs(dg-do).max(), "F gpu(hg)",Fg,"gpu(ho)",Fg_o,"cpu(ho)",Fo,"cpu(hg)",Fo_g, "eig", o.inverse_collapse_time(do)[1])
