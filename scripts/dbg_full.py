import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle as O
from ilqr_amd import BatchILQR
from tests.util import integrator_x0
B,T,DT=96,99,0.02
goal=[1.0,0.5,0.0,0.0]
om=O.Model("integrator",goal=goal); x0=integrator_x0(B); u0=np.zeros((B,T,2))
g=BatchILQR("integrator",B,T,DT,goal=goal); g.generate_trajectory(x0,u0)
ro=O.batch_solve(om,x0,u0,DT)
c=g.cost(); rel=np.abs(c-ro["cost"])/ro["cost"]
st,it,al=g.status()
print("max rel",rel.max(), "n>1e-6", (rel>1e-6).sum(), "n>1e-9",(rel>1e-9).sum())
for b in np.argsort(-rel)[:6]: print(b, rel[b], c[b], ro["cost"][b], "iters", it[b], ro["iters"][b], "status", st[b], ro["status"][b])
