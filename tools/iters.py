import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from bench import *
side=int(sys.argv[1]) if len(sys.argv)>1 else 100
fluid, shell = build_scene(side)
w, f = make_world(fluid, shell, 0)
out=[]
for k in range(int(sys.argv[2]) if len(sys.argv)>2 else 60):
    t0=time.perf_counter(); st=w.step(DT,GRAVITY); ms=(time.perf_counter()-t0)*1e3
    out.append((k,st.n_divergence_iters,st.n_pressure_iters,round(st.divergence_error,3),round(st.density_error,4),round(ms,2)))
print(out)
