import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from golden_scenes import SCENES, run_oracle
from parity import DT, GRAVITY
for name in ['dfsph_xsph_block','dfsph_tank']:
    s=SCENES[name][0]()
    w,fl,bo=s.make_hip()
    try:
        st=w.step(DT,GRAVITY); print(name,'ok',st.ncontacts, st.reserved[0], st.reserved[1], st.reserved[2])
    except Exception as e:
        print(name,'ERR',e)
    o=s.make_oracle(); so=o.step(DT,GRAVITY)
    for f,h in enumerate(fl):
        n=w.contact_counts(h); nb=w.contact_counts(h,True); rho=w.densities(h)
        on=o.contact_counts(f); onb=o.contact_counts(f,True)
        print('  rho max', rho.max(), 'alpha max', w.alphas(h).max())
        print('  nff diff', int((n!=on).sum()), 'nfb diff', int((nb!=onb).sum()), 'rho min', rho.min(), 'nan', int(np.isnan(rho).sum()), 'nff min', n.min())
        bad=np.nonzero(n!=on)[0][:5]; print('  bad idx',bad, n[bad], on[bad])
