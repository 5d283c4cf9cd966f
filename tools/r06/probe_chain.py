import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import test_chain_gpu as T
for side, n, lift, mde in ((16, 60, 0.1, 1e-3), (16, 60, 0.1, 2e-4), (16, 60, 0.1, 5e-5)):
    sc = T._drop_scene(side, lift=lift)
    sc.solver_params["max_density_error"] = mde
    w, f, t = T._run({"SALVA_HIP_NO_SPEC_APPLY": "1"}, n, scene=sc)
    print(side, lift, mde, [x[0] for x in t], [x[1] for x in t])
    c = w.counters
    print('  chained', c.chained_passes, 'breaks', c.chain_breaks, 'adopted', c.pregrid_adopted, 'dropped', c.pregrid_dropped)
