import sys, time, subprocess; sys.path.insert(0,'/root/repo')
import numpy as np
import bench
from bench import *
fluid, shell = build_scene(100)
w, f = make_world(fluid, shell, 0)
for _ in range(5): w.step(DT, GRAVITY)
def clocks():
    try:
        out=subprocess.run(['rocm-smi','--showclocks'],capture_output=True,text=True).stdout
        return [l.split(':')[-1].strip() for l in out.splitlines() if 'sclk' in l or 'mclk' in l]
    except Exception as e: return str(e)
print('idle clocks', clocks())
for k in range(4):
    print('pred_density us', w.time_pred_density(50), w.time_pred_density(200))
t0=time.perf_counter()
for _ in range(20): w.step(DT,GRAVITY)
print('ms/step', (time.perf_counter()-t0)/20*1e3, clocks())
