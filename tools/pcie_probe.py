"""PCIe-inclusive rates: what the step costs when the host reads positions + velocities back after every step, or
re-uploads them before every step (the boundary hands over host arrays; bench.py's `value` keeps state resident).
Every mode runs on a FRESH world over the same steps (5 warm-up + 20 timed: the bench protocol) — the scene changes regime while
it runs (0.7 ms free-fall steps, 4.5 ms settled ones), so modes measured one after the other on one world are not comparable
(round 3's figures, 5.57 ms "with download" against 1.37 resident, were taken that way: the 5.57 were mostly later, slower steps)."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from salva_amd import _lib

fl, sh = bench.build_scene(100)
FP = C.POINTER(C.c_float)


def run(mode, steps=20):
    w, f = bench.make_world(fl, sh, 0)
    n = f.num_particles()
    pageable = (np.empty((n, 3), np.float32), np.empty((n, 3), np.float32))

    def after_step():
        if mode in (1, 2):
            p = f.positions; v = f.velocities           # synchronous download (lazy: one D2H of each array)
        if mode == 2:
            f.positions = p; f.velocities = v           # mark dirty: uploaded by the next step
        if mode == 3:                                   # asynchronous, pinned, one step late: the copy overlaps the next step
            w.wait_download()
            w.download_async(f)
        if mode == 4:                                   # asynchronous into caller-owned pageable arrays (pinned ring + memcpy)
            _lib.check(w._L.salva_hip_wait_download(w._h))
            _lib.check(w._L.salva_hip_get_fluid_async(w._h, f._slot, pageable[0].ctypes.data_as(FP), pageable[1].ctypes.data_as(FP)))
        if mode == 5:                                   # asynchronous but waited for at once: pinned DMA, not overlapped
            w.download_async(f)
            w.wait_download()

    # (round 5: the warm-up steps do what the timed steps do — the first asynchronous read-back allocates its pinned arrays and the
    # copy stream, 6-19 ms once, which rounds 3-4 had inside the timed loop: their "1.45-1.55 ms with the download" was 0.2-0.3 ms
    # of that per step)
    for _ in range(5):
        w.step(bench.DT, bench.GRAVITY)
        after_step()
    w.wait_download()
    _lib.check(w._L.salva_hip_wait_download(w._h))
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step(bench.DT, bench.GRAVITY)
        after_step()
    w.wait_download()
    _lib.check(w._L.salva_hip_wait_download(w._h))
    return (time.perf_counter() - t0) / steps * 1e3


for name, mode in (("resident", 0), ("download pos+vel every step (synchronous salva_hip_get_fluid, pageable)", 1),
                   ("download + re-upload every step", 2),
                   ("download pos+vel every step, asynchronous into pinned arrays, one step late", 3),
                   ("download pos+vel every step, asynchronous into pageable arrays (pinned ring + memcpy), one step late", 4),
                   ("download pos+vel every step, pinned arrays, waited for before the next step", 5),
                   ("resident", 0)):
    print(f"{name}: {run(mode):.2f} ms/step", flush=True)
