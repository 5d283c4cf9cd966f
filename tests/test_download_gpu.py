"""Asynchronous read-back (salva_hip_get_fluid_async / salva_hip_wait_download): what the renderer of the reference reads every
frame (integrations/rapier/testbed_plugin.rs:361-367) without stalling the step.  The arrays must be bit-identical to the
synchronous salva_hip_get_fluid of the same state, whether the copy overlaps a following step or not, into pinned and into
pageable destinations, for one of several fluids."""
import ctypes as C

import numpy as np
import pytest

from parity import DT, GRAVITY
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, _lib, scenes

pytestmark = pytest.mark.gpu
R = 0.025
FP = C.POINTER(C.c_float)


def _world(n=14):
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=11)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    a = Fluid(pos[: len(pos) // 3], R, 1000.0)
    b = Fluid(pos[len(pos) // 3:], R, 1000.0)
    for f in (a, b):
        f.velocities = scenes.random_velocities(f.num_particles(), 0.3, 5).astype(np.float32)
        f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
        w.add_fluid(f)
    lo = pos.min(axis=0)
    gx, gz = np.meshgrid(np.arange(-2, n + 2), np.arange(-2, n + 2), indexing="ij")
    floor = np.stack([lo[0] + gx.ravel() * 2 * R, np.full(gx.size, lo[1] - 2 * R), lo[2] + gz.ravel() * 2 * R], axis=1).astype(np.float32)
    w.add_boundary(Boundary(floor))
    return w, a, b


def _sync_copy(w, f):
    n = f.num_particles()
    p, v = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    _lib.check(w._L.salva_hip_get_fluid(w._h, f._slot, p.ctypes.data_as(FP), v.ctypes.data_as(FP)))
    return p, v


def test_async_download_equals_the_synchronous_one_and_overlaps_the_next_step():
    w, a, b = _world()
    w2, a2, b2 = _world()
    for k in range(6):
        w.step(DT, GRAVITY)
        w2.step(DT, GRAVITY)
        ref = _sync_copy(w2, b2)          # state k of the twin world, read synchronously
        w.download_async(b)               # state k ...
        if k % 2 == 0:
            w.step(DT, GRAVITY)           # ... copied out while step k + 1 runs
            w2.step(DT, GRAVITY)
        got = w.wait_download()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), f"step {k}: the asynchronous read-back differs"
        if k % 2 == 1:                    # no step since: the pinned arrays ARE the fluid's arrays now
            assert b.positions is got[0] and np.array_equal(b.velocities, ref[1])
    # the other fluid, and the host mirror's lazy path afterwards
    w.download_async(a)
    pa = w.wait_download()
    ra = _sync_copy(w, a)
    assert np.array_equal(pa[0], ra[0]) and np.array_equal(pa[1], ra[1])
    # two consecutive read-backs use different pinned pairs: the first result survives the second request
    w.step(DT, GRAVITY)
    w.download_async(b)
    first = w.wait_download()
    keep = first[0].copy()
    w.step(DT, GRAVITY)
    w.download_async(b)
    second = w.wait_download()
    assert np.array_equal(first[0], keep) and not np.array_equal(second[0], keep)


def test_async_download_into_pageable_and_registered_arrays_and_before_the_first_step():
    w, a, b = _world(10)
    w.sync_to_device()
    n = a.num_particles()
    # before any step: served from the staging arrays (host order already)
    p, v = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    _lib.check(w._L.salva_hip_get_fluid_async(w._h, a._slot, p.ctypes.data_as(FP), v.ctypes.data_as(FP)))
    _lib.check(w._L.salva_hip_wait_download(w._h))
    assert np.array_equal(p, a.positions) and np.array_equal(v, a.velocities)
    w.step(DT, GRAVITY)
    w.step(DT, GRAVITY)
    ref = _sync_copy(w, a)
    # pageable destination: through the library's pinned buffers
    p[:] = np.nan
    _lib.check(w._L.salva_hip_get_fluid_async(w._h, a._slot, p.ctypes.data_as(FP), None))
    _lib.check(w._L.salva_hip_wait_download(w._h))
    assert np.array_equal(p, ref[0])
    # the caller's own array, pinned in place
    q = np.full((n, 3), np.nan, np.float32)
    _lib.check(w._L.salva_hip_host_register(w._h, q.ctypes.data_as(C.c_void_p), q.nbytes))
    try:
        _lib.check(w._L.salva_hip_get_fluid_async(w._h, a._slot, None, q.ctypes.data_as(FP)))
        _lib.check(w._L.salva_hip_wait_download(w._h))
        assert np.array_equal(q, ref[1])
    finally:
        _lib.check(w._L.salva_hip_host_unregister(q.ctypes.data_as(C.c_void_p)))
    # errors: a slot that does not exist; waiting with nothing pending is a no-op
    with pytest.raises(_lib.SalvaHipError):
        _lib.check(w._L.salva_hip_get_fluid_async(w._h, 7, p.ctypes.data_as(FP), None))
    _lib.check(w._L.salva_hip_wait_download(w._h))
