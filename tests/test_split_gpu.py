"""Round 6: over-full tiles are cut along x (device_types.h StepCtx::split_s, tile.h Tile::part).

One launch shape serves every tile of a pass, and it used to follow the FULLEST halo: a few compressed tiles at the bottom of a tank
moved every tile of every pass from three workgroups per CU to two.  A tile whose fluid halo exceeds the three-per-CU layouts is now
two slots (own cells ux 0-1 | 2-3, four of the six halo planes each) or four (one plane of own cells, three halo planes).  The
reference's grid has no tiles (geometry/hgrid.rs:22-63: cost follows the occupied cells); what has to hold here is that the cut
changes nothing a particle sees: the same neighbours in the same order — hence the same sums, bit for bit — whatever the cut."""
import os

import numpy as np
import pytest

from parity import DT, GRAVITY, Scene
from salva_amd import scenes

pytestmark = pytest.mark.gpu
R = 0.025
SWITCHES = ("SALVA_HIP_NO_SPLIT", "SALVA_HIP_SPLIT_S", "SALVA_HIP_FOLD_CELLS", "SALVA_HIP_NO_PLANES")


def _make(env, scene):
    old = {k: os.environ.pop(k, None) for k in SWITCHES}
    os.environ.update(env)
    try:
        return scene.make_hip()
    finally:
        for k in SWITCHES:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]


def _run(env, scene, nsteps, look=True):
    w, fls, bds = _make(env, scene)
    trace, halos, seen = [], [], []
    for k in range(nsteps):
        st = w.step(DT, GRAVITY)
        trace.append((st.n_divergence_iters, st.n_pressure_iters, int(st.ncontacts)))
        halos.append(int(st.reserved[0]))
        if look and k in (0, nsteps - 1):
            seen.append([w.fluid_contacts(f) for f in fls] + [w.fluid_contacts(f, True) for f in fls])
    return w, fls, trace, halos, seen


def _block(side=20, solver="dfsph", forces=(("xsph", 0.5, 0.0),), squeeze=1.0, stir=0.2):
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(side, side, side, R)
    fluid = (scenes.jitter(fluid, 0.1 * R, seed=11) * np.float32(squeeze)).astype(np.float32)
    s.add_fluid(fluid, scenes.random_velocities(len(fluid), stir, seed=4), 1000.0, forces=list(forces))
    s.add_boundary(shell)
    return s


def _two_fluids():
    s = Scene(R, 2.0, "dfsph")
    fluid, shell = scenes.tank(16, 24, 16, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=3)
    mid = 0.5 * (float(fluid[:, 1].min()) + float(fluid[:, 1].max()))
    s.add_fluid(np.ascontiguousarray(fluid[fluid[:, 1] < mid]), None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_fluid(np.ascontiguousarray(fluid[fluid[:, 1] >= mid]), None, 500.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    return s


def _same(wa, fa, wb, fb):
    for x, y in zip(fa, fb):
        assert np.array_equal(x.positions, y.positions) and np.array_equal(x.velocities, y.velocities)
        assert np.array_equal(wa.densities(x), wb.densities(y))
        assert np.array_equal(wa.contact_counts(x), wb.contact_counts(y)) and np.array_equal(wa.contact_counts(x, True), wb.contact_counts(y, True))


def _same_exports(sa, sb):
    assert len(sa) == len(sb) > 0
    for a, b in zip(sa, sb):
        for (o1, m1, j1), (o2, m2, j2) in zip(a, b):
            assert np.array_equal(o1, o2) and np.array_equal(m1, m2) and np.array_equal(j1, j2)  # the same neighbours, in the same ORDER


@pytest.mark.parametrize("limit", [1500, 700, 260])
def test_cut_tiles_compute_the_same_bits(limit):
    """A 20^3 block (halos of ~1770 in the interior): limits that cut nothing but the fullest tiles (1500), every interior tile
    into halves (700: 4 of 6 planes ~ 1180 > 700 -> quarters for most) and everything into single planes (260)."""
    sc = _block()
    w0, f0, t0, h0, s0 = _run({"SALVA_HIP_NO_SPLIT": "1"}, sc, 10)
    w1, f1, t1, h1, s1 = _run({"SALVA_HIP_SPLIT_S": str(limit)}, sc, 10)
    assert max(h0) > limit and max(h1) < max(h0), (max(h0), max(h1))  # there was something to cut, and it was cut
    assert t1 == t0
    _same(w1, f1, w0, f0)
    _same_exports(s1, s0)


def test_cut_tiles_in_a_two_mass_world_with_iisph_and_on_a_folded_grid():
    sc = _two_fluids()
    w0, f0, t0, h0, s0 = _run({"SALVA_HIP_NO_SPLIT": "1"}, sc, 8)
    w1, f1, t1, h1, s1 = _run({"SALVA_HIP_SPLIT_S": "900"}, sc, 8)
    assert max(h1) < max(h0)
    assert t1 == t0
    _same(w1, f1, w0, f0)
    _same_exports(s1, s0)
    sc = _block(16, solver="iisph", forces=(("akinci", 1.0, 10.0),))
    w0, f0, t0, h0, s0 = _run({"SALVA_HIP_NO_SPLIT": "1"}, sc, 8)
    w1, f1, t1, h1, s1 = _run({"SALVA_HIP_SPLIT_S": "800"}, sc, 8)
    assert max(h1) < max(h0) and t1 == t0
    _same(w1, f1, w0, f0)
    # a torus of 8 cells per axis: the halo planes of a cut tile wrap around like everybody else's
    sc = _block(14)
    w0, f0, t0, h0, s0 = _run({"SALVA_HIP_NO_SPLIT": "1", "SALVA_HIP_FOLD_CELLS": "8"}, sc, 6)
    w1, f1, t1, h1, s1 = _run({"SALVA_HIP_SPLIT_S": "500", "SALVA_HIP_FOLD_CELLS": "8"}, sc, 6)
    assert max(h1) < max(h0) and t1 == t0
    _same(w1, f1, w0, f0)
    _same_exports(s1, s0)


def test_the_general_kernels_take_cut_tiles_too():
    sc = _block(16)
    w0, f0, t0, h0, _ = _run({"SALVA_HIP_NO_SPLIT": "1", "SALVA_HIP_NO_PLANES": "1"}, sc, 6, look=False)
    w1, f1, t1, h1, _ = _run({"SALVA_HIP_SPLIT_S": "600", "SALVA_HIP_NO_PLANES": "1"}, sc, 6, look=False)
    assert max(h1) < max(h0) and t1 == t0
    _same(w1, f1, w0, f0)


def test_splitting_switches_itself_on_where_a_few_tiles_are_over_full():
    """A column squeezed to 1.6x the rest density at its foot only (the lower third of the block): a minority of tiles beyond
    2080 halo particles.  The world starts unsplit, sees them in its first step's totals, cuts them from the second step on — the
    fullest halo drops under the bound of the three-per-CU layouts — and computes what a world that never splits computes."""
    s = Scene(R, 2.0, "dfsph")
    fluid, shell = scenes.tank(24, 36, 24, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=2)
    y0 = float(fluid[:, 1].min())
    low = fluid[:, 1] < y0 + 12 * 2 * R
    fluid[low, 1] = (y0 + (fluid[low, 1] - y0) * np.float32(0.62)).astype(np.float32)
    fluid[~low, 1] -= np.float32(12 * 2 * R * 0.38)
    s.add_fluid(fluid, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    w0, f0, t0, h0, _ = _run({"SALVA_HIP_NO_SPLIT": "1"}, s, 6, look=False)
    w1, f1, t1, h1, _ = _run({}, s, 6, look=False)
    assert h0[0] > 2080 and h1[0] == h0[0], (h0, h1)  # the first step finds out
    assert h1[1] <= 2080 < h0[1], (h0, h1)  # ... and in the second step the over-full tiles are cut (the squeezed foot then expands)
    assert all(b <= a for a, b in zip(h0, h1)), (h0, h1)
    assert t1 == t0
    _same(w1, f1, w0, f0)


def test_a_fold_that_piles_the_bulk_onto_itself_is_loosened_and_the_pass_repeated():
    """the fold rule looks at the box, not at where the particles sit.  A 20-cell block on a torus of 8 cells per axis puts
    ~15 images of the bulk into every cell: halos of 25 000 particles that fit no kernel's LDS — before round 6 the step died of
    "a tile's halo does not fit".  The tile totals show it before any solver kernel has run; the fold is loosened (8 -> 32 cells, wider
    than the block) and the pass repeated: the step computes what the unfolded grid computes."""
    s = Scene(R, 2.0, "dfsph")
    fluid = scenes.jitter(scenes.cube_fluid_positions(40, 40, 40, R), 0.1 * R, seed=5)
    s.add_fluid(fluid, scenes.random_velocities(len(fluid), 0.2, seed=6), 1000.0, forces=[("xsph", 0.5, 0.0)])
    old = {k: os.environ.pop(k, None) for k in ("SALVA_HIP_FOLD_CELLS", "SALVA_HIP_NO_FOLD")}
    try:
        os.environ["SALVA_HIP_NO_FOLD"] = "1"
        w0, (f0,), _ = s.make_hip()
        os.environ.pop("SALVA_HIP_NO_FOLD")
        os.environ["SALVA_HIP_FOLD_CELLS"] = "8"
        w1, (f1,), _ = s.make_hip()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    for _ in range(3):
        a, b = w0.step(DT, GRAVITY), w1.step(DT, GRAVITY)
        assert (a.n_divergence_iters, a.n_pressure_iters, a.ncontacts) == (b.n_divergence_iters, b.n_pressure_iters, b.ncontacts)
        assert int(b.reserved[0]) < 4000  # (a torus wider than the block: no tile sees an image of the bulk)
    assert np.array_equal(w0.contact_counts(f0), w1.contact_counts(f1))
    assert np.abs(f0.positions - f1.positions).max() <= 1e-6 * R * 10
