"""Round 6: two launch classes per pass (device_types.h StepCtx::slot_order).

A stray particle owns a tile; a tile kernel used to give it what it gives a full tile — a 512-thread workgroup and the launch's LDS
request, three per CU — for a list of one entry.  When a step's sparse slots (one slice of own particles, a halo of a few cells'
worth) are many, every pass now runs them in a launch of their own: 64 threads, a few KB of LDS, the run-time-layout instantiation of
the kernel.  The reference pays per occupied cell (geometry/hgrid.rs:22-63) and nothing per stray; here the contract is that a
particle's sums do not know which launch computed them: bit-identical to one launch per pass, for every kernel family."""
import os

import numpy as np
import pytest

from parity import DT, GRAVITY, Scene
from salva_amd import scenes

pytestmark = pytest.mark.gpu
R = 0.025
SWITCHES = ("SALVA_HIP_NO_CLASSES", "SALVA_HIP_CLASSES", "SALVA_HIP_NO_PLANES", "SALVA_HIP_NO_FOLD", "SALVA_HIP_NO_SPLIT", "SALVA_HIP_LIGHT", "SALVA_HIP_SPLIT_S")


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


def _with_strays(block, nstray=60, seed=1):
    """`nstray` particles scattered around the block, most of them alone in their tile, some in pairs that touch, a few beside the
    block (their tiles hold a face of it: sparse by their own count, not by their halo)."""
    rng = np.random.default_rng(seed)
    lo, hi = block.min(axis=0), block.max(axis=0)
    out = []
    for k in range(nstray):
        d = rng.uniform(0.6, 3.0)
        axis = k % 3
        p = rng.uniform(lo, hi)
        p[axis] = (hi[axis] + d) if k % 2 else (lo[axis] - d)
        out.append(p)
        if k % 7 == 0:
            out.append(p + np.array([1.2 * R, 0.3 * R, 0.0]))  # a touching pair
    for k in range(8):  # close to the block: 1.5 cells off a face
        p = rng.uniform(lo, hi)
        p[0] = hi[0] + 0.15
        out.append(p)
    return np.concatenate([block, np.asarray(out, np.float32)]).astype(np.float32)


def _run(env, scene, nsteps):
    w, fls, bds = _make(env, scene)
    tr = []
    for _ in range(nsteps):
        st = w.step(DT, GRAVITY)
        tr.append((st.n_divergence_iters, st.n_pressure_iters, int(st.ncontacts), float(st.density_error)))
    return w, fls, tr


def _same(wa, fa, wb, fb):
    for x, y in zip(fa, fb):
        assert np.array_equal(x.positions, y.positions) and np.array_equal(x.velocities, y.velocities)
        assert np.array_equal(wa.densities(x), wb.densities(y))
        assert np.array_equal(wa.velocity_changes(x), wb.velocity_changes(y))
        assert np.array_equal(wa.contact_counts(x), wb.contact_counts(y)) and np.array_equal(wa.contact_counts(x, True), wb.contact_counts(y, True))


def _scene(solver="dfsph", forces=(("xsph", 0.5, 0.0),), side=14, two=False, boundary=True):
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(side, side, side, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=13)
    pos = _with_strays(fluid)
    vel = scenes.random_velocities(len(pos), 0.3, seed=14)
    if two:
        half = len(pos) // 2
        s.add_fluid(pos[:half], vel[:half], 1000.0, forces=list(forces))
        s.add_fluid(pos[half:], vel[half:], 500.0, forces=list(forces))
    else:
        s.add_fluid(pos, vel, 1000.0, forces=list(forces))
    if boundary:
        s.add_boundary(shell, wants_forces=False)
    return s


CASES = {
    "dfsph+xsph": dict(),
    "dfsph general kernels": dict(env={"SALVA_HIP_NO_PLANES": "1"}),
    "two masses": dict(two=True),
    "iisph+akinci": dict(solver="iisph", forces=(("akinci", 1.0, 10.0),)),
    "artificial+he2014": dict(forces=(("artificial", 0.05, 0.02), ("he2014", 0.5, 0.2))),
    "wcsph tension, no boundary": dict(forces=(("wcsph_tension", 0.3, 0.0),), boundary=False),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_a_launch_of_their_own_changes_nothing_for_the_sparse_slots(case):
    kw = dict(CASES[case])
    env = kw.pop("env", {})
    sc = _scene(**kw)
    w0, f0, t0 = _run(dict(env, SALVA_HIP_NO_CLASSES="1"), sc, 8)
    w1, f1, t1 = _run(dict(env, SALVA_HIP_CLASSES="1"), sc, 8)
    assert t1 == t0
    _same(w1, f1, w0, f0)
    a, b = w0.fluid_contacts(f0[0]), w1.fluid_contacts(f1[0])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_the_classes_switch_themselves_on_when_the_strays_are_many():
    """A thousand particles that have left the scene: every pass takes two launches (SALVA_HIP_TILE_TRACE would say `tiny 1000`), and
    the step computes what the one-launch world computes."""
    s = Scene(R, 2.0, "dfsph")
    fluid, shell = scenes.tank(16, 16, 16, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=3)
    rng = np.random.default_rng(5)
    strays = np.stack([rng.uniform(-30, 30, 1000), rng.uniform(-60, -2, 1000), rng.uniform(-30, 30, 1000)], axis=1).astype(np.float32)
    pos = np.concatenate([fluid, strays])
    s.add_fluid(pos, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    w0, f0, t0 = _run({"SALVA_HIP_NO_CLASSES": "1"}, s, 6)
    w1, f1, t1 = _run({}, s, 6)
    assert t1 == t0
    _same(w1, f1, w0, f0)


def test_the_convergence_test_that_rides_in_an_apply_pass_runs_once_per_pass():
    """A solve that needs many iterations step after step runs its applies speculatively, with the convergence test of the
    iteration in workgroup 0 of the apply (dfsph.hip spec_decide) — of the pass's FIRST launch only: found in round 6, the sparse
    class's launch decided a second time, over its own share of the error partials, and solves stopped early (1 iteration where 3
    were due).  A block driven into the tank's floor, with strays around it."""
    s = Scene(R, 2.0, "dfsph")
    fluid, shell = scenes.tank(14, 14, 14, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=13)
    pos = _with_strays(fluid)
    vel = scenes.random_velocities(len(pos), 0.3, seed=14)
    vel[: len(fluid), 1] -= np.float32(2.5)
    s.add_fluid(pos, vel, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    w0, f0, t0 = _run({"SALVA_HIP_NO_CLASSES": "1"}, s, 12)
    w1, f1, t1 = _run({"SALVA_HIP_CLASSES": "1"}, s, 12)
    its = [t[0] for t in t0]
    assert sum(a >= 4 and b >= 4 for a, b in zip(its, its[1:])) >= 4, its  # (the speculative path follows a step of four iterations or more)
    assert t1 == t0
    _same(w1, f1, w0, f0)


# ---- the light class: slots whose halo fits the three-per-CU layouts, beside slots whose halo does not (StepCtx::nlight)
def _squeezed_column(nx, ny, nz, frac=0.34, solver="dfsph", forces=(("xsph", 0.5, 0.0),), two=False, strays=True):
    """A column whose foot (the lower `frac` of it) is squeezed to 1.6x the rest density: halos beyond 2080 particles there, ordinary
    ones above, a few strays around it."""
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(nx, ny, nz, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=2)
    y0 = float(fluid[:, 1].min())
    cut = int(ny * frac) * 2 * R
    low = fluid[:, 1] < y0 + cut
    fluid[low, 1] = (y0 + (fluid[low, 1] - y0) * np.float32(0.62)).astype(np.float32)
    fluid[~low, 1] -= np.float32(cut * 0.38)
    pos = _with_strays(fluid, nstray=20) if strays else fluid
    if two:
        upper = pos[:, 1] > np.median(pos[:, 1])
        s.add_fluid(np.ascontiguousarray(pos[~upper]), None, 1000.0, forces=list(forces))
        s.add_fluid(np.ascontiguousarray(pos[upper]), None, 500.0, forces=list(forces))
    else:
        s.add_fluid(pos, None, 1000.0, forces=list(forces))
    s.add_boundary(shell)
    return s


LIGHT_CASES = {
    "dfsph+xsph": dict(),
    "dfsph general kernels": dict(env={"SALVA_HIP_NO_PLANES": "1"}),
    "two masses": dict(two=True),
    "iisph+akinci": dict(solver="iisph", forces=(("akinci", 1.0, 10.0),)),
    "artificial+he2014": dict(forces=(("artificial", 0.05, 0.02), ("he2014", 0.5, 0.2))),
}


@pytest.mark.parametrize("case", sorted(LIGHT_CASES))
def test_light_slots_on_the_small_layouts_beside_full_ones_change_nothing(case):
    kw = dict(LIGHT_CASES[case])
    env = dict(kw.pop("env", {}), SALVA_HIP_NO_SPLIT="1")  # (or the over-full tiles of so small a scene would simply be cut)
    sc = _squeezed_column(24, 36, 24, **kw)
    w0, f0, t0 = _run(dict(env, SALVA_HIP_NO_CLASSES="1"), sc, 8)
    w1, f1, t1 = _run(dict(env, SALVA_HIP_CLASSES="1"), sc, 8)
    # (the squeezed foot expands within a few steps, sooner under the stronger forces: then no halo is beyond the small layouts)
    assert w0.counters.light_class_passes == 0 and w1.counters.light_class_passes >= 2, w1.counters
    assert w1.counters.sparse_class_passes >= 8, w1.counters  # (three launches per pass while it lasts: full | light | sparse)
    assert t1 == t0
    _same(w1, f1, w0, f0)
    a, b = w0.fluid_contacts(f0[0]), w1.fluid_contacts(f1[0])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_the_light_class_when_asked_for_runs_where_many_halos_fit_the_small_layouts_and_some_do_not():
    """A third of a million particles whose lower 40 % are over-full — too many tiles to cut them all (World::substep keeps tiles
    whole when more than a fifth are over-full), and hundreds that are not: with SALVA_HIP_LIGHT=1 those run three per CU again, in
    launches of their own.  (Opt-in: on the bench scene the second launch per pass costs more than it brings, world.hip.)"""
    sc = _squeezed_column(64, 80, 64, frac=0.4, strays=False)
    os.environ.pop("SALVA_HIP_SPLIT_S", None)
    w0, f0, t0 = _run({"SALVA_HIP_NO_CLASSES": "1"}, sc, 4)
    w2, f2, t2 = _run({}, sc, 4)
    assert w2.counters.light_class_passes == 0 and t2 == t0
    w1, f1, t1 = _run({"SALVA_HIP_LIGHT": "1"}, sc, 4)
    assert w1.counters.light_class_passes >= 2 and w1.counters.sparse_class_passes == 0, w1.counters  # (until the foot has expanded)
    assert t1 == t0
    _same(w1, f1, w0, f0)


def test_launch_classes_under_chained_steps_and_the_pre_enqueued_grid():
    """The fast paths of tests/test_chain_gpu.py (solves enqueued without a host wait behind device-side gates, the next step's grid part
    enqueued ahead) with two launches per pass: a block in free fall, then on the floor, amid strays — against the plain world (no chain,
    no pre-enqueued grid, one launch per pass), bit for bit.  Every class launch honours the gate; the order table belongs to the step
    whose tables it was built from."""
    s = Scene(R, 2.0, "dfsph")
    fluid, shell = scenes.tank(16, 16, 16, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
    fluid[:, 1] += np.float32(0.12)
    pos = _with_strays(fluid, nstray=40, seed=3)
    s.add_fluid(pos, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    switches = ("SALVA_HIP_NO_CHAIN", "SALVA_HIP_NO_PREGRID")
    old = {k: os.environ.pop(k, None) for k in switches}
    try:
        os.environ.update({"SALVA_HIP_NO_CHAIN": "1", "SALVA_HIP_NO_PREGRID": "1"})
        w0, f0, t0 = _run({"SALVA_HIP_NO_CLASSES": "1"}, s, 40)
        for k in switches:
            os.environ.pop(k, None)
        w1, f1, t1 = _run({"SALVA_HIP_CLASSES": "1"}, s, 40)
    finally:
        for k in switches:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]
    c = w1.counters
    assert c.chained_passes >= 10 and c.chain_breaks >= 1 and c.pregrid_adopted >= 10 and c.sparse_class_passes >= 40, c
    assert max(t[0] for t in t0) >= 8  # (the block did land)
    assert t1 == t0
    _same(w1, f1, w0, f0)
