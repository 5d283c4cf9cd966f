"""Round 6: the host leaves the step's critical path.

* Chained steps — both solves of a DFSPH step, the kernels between them and the end of the step are enqueued without a wait for
  either solve's outcome; a solve that needs more iterations than the batch enqueued for it shuts a device-side gate, everything
  behind it returns at once, and the host continues the classic way (`World::dfsph_solve`, `StepCtx::gate`).
* The pre-enqueued grid — while the particles' cell box stands still the end of a step enqueues the next step's keys, cell sort and
  tile tables already, into a second set of tables, gated by "the box this step found is the box that work was enqueued for"
  (`World::pre_enqueue_grid`).

Neither may change a bit of the results: the same kernels run on the same data in the same order; only who waits for whom differs.
The reference has no counterpart (its step is synchronous host code, liquid_world.rs:62-158); what is pinned here is that the fast
path equals the plain one — which the parity suites hold against the oracle — including when the host looks at the world, or edits
it, between two steps."""
import os

import numpy as np
import pytest

from parity import DT, GRAVITY, Scene
from salva_amd import scenes

pytestmark = pytest.mark.gpu
R = 0.025
SWITCHES = ("SALVA_HIP_NO_CHAIN", "SALVA_HIP_NO_PREGRID", "SALVA_HIP_NO_SPEC_APPLY")


def _make(env, scene):
    old = {k: os.environ.pop(k, None) for k in SWITCHES}
    os.environ.update(env)
    try:
        return scene.make_hip()  # the switches are read when the world is created
    finally:
        for k in SWITCHES:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]


def _drop_scene(side=24, forces=(("xsph", 0.5, 0.0),), solver="dfsph", lift=0.25, stir=0.0):
    """The bench scene in small: a lattice block falling into an open tank from `lift` metres above its rest position — free fall
    (one iteration per solve, a cell box that stands still for a dozen steps and then moves on by a cell), then the impact (the
    divergence iterations jump from 1 to 20 and beyond)."""
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(side, side, side, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
    fluid[:, 1] += np.float32(lift)
    s.add_fluid(fluid, scenes.random_velocities(len(fluid), stir) if stir else None, 1000.0, forces=list(forces))
    s.add_boundary(shell)
    return s


PLAIN = {"SALVA_HIP_NO_CHAIN": "1", "SALVA_HIP_NO_PREGRID": "1"}


def _run(env, nsteps=64, scene=None, between=None):
    w, (fl,), _ = _make(env, scene or _drop_scene())
    trace = []
    for k in range(nsteps):
        st = w.step(DT, GRAVITY)
        trace.append((st.n_divergence_iters, st.n_pressure_iters, int(st.ncontacts), float(st.divergence_error), float(st.density_error)))
        if between:
            between(k, w, fl, trace)
    return w, fl, trace


def _same_state(wa, fa, wb, fb):
    assert np.array_equal(fa.positions, fb.positions) and np.array_equal(fa.velocities, fb.velocities)
    assert np.array_equal(wa.velocity_changes(fa), wb.velocity_changes(fb))
    assert np.array_equal(wa.densities(fa), wb.densities(fb))
    assert np.array_equal(wa.contact_counts(fa), wb.contact_counts(fb))


def test_chained_steps_and_the_pre_enqueued_grid_change_nothing():
    w0, f0, t0 = _run(PLAIN)
    w1, f1, t1 = _run({})
    w2, f2, t2 = _run({"SALVA_HIP_NO_PREGRID": "1"})
    w3, f3, t3 = _run({"SALVA_HIP_NO_CHAIN": "1"})
    # the scene does what the fast paths are for, and what breaks them
    assert sum(t[0] <= 1 for t in t0) >= 12 and max(t[0] for t in t0) >= 16, [t[0] for t in t0]
    c0, c1, c2, c3 = w0.counters, w1.counters, w2.counters, w3.counters
    assert c0.chained_passes == 0 and c0.chain_breaks == 0 and c0.pregrid_adopted == 0 and c0.pregrid_dropped == 0
    assert c1.chained_passes >= 30 and c1.chain_breaks >= 1, c1          # the chain held in free fall and broke at the impact
    assert c1.pregrid_adopted >= 30 and c1.pregrid_dropped >= 3, c1       # the box stood still, and moved on
    assert c2.chained_passes >= 30 and c2.pregrid_adopted == 0
    assert c3.chained_passes == 0 and c3.pregrid_adopted >= 30
    for w, f, t in ((w1, f1, t1), (w2, f2, t2), (w3, f3, t3)):
        assert t == t0  # iteration counts, contacts AND the errors the solves stopped at, bit for bit
        _same_state(w, f, w0, f0)


def test_chains_that_break_in_either_solve_continue_there():
    """max_density_error tightened until the pressure solve of the compressed block needs 1 ... 7 iterations, a different number
    from step to step, behind a divergence solve that needs 15 ... 18 (speculative applies off, or the chain would not be tried at
    such counts): batches fall short in the first solve of the chain (stage 1) and in the second (stage 2), again and again."""
    def run(env):
        sc = _drop_scene(16, lift=0.1)
        sc.solver_params["max_density_error"] = 5e-5
        return _run(dict(env, SALVA_HIP_NO_SPEC_APPLY="1"), 60, scene=sc)
    w0, f0, t0 = run(PLAIN)
    w1, f1, t1 = run({})
    assert len({t[1] for t in t0}) >= 4 and max(t[1] for t in t0) >= 5, [t[1] for t in t0]  # the pressure iterations do vary
    assert len({t[0] for t in t0[30:]}) >= 3, [t[0] for t in t0]                             # ... and so do the divergence iterations
    assert w1.counters.chain_breaks >= 6 and w1.counters.chained_passes >= 30, w1.counters
    assert t1 == t0
    _same_state(w1, f1, w0, f0)


def test_looking_at_the_world_between_steps_sees_the_finished_step():
    """Contact export and queries read the tables of the step that has run (World::last_ctx) while the NEXT step's grid part may be
    executing on the device already — it works on the other set of tables (world.h GridTabs)."""
    seen = {}

    def look(tag):
        def between(k, w, fl, trace):
            if k % 3 == 0:
                offs, jm, j = w.fluid_contacts(fl)
                hits = w.particles_intersecting_aabb((-0.2, -10.0, -0.2), (0.3, 10.0, 0.1))
                seen.setdefault(tag, []).append((offs.copy(), jm.copy(), j.copy(), sorted((kind, idx) for kind, _, idx in hits)))
        return between

    w0, f0, t0 = _run(PLAIN, 24, between=look("plain"))
    w1, f1, t1 = _run({}, 24, between=look("fast"))
    assert w1.counters.pregrid_adopted >= 6, w1.counters
    assert t1 == t0
    assert len(seen["plain"]) == len(seen["fast"]) == 8
    for a, b in zip(seen["plain"], seen["fast"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3] == b[3]
    _same_state(w1, f1, w0, f0)


def test_editing_the_world_between_steps_drops_the_pre_enqueued_grid():
    """A host edit between two steps (velocities overwritten as heightfield3.rs:40 does, particles added as faucet3.rs:69-104 does,
    a boundary moved) invalidates the grid part that was enqueued for the next step: it has to be dropped, and the step must run
    on the edited world exactly as a world without the fast path does."""
    def edits(k, w, fl, trace):
        if k == 5:
            v = np.array(fl.velocities, dtype=np.float32)
            v[:, 0] += np.float32(0.25)
            fl.velocities = v
        if k == 9:
            top = float(np.asarray(fl.positions)[:, 1].max())
            sheet = scenes.cube_fluid_positions(6, 1, 6, R)
            sheet[:, 1] += np.float32(top + 6 * R - float(sheet[:, 1].min()))
            fl.add_particles(sheet)
        if k == 13:
            p = np.asarray(fl.positions, dtype=np.float32).copy()
            p[:50, 1] += np.float32(0.4)
            fl.positions = p
        if k == 16:
            fl.delete_particle_at_next_timestep(3)
            fl.delete_particle_at_next_timestep(77)

    w0, f0, t0 = _run(PLAIN, 22, between=edits)
    w1, f1, t1 = _run({}, 22, between=edits)
    assert w1.counters.pregrid_adopted >= 4 and w1.counters.pregrid_dropped >= 3, w1.counters
    assert t1 == t0
    _same_state(w1, f1, w0, f0)


def test_the_fast_paths_with_other_forces_and_boundary_reactions():
    """Artificial viscosity + Akinci surface tension in the force list (every force kernel honours the gate) and a tank that wants its
    reaction forces (the applies accumulate into boundary.forces: a gated kernel must not have added anything)."""
    def build(env):
        s = Scene(R, 2.0, "dfsph")
        fluid, shell = scenes.tank(16, 16, 16, R)
        fluid = scenes.jitter(fluid, 0.1 * R, seed=5)
        fluid[:, 1] += np.float32(0.1)
        s.add_fluid(fluid, None, 1000.0, forces=[("artificial", 0.05, 0.02), ("akinci", 1.0, 10.0)])
        s.add_boundary(shell, wants_forces=True)
        w, (fl,), (bd,) = _make(env, s)
        tr, forces = [], []
        for _ in range(44):
            st = w.step(DT, GRAVITY)
            tr.append((st.n_divergence_iters, st.n_pressure_iters, int(st.ncontacts)))
            forces.append(np.array(bd.forces, dtype=np.float32).copy())
            bd.clear_forces()
        return w, fl, tr, forces

    w0, f0, t0, b0 = build(PLAIN)
    w1, f1, t1, b1 = build({})
    assert w1.counters.chained_passes >= 8 and w1.counters.chain_breaks >= 1, w1.counters
    assert t1 == t0
    # (boundary.forces are accumulated with float atomics — the reference's own `apply_force` order is thread-schedule dependent,
    # boundary.rs:62-67 — so two runs of ONE build differ in the last bits: compare to summation order; a kernel that ran where the
    # gate should have stopped it, or twice, would be off by a whole contribution)
    for a, b in zip(b0, b1):
        assert np.abs(a - b).max() <= 2e-5 * max(float(np.abs(a).max()), 1e-6), (np.abs(a - b).max(), np.abs(a).max())
    assert sum(float(np.abs(a).max()) > 0 for a in b0) >= 10  # the tank did feel the impact
    _same_state(w1, f1, w0, f0)


def test_iisph_and_timed_steps_take_the_plain_path():
    """IISPH's last kernels take their pressure buffer by the parity of the iteration count (a host decision) and timed steps record
    events between the phases: neither is chained or pre-enqueued — and both still run."""
    s = _drop_scene(14, forces=(("akinci", 1.0, 10.0),), solver="iisph")
    w, (fl,), _ = _make({}, s)
    for _ in range(6):
        w.step(DT, GRAVITY)
    assert w.counters.chained_passes == 0 and w.counters.chain_breaks == 0
    assert w.counters.pregrid_adopted >= 2  # (the grid part does not depend on the solver)
    w2, (f2,), _ = _make({}, _drop_scene(14))
    w2.counters.enable()
    for _ in range(8):
        w2.step(DT, GRAVITY)
    assert w2.counters.pregrid_adopted == 0 and w2.counters.chained_passes >= 2
    assert w2.counters.step_time > 0
