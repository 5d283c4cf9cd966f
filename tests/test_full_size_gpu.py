"""The BASELINE.json configurations at their FULL sizes (10^6 / 2x10^6 particles), on the GPU box.

At these sizes the oracle still finishes a step in seconds on the box's host cores, so the first steps are compared
directly (contact counts exact, densities, positions, iteration counts); on top of that the size-independent
properties: sum of list lengths == reported contacts, run-to-run bitwise determinism, linear momentum of the symmetric
force passes.  bench.py measures config 2; configs 3 and 4 are parity cases only (their timings are in DESIGN.md)."""
import os

import numpy as np
import pytest

from parity import DT, GRAVITY, Scene, max_norm_diff, rel_err
from salva_amd import scenes

pytestmark = pytest.mark.gpu
R = 0.025


def host_threads():
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def check_against_oracle(scene, nsteps, gravity, pos_tol_r, label, noise_floor=False):
    """`noise_floor`: also run the oracle in f64 and accept max(stated tolerance, 2 x |oracle f32 - oracle f64|) — the
    parity protocol of SURVEY.md §8c for passes whose own rounding sensitivity exceeds the stated tolerance (Akinci's
    normalised cohesion / curvature terms move by 1e-3 m/s between f32 and f64 in ONE step at this size)."""
    w, fls, _ = scene.make_hip()
    o = scene.make_oracle(threads=host_threads())
    o64 = scene.make_oracle(threads=host_threads(), f64=True) if noise_floor else None
    for k in range(nsteps):
        st = w.step(DT, gravity)
        so = o.step(DT, gravity)
        if o64 is not None:
            o64.step(DT, gravity)
        # identical positions (step 0) give the identical contact set; afterwards positions agree to rounding only and a
        # pair sitting exactly on d = h may fall on either side
        slack = 0 if k == 0 else max(4, int(1e-6 * so.ncontacts))
        assert abs(int(st.ncontacts) - int(so.ncontacts)) <= slack, f"{label} step {k}: contacts {st.ncontacts} vs {so.ncontacts}"
        assert abs(st.n_pressure_iters - so.n_press_iters) <= 1, f"{label} step {k}: pressure iterations"
        if scene.solver == "dfsph":
            assert abs(st.n_divergence_iters - so.n_div_iters) <= 1, f"{label} step {k}: divergence iterations"
        if k == 0:
            tot = 0
            for f, h in enumerate(fls):
                cnt = w.contact_counts(h)
                assert (cnt == o.contact_counts(f)).all(), f"{label}: fluid-fluid contact counts of fluid {f}"
                assert (w.contact_counts(h, True) == o.contact_counts(f, True)).all()
                tot += int(cnt.sum()) + int(w.contact_counts(h, True).sum())
                assert rel_err(w.densities(h), o.fluid_scalar(f, "densities")) < 1e-5, f"{label}: densities of fluid {f}"
            assert tot <= st.ncontacts  # the remainder are boundary-boundary contacts
    for f, h in enumerate(fls):
        d = max_norm_diff(h.positions, o.fluid_vec(f, "positions")) / R
        vref = max(float(np.abs(o.fluid_vec(f, "velocities")).max()), 2 * R / DT * 1e-2)
        dv = max_norm_diff(h.velocities, o.fluid_vec(f, "velocities")) / vref
        tol_p, tol_v = pos_tol_r * nsteps, 1e-4 * nsteps
        if o64 is not None:
            tol_p = max(tol_p, 2 * max_norm_diff(o.fluid_vec(f, "positions"), o64.fluid_vec(f, "positions")) / R)
            tol_v = max(tol_v, 2 * max_norm_diff(o.fluid_vec(f, "velocities"), o64.fluid_vec(f, "velocities")) / vref)
        assert d < tol_p, f"{label}: positions of fluid {f} differ by {d:.2e} r after {nsteps} steps (tolerance {tol_p:.2e})"
        assert dv < tol_v, f"{label}: velocities of fluid {f} differ by {dv:.2e} v_ref (tolerance {tol_v:.2e})"
    return w, fls


def test_config2_dfsph_xsph_1m_tank():
    """BASELINE config 2 = the bench.py scene: 100^3 block in the open tank, DFSPH + XSPH(0.5, 0)."""
    import bench

    fluid, shell = bench.build_scene(100)
    s = Scene(R, 2.0, "dfsph")
    s.add_fluid(fluid, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    w, (fl,) = check_against_oracle(s, 2, GRAVITY, 1e-4, "config 2")
    # determinism: a second world gives bit-identical output
    w2, (fl2,), _ = s.make_hip()
    for _ in range(2):
        w2.step(DT, GRAVITY)
    assert np.array_equal(fl.positions, fl2.positions) and np.array_equal(w.velocity_changes(fl), w2.velocity_changes(fl2))


def test_config3_iisph_akinci_1m():
    """BASELINE config 3: IISPH defaults + Akinci2013SurfaceTension(1.0, 0.0), 100^3 free block (no boundaries),
    seeded +-0.1 m/s velocities so that the solver has work from the first step; zero gravity -> momentum is conserved."""
    n = 100
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R)
    vel = scenes.random_velocities(len(pos), 0.1)
    s = Scene(R, 2.0, "iisph")
    s.add_fluid(pos, vel, 1000.0, forces=[("akinci", 1.0, 0.0)])
    w, (fl,) = check_against_oracle(s, 2, (0.0, 0.0, 0.0), 1e-4, "config 3", noise_floor=True)
    m = np.float64(fl.particle_mass(0))
    p0 = m * vel.astype(np.float64).sum(axis=0)
    p1 = m * (fl.velocities.astype(np.float64) + w.velocity_changes(fl).astype(np.float64)).sum(axis=0)
    assert np.max(np.abs(p1 - p0)) < 1e-4 * m * np.abs(vel).sum()


def test_config4_two_phase_2m():
    """BASELINE config 4: two fluids of 10^6 particles stacked (rho0 = 1000 below, 500 above), each with XSPH(0.5, 0),
    DFSPH: per-model rest density / mass lookup and cross-model contacts at scale."""
    n = 100
    lower = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=42)
    upper = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=43)
    upper[:, 1] += np.float32(n * 2 * R)
    s = Scene(R, 2.0, "dfsph")
    s.add_fluid(lower, scenes.random_velocities(len(lower), 0.1, seed=1), 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_fluid(upper, scenes.random_velocities(len(upper), 0.1, seed=2), 500.0, forces=[("xsph", 0.5, 0.0)])
    w, fls = check_against_oracle(s, 2, GRAVITY, 1e-4, "config 4")
    assert w.last_stats.nparticles == 2 * n ** 3
    # cross-model contacts exist: particles at the interface have more contacts than their own fluid alone provides
    bottom_of_upper = upper[:, 1] < upper[:, 1].min() + 2 * R
    assert w.contact_counts(fls[1])[bottom_of_upper].mean() > 25


def long_run_against_oracle(scene, nsteps, gravity, label):
    """>= 30 steps at full size, so that the regime the bench headline lives in (the divergence solve pinned at its
    50-iteration cap once the block has hit the floor) is compared with the oracle, not only the first free-fall steps:
    per-step contact counts within a bounded slack, iteration traces within +-1 wherever the oracle's f32 and f64 builds agree with
    each other, final positions / velocities within max(1e-4 r N, 2 x oracle f32-vs-f64)."""
    w, fls, _ = scene.make_hip()
    o = scene.make_oracle(threads=host_threads())
    # the oracle's f64 build runs along: where its iteration counts equal the f32 build's, the device must
    # hit them within +-1; at the steps where the restatement's OWN two precisions disagree — the solve creeping along its
    # tolerance, where the stopping iteration is rounding noise — the allowance is that disagreement, and such steps are bounded
    # (at most a fifth of the run)
    o64 = scene.make_oracle(threads=host_threads(), f64=True)
    trace = []
    loose_steps = 0
    for k in range(nsteps):
        st = w.step(DT, gravity)
        so = o.step(DT, gravity)
        s64 = o64.step(DT, gravity)
        slack = 0 if k == 0 else max(4, int(2e-6 * so.ncontacts) * (k + 1))
        assert abs(int(st.ncontacts) - int(so.ncontacts)) <= slack, f"{label} step {k}: contacts {st.ncontacts} vs {so.ncontacts}"

        def tol_it(a, b, a64):
            # (round 5: the 10 % allowance of rounds 3-4 is gone — what remains is the restatement's own f32-vs-f64 distance at this step)
            return max(1, abs(b - a64))
        tp = tol_it(st.n_pressure_iters, so.n_press_iters, s64.n_press_iters)
        td = tol_it(st.n_divergence_iters, so.n_div_iters, s64.n_div_iters)
        loose_steps += (tp > 1) or (td > 1)
        assert abs(st.n_pressure_iters - so.n_press_iters) <= tp, f"{label} step {k}: pressure iterations {st.n_pressure_iters} vs {so.n_press_iters} (f64 oracle: {s64.n_press_iters})"
        assert abs(st.n_divergence_iters - so.n_div_iters) <= td, f"{label} step {k}: divergence iterations {st.n_divergence_iters} vs {so.n_div_iters} (f64 oracle: {s64.n_div_iters})"
        trace.append((st.n_divergence_iters, st.n_pressure_iters, so.n_div_iters, so.n_press_iters, s64.n_div_iters, s64.n_press_iters))
    for f, h in enumerate(fls):
        po, vo = o.fluid_vec(f, "positions"), o.fluid_vec(f, "velocities")
        d = max_norm_diff(h.positions, po) / R
        vref = max(float(np.abs(vo).max()), 2 * R / DT * 1e-2)
        dv = max_norm_diff(h.velocities, vo) / vref
        tol_p = max(1e-4 * nsteps, 2 * max_norm_diff(po, o64.fluid_vec(f, "positions")) / R)
        tol_v = max(1e-4 * nsteps, 2 * max_norm_diff(vo, o64.fluid_vec(f, "velocities")) / vref)
        assert d < tol_p, f"{label}: positions of fluid {f} differ by {d:.2e} r after {nsteps} steps (tolerance {tol_p:.2e})"
        assert dv < tol_v, f"{label}: velocities of fluid {f} differ by {dv:.2e} v_ref (tolerance {tol_v:.2e})"
    print(label, f"steps with the loose iteration allowance (oracle f32 != f64): {loose_steps} of {nsteps}")
    assert loose_steps <= nsteps // 5, f"{label}: the oracle's two precisions disagree by more than one iteration in {loose_steps} of {nsteps} steps"
    print(label, "iterations (gpu div, gpu press, oracle div, oracle press, oracle-f64 div, oracle-f64 press):", trace)
    return trace


def test_config2_1m_tank_32_steps_into_the_saturated_divergence_regime():
    """The bench scene itself for 32 steps: free fall, impact, and the first steps in which the divergence solve does not
    converge within its 50 iterations any more (the state bench.py's 5 + 50 protocol spends most of its time in)."""
    import bench

    fluid, shell = bench.build_scene(100)
    s = Scene(R, 2.0, "dfsph")
    s.add_fluid(fluid, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    trace = long_run_against_oracle(s, 32, GRAVITY, "config 2, 32 steps")
    assert max(t[0] for t in trace) >= 40, f"the run never reached the saturated regime: {trace}"


def test_config4_two_phase_2m_30_steps():
    n = 100
    lower = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=42)
    upper = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=43)
    upper[:, 1] += np.float32(n * 2 * R)
    s = Scene(R, 2.0, "dfsph")
    s.add_fluid(lower, scenes.random_velocities(len(lower), 0.1, seed=1), 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_fluid(upper, scenes.random_velocities(len(upper), 0.1, seed=2), 500.0, forces=[("xsph", 0.5, 0.0)])
    long_run_against_oracle(s, 30, GRAVITY, "config 4, 30 steps")


def test_config3_iisph_akinci_1m_tank_30_steps():
    """BASELINE config 3 as bench.py --config 3 runs it — the 100^3 block in the open tank, IISPH defaults,
    Akinci2013SurfaceTension(1.0, 10.0) (cohesion + curvature between fluid particles, adhesion to the tank) — for 30 steps
    against the oracle, like configs 2 and 4: per-step contact counts, Jacobi iteration traces, final positions and velocities
    within max(1e-4 r N, 2 x the oracle's own f32-vs-f64 distance)."""
    import bench

    fluids, shell = bench.build_config(3, 100)
    s = Scene(R, 2.0, "iisph")
    s.add_fluid(fluids[0][0], None, 1000.0, forces=[("akinci", 1.0, 10.0)])
    s.add_boundary(shell)
    trace = long_run_against_oracle(s, 30, GRAVITY, "config 3, 30 steps")
    assert max(t[1] for t in trace) >= 2, f"the pressure solve never iterated: {trace}"
