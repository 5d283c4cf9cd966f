"""The oracle (oracle/salva_oracle.cpp, f64 build) against an INDEPENDENT second reading of the same Rust
(tests/numpy_reading.py: dense numpy/f64, no grid, no contact lists): every intermediate field of the DFSPH and IISPH steps,
step by step.  Both are f64, so they must agree to summation-order rounding times the conditioning of the solves (1e-7
relative here) — a transcription error in
either reading of dfsph_solver.rs / iisph_solver.rs / the kernel / the dt lag shows up as a difference of order one.
CPU only."""
import numpy as np
import pytest

from numpy_reading import DenseWorld
from oracle import oracle as O
from salva_amd import scenes

R = 0.025
DT = 1.0 / 200.0
G = (0.0, -9.81, 0.0)
R32 = float(np.float32(R))  # the oracle's f64 build takes the radius as the f32 the C ABI hands over, widened
DT32 = float(np.float32(DT))   # likewise dt and gravity: f32 across the C ABI
G32 = tuple(float(np.float32(g)) for g in G)


def make_scene(seed=3, n=7):
    # (compressed by 15 %: densities well above rho0, so that the pressure solves have work too)
    pos = (scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.15 * R, seed) * 0.85).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.8, seed + 1).astype(np.float32)
    # a two-layer floor plate right under the block + one wall, so that boundary terms, boundary volumes and the < 20 contacts
    # rule (particles at the free faces) are all exercised
    d = 2 * R
    lo = pos.min(axis=0)
    gx, gz = np.meshgrid(np.arange(-2, n + 2), np.arange(-2, n + 2), indexing="ij")
    floor = np.stack([lo[0] + gx.ravel() * d, np.full(gx.size, lo[1] - d), lo[2] + gz.ravel() * d], axis=1)
    floor2 = floor.copy(); floor2[:, 1] -= d
    gy, gz2 = np.meshgrid(np.arange(0, n), np.arange(0, n), indexing="ij")
    wall = np.stack([np.full(gy.size, lo[0] - d), lo[1] + gy.ravel() * d, lo[2] + gz2.ravel() * d], axis=1)
    bpos = np.concatenate([floor, floor2, wall]).astype(np.float32)
    return pos, vel, bpos


def rel(a, b, floor=1e-300):
    """max |a - b| relative to the field's largest magnitude (`floor`: the scale below which a field counts as zero — the
    divergences a converged solve leaves behind are residuals 10^9 times smaller than the terms they are the difference of)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


# `caps` = (max_divergence_iter, max_pressure_iter, tolerance).  With the default caps the first divergence solve runs all 50
# iterations (its tolerance is 0 on the first step: inv_dt lags, timestep_manager.rs:78-85) and the residual it ends on is a
# difference of terms 10^5 times larger, so the two f64 readings agree to 1e-6 there instead of 1e-9; the capped runs compare the
# passes themselves at summation-order rounding.
@pytest.mark.parametrize("xsph,caps", [(None, (3, 3, 1e-7)), ((0.5, 0.3), (3, 3, 1e-7)), ((0.5, 0.3), (50, 50, 2e-5))])
def test_dfsph_step_by_step(xsph, caps):
    pos, vel, bpos = make_scene()
    o = O.OracleWorld(R, 2.0, O.DFSPH, f64=True)
    o.set_solver_params(max_divergence_iter=caps[0], max_pressure_iter=caps[1])
    tol = caps[2]
    f = o.add_fluid(pos, 1000.0, vel)
    if xsph:
        o.add_xsph(f, *xsph)
    b = o.add_boundary(bpos, wants_forces=True)
    d = DenseWorld(R32, 2.0, "dfsph")
    d.set_fluid(pos, 1000.0, vel)
    d.set_boundary(bpos)
    d.max_divergence_iter, d.max_pressure_iter = caps[0], caps[1]
    if xsph:
        d.set_xsph(*xsph)
    iters = []
    for k in range(8):
        so = o.step(DT, G)
        d.step(DT32, G32)
        assert int(so.ncontacts) == d.ncontacts, f"step {k}: contacts"
        assert (so.n_div_iters, so.n_press_iters) == (d.n_div, d.n_press), f"step {k}: iterations {(so.n_div_iters, so.n_press_iters)} vs {(d.n_div, d.n_press)}"
        iters.append((d.n_div, d.n_press))
        for name, mine in [("densities", d.rho), ("alphas", d.alpha), ("predicted_densities", d.rho_pred)]:
            assert rel(o.fluid_scalar(f, name), mine) < tol, f"step {k}: {name}"
        assert rel(o.fluid_scalar(f, "divergences"), d.div, floor=1.0) < 10 * tol, f"step {k}: divergences"  # (D rho / Dt of the scene: ~1e4)
        for name, mine in [("velocity_changes", d.dv), ("velocities", d.v), ("positions", d.x)]:
            assert rel(o.fluid_vec(f, name), mine) < tol, f"step {k}: {name}"
        assert rel(o.boundary_volumes(0), d.volb) < 1e-12
        # boundary.forces: what the divergence and pressure applies and the XSPH boundary arm handed to the boundary particles
        # (apply_force), accumulated over the steps so far (nobody clears them here)
        assert np.abs(d.bforce).max() > 0 and rel(o.boundary_vec(b, "forces"), d.bforce) < 10 * tol, f"step {k}: boundary forces"
        assert abs(so.div_error - d.div_err) <= 10 * tol * max(abs(d.div_err), 1e-3) and abs(so.density_error - d.press_err) <= tol * max(abs(d.press_err), 1e-6)
    assert max(i[0] for i in iters) >= 2 and min(i[1] for i in iters) >= 1, f"the solves were meant to iterate: {iters}"
    if caps[0] == 50:
        assert iters[0][0] == 50 and min(i[0] for i in iters[1:]) < 50, f"first-step tolerance 0 (dt lag), then converging: {iters}"


def test_iisph_step_by_step():
    pos, vel, bpos = make_scene(seed=5)
    o = O.OracleWorld(R, 2.0, O.IISPH, f64=True)
    f = o.add_fluid(pos, 1000.0, vel)
    o.add_xsph(f, 0.2, 0.1)
    b = o.add_boundary(bpos, wants_forces=True)
    d = DenseWorld(R32, 2.0, "iisph")
    d.set_fluid(pos, 1000.0, vel)
    d.set_boundary(bpos)
    d.set_xsph(0.2, 0.1)
    iters = []
    for k in range(8):
        so = o.step(DT, G)
        d.step(DT32, G32)
        assert int(so.ncontacts) == d.ncontacts, f"step {k}: contacts"
        assert so.n_press_iters == d.n_press, f"step {k}: Jacobi iterations {so.n_press_iters} vs {d.n_press}"
        iters.append(d.n_press)
        for name, mine in [("densities", d.rho), ("predicted_densities", d.rho_pred), ("aii", d.aii), ("pressures", d.p)]:
            assert rel(o.fluid_scalar(f, name), mine) < 1e-7, f"step {k}: {name}"
        for name, mine in [("dii", d.dii), ("dij_pjl", d.dijpj), ("velocities", d.v), ("positions", d.x)]:
            assert rel(o.fluid_vec(f, name), mine) < 1e-7, f"step {k}: {name}"
        assert np.abs(d.bforce).max() > 0 and rel(o.boundary_vec(b, "forces"), d.bforce) < 1e-6, f"step {k}: boundary forces"
    assert max(iters) >= 3, f"the pressure solve was meant to iterate: {iters}"


KINDS = {"cubic": 0, "poly6": 1, "spiky": 2, "viscosity": 3}


def test_other_kernels_closed_forms():
    """kernel/{poly6,spiky,viscosity}_kernel.rs as the oracle restates them: against the numpy reading's closed forms on a grid
    of radii (incl. 0, h and beyond), the derivative against a central difference of the value, and the unit integral of the two
    normalised ones."""
    from numpy_reading import KERNELS
    lib = O.lib()
    h = 0.1
    rs = np.concatenate([[0.0], np.linspace(1e-4, 1.2 * h, 400), [h]])
    for name, kind in KINDS.items():
        w = np.array([lib.so_kernel_scalar_f64(kind, 0, float(r), h) for r in rs])
        dw = np.array([lib.so_kernel_scalar_f64(kind, 1, float(r), h) for r in rs])
        ref_w, ref_dw = KERNELS[name][0](rs, h), KERNELS[name][1](rs, h)
        assert np.allclose(w, ref_w, rtol=1e-12, atol=1e-14 * np.abs(ref_w).max()), name
        assert np.allclose(dw, ref_dw, rtol=1e-12, atol=1e-14 * np.abs(ref_dw).max()), name  # (atol: the viscosity kernel's terms cancel to 0 at r = h)
        mid = rs[(rs > 0.05 * h) & (rs < 0.95 * h)]
        mid = mid[np.abs(mid - 0.5 * h) > 1e-3 * h]  # (the spline's second derivative jumps at h / 2)
        e = 1e-6 * h
        fd = np.array([(lib.so_kernel_scalar_f64(kind, 0, float(r + e), h) - lib.so_kernel_scalar_f64(kind, 0, float(r - e), h)) / (2 * e) for r in mid])
        an = np.array([lib.so_kernel_scalar_f64(kind, 1, float(r), h) for r in mid])
        assert np.allclose(fd, an, rtol=1e-5, atol=1e-6 * np.abs(an).max()), name
        if name != "viscosity":  # (the viscosity kernel is not a density kernel: its integral diverges at r -> 0 ... it is 1/r there)
            r = np.linspace(0, h, 200001)
            integral = np.trapezoid(4 * np.pi * r * r * KERNELS[name][0](r, h), r)
            assert abs(integral - 1.0) < 1e-6, (name, integral)


@pytest.mark.parametrize("solver,kd,kg", [("dfsph", "poly6", "spiky"), ("iisph", "poly6", "spiky"), ("dfsph", "spiky", "viscosity"), ("dfsph", "cubic", "spiky")])
def test_other_kernels_step_by_step(solver, kd, kg):
    """DFSPHSolver<KernelDensity, KernelGradient> / IISPHSolver<..> with non-default type parameters: oracle f64 vs the numpy reading."""
    pos, vel, bpos = make_scene(seed=7)
    o = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH, f64=True)
    o.set_kernels(KINDS[kd], KINDS[kg])
    o.set_solver_params(max_divergence_iter=3, max_pressure_iter=3)
    f = o.add_fluid(pos, 1000.0, vel)
    o.add_xsph(f, 0.5, 0.3)
    o.add_boundary(bpos)
    d = DenseWorld(R32, 2.0, solver, kd, kg)
    d.set_fluid(pos, 1000.0, vel)
    d.set_boundary(bpos)
    d.max_divergence_iter, d.max_pressure_iter = 3, 3
    d.set_xsph(0.5, 0.3)
    for k in range(4):
        so = o.step(DT, G)
        d.step(DT32, G32)
        assert int(so.ncontacts) == d.ncontacts, f"step {k}: contacts"
        for name, mine in [("densities", d.rho), ("predicted_densities", d.rho_pred)]:
            assert rel(o.fluid_scalar(f, name), mine) < 1e-7, f"step {k}: {name}"
        for name, mine in [("velocities", d.v), ("positions", d.x)]:
            assert rel(o.fluid_vec(f, name), mine) < 1e-7, f"step {k}: {name}"
        assert rel(o.boundary_volumes(0), d.volb) < 1e-12


def f32(x):
    return float(np.float32(x))  # force parameters cross the oracle's C interface as f32


FORCE_CASES = {
    # name: (oracle call, numpy_reading kind, parameters)
    "artificial": (lambda o, f: o.add_artificial_viscosity(f, 0.8, 0.4, alpha=1.0, beta=0.5, speed_of_sound=10.0),
                   ("artificial", f32(0.8), f32(0.4), 1.0, 0.5, 10.0)),
    "akinci2013": (lambda o, f: o.add_akinci2013(f, 0.7, 2.0), ("akinci2013", f32(0.7), 2.0)),
    "he2014": (lambda o, f: o.add_he2014(f, 0.4, 0.6), ("he2014", f32(0.4), f32(0.6))),
    "wcsph": (lambda o, f: o.add_wcsph_tension(f, 50.0, 0.0), ("wcsph", 50.0)),
}


@pytest.mark.parametrize("solver", ["dfsph", "iisph"])
@pytest.mark.parametrize("force", sorted(FORCE_CASES))
def test_nonpressure_forces_step_by_step(solver, force):
    """ArtificialViscosity, Akinci2013SurfaceTension, He2014SurfaceTension and WCSPHSurfaceTension (fluid arm) as dense pair
    expressions (numpy_reading.DenseWorld._force_*) against the oracle's per-contact loops, through both solvers: the force
    enters the accelerations, so velocities, positions, densities and iteration counts of the following steps all carry it.
    The floor and the wall of the scene give the boundary arms (viscosity against the wall, adhesion, the boundary term of the
    colour field) something to act on."""
    pos, vel, bpos = make_scene(seed=9, n=6)
    add, spec = FORCE_CASES[force]
    o = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH, f64=True)
    o.set_solver_params(max_divergence_iter=4, max_pressure_iter=6)
    f = o.add_fluid(pos, 1000.0, vel)
    add(o, f)
    b = o.add_boundary(bpos, wants_forces=True)
    d = DenseWorld(R32, 2.0, solver)
    d.max_divergence_iter, d.max_pressure_iter = 4, 6
    d.set_fluid(pos, 1000.0, vel)
    d.set_boundary(bpos)
    d.add_force(*spec)
    # what the force alone does to the first step: compare with a force-free twin to make sure the case is not vacuous
    twin = DenseWorld(R32, 2.0, solver)
    twin.max_divergence_iter, twin.max_pressure_iter = 4, 6
    twin.set_fluid(pos, 1000.0, vel)
    twin.set_boundary(bpos)
    for k in range(6):
        so = o.step(DT, G)
        d.step(DT32, G32)
        if k == 0:
            twin.step(DT32, G32)
            assert np.abs(d.x - twin.x).max() > 1e-7 * d.h, "the force did nothing"
        assert int(so.ncontacts) == d.ncontacts, f"step {k}: contacts"
        if solver == "dfsph":
            assert (so.n_div_iters, so.n_press_iters) == (d.n_div, d.n_press), f"step {k}: iterations"
        else:
            assert so.n_press_iters == d.n_press, f"step {k}: iterations"
        for name, mine in [("velocities", d.v), ("positions", d.x)]:
            assert rel(o.fluid_vec(f, name), mine) < 1e-7, f"step {k}: {name} ({rel(o.fluid_vec(f, name), mine):.2e})"
        assert rel(o.fluid_scalar(f, "densities"), d.rho) < 1e-7, f"step {k}: densities"
        # boundary.forces (the reactions of the pressure solve + of this force's boundary arm).  Not for ArtificialViscosity: its
        # boundary arm hands each boundary particle the RUNNING sum over the contact list (artificial_viscosity.rs:110-117 applies
        # `boundary_acc`, not this contact's term), which depends on the list order a dense formulation does not have
        if force != "artificial":
            assert rel(o.boundary_vec(b, "forces"), d.bforce) < 1e-6, f"step {k}: boundary forces ({rel(o.boundary_vec(b, 'forces'), d.bforce):.2e})"


@pytest.mark.parametrize("max_iter", [3, 50])
def test_dfsph_viscosity_step_by_step(max_iter):
    """DFSPHViscosity (dfsph_viscosity.rs) on a slightly perturbed sheared block: the 6 x 6 beta matrices with the reference's
    three-column preconditioner, the strain-rate error the loop ends on, its iteration count, and the state that results — two
    readings of an iteration that amplifies differences, agreeing to 1e-12."""
    pos = scenes.jitter(scenes.cube_fluid_positions(7, 7, 7, R), 0.02 * R, seed=42).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.01, seed=12345).astype(np.float32)
    vel[:, 0] += np.float32(2.0) * pos[:, 1]
    o = O.OracleWorld(R, 2.0, O.DFSPH, f64=True)
    o.set_solver_params(max_divergence_iter=4, max_pressure_iter=6)
    f = o.add_fluid(pos, 1000.0, vel)
    o.add_dfsph_viscosity(f, 0.6, 1, max_iter, 0.01)
    d = DenseWorld(R32, 2.0, "dfsph")
    d.max_divergence_iter, d.max_pressure_iter = 4, 6
    d.set_fluid(pos, 1000.0, vel)
    d.add_force("dfsph_viscosity", f32(0.6), 1, max_iter, f32(0.01))
    for k in range(6):
        so = o.step(DT, G)
        d.step(DT32, G32)
        it, err = o.viscosity_stats(f)
        assert it == d.visc_iters and (so.n_div_iters, so.n_press_iters) == (d.n_div, d.n_press), f"step {k}: iterations"
        assert abs(err - d.visc_err) <= 1e-10 * max(d.visc_err, 1e-3), f"step {k}: strain-rate error {err} vs {d.visc_err}"
        assert rel(o.viscosity_betas(f), d.visc_betas) < 1e-11, f"step {k}: betas"
        for name, mine in [("velocities", d.v), ("positions", d.x)]:
            assert rel(o.fluid_vec(f, name), mine) < 1e-11, f"step {k}: {name}"
    assert d.visc_err > 1e-3, "the viscosity loop was meant to have work"


def test_dfsph_viscosity_diverges_in_both_readings():
    """On a lattice jittered by 0.15 r the reference's viscosity loop does not converge but EXPLODES (tests/golden_scenes.py
    scene_dfsph_viscous has the story: the preconditioner touches three of the six columns).  That is a property of the Rust as
    written, so an independent reading must show it too: both readings reach the same astronomically large strain-rate error
    (> 1e60) within two steps, agreeing to six digits."""
    pos, vel, bpos = make_scene(seed=9, n=6)
    o = O.OracleWorld(R, 2.0, O.DFSPH, f64=True)
    o.set_solver_params(max_divergence_iter=4, max_pressure_iter=6)
    f = o.add_fluid(pos, 1000.0, vel)
    o.add_dfsph_viscosity(f, 0.5, 1, 50, 0.01)
    o.add_boundary(bpos)
    d = DenseWorld(R32, 2.0, "dfsph")
    d.max_divergence_iter, d.max_pressure_iter = 4, 6
    d.set_fluid(pos, 1000.0, vel)
    d.set_boundary(bpos)
    d.add_force("dfsph_viscosity", 0.5, 1, 50, f32(0.01))
    worst = 0.0
    for k in range(2):
        o.step(DT, G)
        d.step(DT32, G32)
        it, err = o.viscosity_stats(f)
        assert it == d.visc_iters == 50
        assert abs(err - d.visc_err) <= 1e-6 * d.visc_err, f"step {k}: {err} vs {d.visc_err}"
        assert rel(o.viscosity_betas(f), d.visc_betas) < 1e-10
        worst = max(worst, d.visc_err)
    assert worst > 1e60


@pytest.mark.parametrize("solver", ["dfsph", "iisph"])
def test_two_fluids_and_interaction_groups_step_by_step(solver):
    """BASELINE config 4's ingredients in small: two fluids of different density0 in contact (a light block resting on a heavy
    one), each with its own force list, two boundaries, and InteractionGroups that hide the wall from the light fluid.  What the
    multi-object rules of the Rust decide — whose density0 weighs a boundary particle (`fluid_i.density0`), which fluid arms skip
    foreign contacts (`c.i_model == c.j_model`), the per-fluid error averages and their maximum, which pairs exist at all
    (contacts.rs:277-359) — read twice."""
    n = 5
    d = 2 * R
    lower = (scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.12 * R, 21) * 0.9).astype(np.float32)
    upper = (scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.12 * R, 22) * 0.9).astype(np.float32)
    upper[:, 1] += np.float32(lower[:, 1].max() - upper[:, 1].min() + 0.9 * d)
    v_lower = scenes.random_velocities(len(lower), 0.5, 23).astype(np.float32)
    v_upper = scenes.random_velocities(len(upper), 0.5, 24).astype(np.float32)
    v_upper[:, 1] -= 0.5
    lo = lower.min(axis=0)
    gx, gz = np.meshgrid(np.arange(-2, n + 2), np.arange(-2, n + 2), indexing="ij")
    floor = np.stack([lo[0] + gx.ravel() * d, np.full(gx.size, lo[1] - d), lo[2] + gz.ravel() * d], axis=1).astype(np.float32)
    gy, gz2 = np.meshgrid(np.arange(0, 2 * n + 1), np.arange(0, n), indexing="ij")
    wall = np.stack([np.full(gy.size, lo[0] - d), lo[1] + gy.ravel() * d, lo[2] + gz2.ravel() * d], axis=1).astype(np.float32)
    # groups: lower fluid 0b01, upper fluid 0b10 (both see everything that lets them); the wall only lets group 0b01 in
    G_LOWER, G_UPPER, G_FLOOR, G_WALL = (1, 0xFFFFFFFF), (2, 0xFFFFFFFF), (1, 0xFFFFFFFF), (1, 1)

    o = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH, f64=True)
    o.set_solver_params(max_divergence_iter=5, max_pressure_iter=8)
    f0 = o.add_fluid(lower, 1000.0, v_lower, memberships=G_LOWER[0], filter=G_LOWER[1])
    f1 = o.add_fluid(upper, 500.0, v_upper, memberships=G_UPPER[0], filter=G_UPPER[1])
    o.add_xsph(f0, 0.5, 0.2)
    o.add_akinci2013(f0, 0.6, 1.5)
    o.add_xsph(f1, 0.3, 0.4)
    o.add_he2014(f1, 0.5, 0.3)
    b0 = o.add_boundary(floor, memberships=G_FLOOR[0], filter=G_FLOOR[1], wants_forces=True)
    b1 = o.add_boundary(wall, memberships=G_WALL[0], filter=G_WALL[1], wants_forces=True)

    w = DenseWorld(R32, 2.0, solver)
    w.max_divergence_iter, w.max_pressure_iter = 5, 8
    w.add_fluid(lower, 1000.0, v_lower, *G_LOWER)
    w.add_fluid(upper, 500.0, v_upper, *G_UPPER)
    w.set_xsph(f32(0.5), f32(0.2), fluid=0)
    w.add_force("akinci2013", f32(0.6), 1.5, fluid=0)
    w.set_xsph(f32(0.3), f32(0.4), fluid=1)
    w.add_force("he2014", 0.5, f32(0.3), fluid=1)
    w.add_boundary(floor, *G_FLOOR)
    w.add_boundary(wall, *G_WALL)
    r0, r1 = w.fluid_rows(0), w.fluid_rows(1)
    for k in range(6):
        so = o.step(DT, G)
        w.step(DT32, G32)
        if k == 0:
            # the scene does what it is for: the two fluids touch, the lower one touches the wall, the upper one is within reach
            # of it but has no contact with it
            assert w.ff[r0][:, r1].any() and w.fb[r0][:, w.bmodel == 1].any() and not w.fb[r1][:, w.bmodel == 1].any()
            reach = ((upper[:, None, :].astype(np.float64) - wall[None, :, :]) ** 2).sum(axis=2) <= w.h ** 2
            assert reach.any()
        assert int(so.ncontacts) == w.ncontacts, f"step {k}: contacts {so.ncontacts} vs {w.ncontacts}"
        if solver == "dfsph":
            assert (so.n_div_iters, so.n_press_iters) == (w.n_div, w.n_press), f"step {k}: iterations"
        else:
            assert so.n_press_iters == w.n_press, f"step {k}: iterations"
        for fid, rows in ((f0, r0), (f1, r1)):
            assert rel(o.fluid_scalar(fid, "densities"), w.rho[rows]) < 1e-7, f"step {k}: densities of fluid {fid}"
            for name, mine in [("velocities", w.v), ("positions", w.x)]:
                assert rel(o.fluid_vec(fid, name), mine[rows]) < 1e-7, f"step {k}: {name} of fluid {fid}"
        for bid in (b0, b1):
            assert rel(o.boundary_volumes(bid), w.volb[w.bmodel == bid]) < 1e-12
            assert rel(o.boundary_vec(bid, "forces"), w.bforce[w.bmodel == bid]) < 1e-6, f"step {k}: forces on boundary {bid}"


@pytest.mark.parametrize("solver", ["dfsph", "iisph"])
def test_particles_added_and_deleted_between_steps(solver):
    """The emitter / sink life cycle of examples3d/faucet3.rs:69-104 read twice: `Fluid::add_particles` between steps (the new
    particles enter the solver with zero velocity change and zero IISPH pressure, the old ones keep theirs),
    `delete_particle_at_next_timestep` (gone from the fluid AND from the solver's warm-start buffers at the top of the next step,
    survivors in order) — including a particle that is added and deleted before it ever takes part in a step."""
    pos, vel, bpos = make_scene(seed=13, n=5)
    o = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH, f64=True)
    o.set_solver_params(max_divergence_iter=4, max_pressure_iter=6)
    f = o.add_fluid(pos, 1000.0, vel)
    o.add_xsph(f, 0.4, 0.1)
    o.add_boundary(bpos)
    w = DenseWorld(R32, 2.0, solver)
    w.max_divergence_iter, w.max_pressure_iter = 4, 6
    w.set_fluid(pos, 1000.0, vel)
    w.set_xsph(f32(0.4), f32(0.1))
    w.set_boundary(bpos)
    rng = np.random.default_rng(5)
    top = float(pos[:, 1].max())
    counts = []
    for k in range(9):
        if k in (2, 4, 5):  # a sheet of new particles just above the block, moving down
            d = 2 * R
            gx, gz = np.meshgrid(np.arange(4), np.arange(4), indexing="ij")
            sheet = np.stack([pos[:, 0].min() + gx.ravel() * d, np.full(gx.size, top + (1.0 + 0.3 * k) * d), pos[:, 2].min() + gz.ravel() * d],
                             axis=1).astype(np.float32)
            sv = np.tile(np.array([[0.1, -1.0, 0.0]], np.float32), (len(sheet), 1))
            o.add_particles(f, sheet, sv)
            w.add_particles(0, sheet, sv)
        if k in (3, 5, 6):  # delete a random handful, by index into the current host order
            n_now = o.fluid_len(f)
            assert n_now == int((w.model == 0).sum())
            for i in sorted(set(int(x) for x in rng.integers(0, n_now, size=7))):
                o.delete_particle_at_next_timestep(f, i)
                w.delete_particle_at_next_timestep(0, i)
            if k == 5:  # one of the particles added a moment ago, never stepped
                o.delete_particle_at_next_timestep(f, n_now - 3)
                w.delete_particle_at_next_timestep(0, n_now - 3)
        so = o.step(DT, G)
        w.step(DT32, G32)
        counts.append(o.fluid_len(f))
        assert o.fluid_len(f) == len(w.x), f"step {k}: {o.fluid_len(f)} vs {len(w.x)} particles"
        assert int(so.ncontacts) == w.ncontacts, f"step {k}: contacts"
        for name, mine in [("velocities", w.v), ("positions", w.x)]:
            assert rel(o.fluid_vec(f, name), mine) < 1e-7, f"step {k}: {name} ({rel(o.fluid_vec(f, name), mine):.2e})"
        assert rel(o.fluid_scalar(f, "densities"), w.rho) < 1e-7, f"step {k}: densities"
        if solver == "iisph":
            assert rel(o.fluid_scalar(f, "pressures"), w.p) < 1e-7, f"step {k}: pressures (the next step's warm start)"
    assert len(set(counts)) >= 5, f"the particle count was meant to change: {counts}"


def two_fluid_scene():
    """The scene of test_two_fluids_and_interaction_groups_step_by_step, as (lower, upper, v_lower, v_upper, floor, wall, groups)."""
    n = 5
    d = 2 * R
    lower = (scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.12 * R, 21) * 0.9).astype(np.float32)
    upper = (scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.12 * R, 22) * 0.9).astype(np.float32)
    upper[:, 1] += np.float32(lower[:, 1].max() - upper[:, 1].min() + 0.9 * d)
    v_lower = scenes.random_velocities(len(lower), 0.5, 23).astype(np.float32)
    v_upper = scenes.random_velocities(len(upper), 0.5, 24).astype(np.float32)
    v_upper[:, 1] -= 0.5
    lo = lower.min(axis=0)
    gx, gz = np.meshgrid(np.arange(-2, n + 2), np.arange(-2, n + 2), indexing="ij")
    floor = np.stack([lo[0] + gx.ravel() * d, np.full(gx.size, lo[1] - d), lo[2] + gz.ravel() * d], axis=1).astype(np.float32)
    gy, gz2 = np.meshgrid(np.arange(0, 2 * n + 1), np.arange(0, n), indexing="ij")
    wall = np.stack([np.full(gy.size, lo[0] - d), lo[1] + gy.ravel() * d, lo[2] + gz2.ravel() * d], axis=1).astype(np.float32)
    groups = dict(lower=(1, 0xFFFFFFFF), upper=(2, 0xFFFFFFFF), floor=(1, 0xFFFFFFFF), wall=(1, 1))
    return lower, upper, v_lower, v_upper, floor, wall, groups


@pytest.mark.parametrize("solver", ["dfsph", "iisph"])
def test_f32_build_against_the_reading_with_f32_contacts(solver):
    """The oracle's f32 build — the arithmetic the HIP path is held to — against the numpy reading told to decide contacts in f32
    (`f32_contacts`): the lattice floor and wall hold dozens of pairs at exactly d = h, which f32 and f64 arithmetic put on
    different sides (8771 against 8703 contacts in this scene; the device reports 8771, profiles/r03_peer/numpy_reading_gpu.log).
    With the criterion in f32 the counts agree exactly at every step and the states to f32 noise — the template for comparing
    the device with the reading directly."""
    lower, upper, v_lower, v_upper, floor, wall, g = two_fluid_scene()
    o = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH, f64=False)
    o.set_solver_params(max_divergence_iter=5, max_pressure_iter=8)
    f0 = o.add_fluid(lower, 1000.0, v_lower, memberships=g["lower"][0], filter=g["lower"][1])
    f1 = o.add_fluid(upper, 500.0, v_upper, memberships=g["upper"][0], filter=g["upper"][1])
    o.add_xsph(f0, 0.5, 0.2)
    o.add_akinci2013(f0, 0.6, 1.5)
    o.add_xsph(f1, 0.3, 0.4)
    o.add_he2014(f1, 0.5, 0.3)
    o.add_boundary(floor, memberships=g["floor"][0], filter=g["floor"][1])
    o.add_boundary(wall, memberships=g["wall"][0], filter=g["wall"][1])
    w = DenseWorld(R32, 2.0, solver, f32_contacts=True)
    w.max_divergence_iter, w.max_pressure_iter = 5, 8
    w.add_fluid(lower, 1000.0, v_lower, *g["lower"])
    w.add_fluid(upper, 500.0, v_upper, *g["upper"])
    w.set_xsph(f32(0.5), f32(0.2), fluid=0)
    w.add_force("akinci2013", f32(0.6), 1.5, fluid=0)
    w.set_xsph(f32(0.3), f32(0.4), fluid=1)
    w.add_force("he2014", 0.5, f32(0.3), fluid=1)
    w.add_boundary(floor, *g["floor"])
    w.add_boundary(wall, *g["wall"])
    plain = DenseWorld(R32, 2.0, solver)
    plain.add_fluid(lower, 1000.0, v_lower, *g["lower"]); plain.add_fluid(upper, 500.0, v_upper, *g["upper"])
    plain.add_boundary(floor, *g["floor"]); plain.add_boundary(wall, *g["wall"])
    plain.step(DT32, G32)
    r0, r1 = w.fluid_rows(0), w.fluid_rows(1)
    for k in range(6):
        so = o.step(DT, G)
        w.step(DT32, G32)
        if k == 0:
            assert w.ncontacts != plain.ncontacts, "the scene was meant to hold pairs at exactly d = h"
        assert int(so.ncontacts) == w.ncontacts, f"step {k}: contacts {so.ncontacts} vs {w.ncontacts}"
        for fid, rows in ((f0, r0), (f1, r1)):
            dx = np.abs(o.fluid_vec(fid, "positions") - w.x[rows]).max()
            dv = np.abs(o.fluid_vec(fid, "velocities") - w.v[rows]).max()
            # (measured: 1e-6 h and 4e-6 m/s after six steps — f32 rounding of sums of ~40 terms)
            assert dx < 1e-5 * w.h and dv < 5e-5, f"step {k}: fluid {fid} differs by {dx / w.h:.2e} h, {dv:.2e} m/s"
            assert rel(o.fluid_scalar(fid, "densities"), w.rho[rows]) < 2e-5, f"step {k}: densities of fluid {fid}"
