"""GPU parity tests: the HIP path (through the C ABI, via the Python mirror of the salva3d API) against the CPU
oracle on identical seeded inputs, and against the committed golden fixtures.

Tolerances (f32; the reference's own summation order is unspecified):
  contact counts ................ exact (the d^2 <= h^2 test is evaluated with the reference's rounding)
  densities, boundary volumes ... rel 1e-5      alphas ... rel 1e-4
  positions after N steps ....... 1e-4 * particle_radius * N   (the oracle's own f32/f64/order noise is ~1e-6 r per step,
                                  tests/test_oracle.py::test_noise_floor_f64_order_threads)
  velocities / velocity changes . 1e-4 * N * v_ref with v_ref = max(|v|, 2r/dt * 1e-2)
  iteration counts .............. equal, +-1 tolerated where the error sits on the threshold
"""
import os

import numpy as np
import pytest

from golden_scenes import R, SCENES, run_oracle
from parity import DT, GRAVITY, Scene, max_norm_diff, rel_err
from salva_amd import scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_hip(scene: Scene, nsteps: int):
    """Mirror of golden_scenes.run_oracle on the HIP path."""
    w, fl, bo = scene.make_hip()
    out = {}
    iters = []
    for step in range(nsteps):
        st = w.step(DT, GRAVITY)
        iters.append([st.n_divergence_iters, st.n_pressure_iters, st.ncontacts])
        for f, h in enumerate(fl):
            for k, force in enumerate(h.nonpressure_forces):
                if hasattr(force, "viscosity_coefficient"):
                    out.setdefault(f"visc_iters_{f}_{k}", []).append(force.num_iterations)
        if step == 0:
            for f, h in enumerate(fl):
                out[f"s1_density_{f}"] = w.densities(h)
                out[f"s1_alpha_{f}"] = w.alphas(h)
                out[f"s1_nff_{f}"] = w.contact_counts(h, False)
                out[f"s1_nfb_{f}"] = w.contact_counts(h, True)
                out[f"s1_pos_{f}"] = h.positions.copy()
                out[f"s1_vel_{f}"] = h.velocities.copy()
                out[f"s1_dv_{f}"] = w.velocity_changes(h)
            for b, h in enumerate(bo):
                out[f"s1_bvol_{b}"] = h.volumes
    for f, h in enumerate(fl):
        out[f"pos_{f}"] = h.positions.copy()
        out[f"vel_{f}"] = h.velocities.copy()
        out[f"dv_{f}"] = w.velocity_changes(h)
        out[f"density_{f}"] = w.densities(h)
        if scene.solver == "iisph":
            out[f"pressure_{f}"] = w.pressures(h)
    for b, h in enumerate(bo):
        if h.wants_forces:
            out[f"bforce_{b}"] = h.forces
    out["iters"] = np.asarray(iters, dtype=np.int64)
    for k in [k for k in out if k.startswith("visc_iters_")]:
        out[k] = np.asarray(out[k], dtype=np.int64)
    return out


def compare(got, ref, scene, nsteps, label):
    nf = len(scene.fluids)
    ts = scene.tol_scale  # 1 except for ill-conditioned passes (parity.Scene)
    vref = max(2 * R / DT * 1e-2, max(float(np.abs(ref[f"vel_{f}"]).max()) for f in range(nf)))
    for f in range(nf):
        assert (got[f"s1_nff_{f}"] == ref[f"s1_nff_{f}"]).all(), f"{label}: fluid-fluid contact counts differ"
        assert (got[f"s1_nfb_{f}"] == ref[f"s1_nfb_{f}"]).all(), f"{label}: fluid-boundary contact counts differ"
        assert rel_err(got[f"s1_density_{f}"], ref[f"s1_density_{f}"]) < 1e-5, label
        if scene.solver == "dfsph":
            a, b = got[f"s1_alpha_{f}"], ref[f"s1_alpha_{f}"]
            assert rel_err(a, b, floor=float(np.abs(b).max()) * 1e-3) < 1e-4, label
        assert max_norm_diff(got[f"s1_pos_{f}"], ref[f"s1_pos_{f}"]) < 1e-4 * R * ts, label
        assert max_norm_diff(got[f"s1_vel_{f}"], ref[f"s1_vel_{f}"]) < 1e-4 * vref * ts, label
        assert max_norm_diff(got[f"s1_dv_{f}"], ref[f"s1_dv_{f}"]) < 1e-4 * vref * ts, label
        assert max_norm_diff(got[f"pos_{f}"], ref[f"pos_{f}"]) < 1e-4 * R * nsteps * ts, label
        assert max_norm_diff(got[f"vel_{f}"], ref[f"vel_{f}"]) < 1e-4 * vref * nsteps * ts, label
        assert max_norm_diff(got[f"dv_{f}"], ref[f"dv_{f}"]) < 1e-4 * vref * nsteps * ts, label
        assert rel_err(got[f"density_{f}"], ref[f"density_{f}"]) < 1e-4 * ts, label
        if scene.solver == "iisph":
            pr = ref[f"pressure_{f}"]
            assert np.max(np.abs(got[f"pressure_{f}"] - pr)) < 1e-3 * max(1.0, float(pr.max())), label
    for b, bd in enumerate(scene.boundaries):
        assert rel_err(got[f"s1_bvol_{b}"], ref[f"s1_bvol_{b}"]) < 1e-5, label
        if bd["wants_forces"]:
            fr = ref[f"bforce_{b}"]
            scale = max(float(np.abs(fr).max()), 1e-6)
            assert np.max(np.abs(got[f"bforce_{b}"] - fr)) < 2e-3 * scale, label
    for k in [k for k in ref if k.startswith("visc_iters_")]:
        assert (np.abs(got[k] - ref[k]) <= 1).all(), f"{label}: viscosity iterations {got[k].tolist()} vs {ref[k].tolist()}"
    gi, ri = got["iters"], ref["iters"]
    assert (gi[:, 2] == ri[:, 2]).all(), f"{label}: ncontacts differ {gi[:, 2]} vs {ri[:, 2]}"
    assert (np.abs(gi[:, :2] - ri[:, :2]) <= 1).all(), f"{label}: iteration counts {gi[:, :2].tolist()} vs {ri[:, :2].tolist()}"


@pytest.mark.parametrize("name", sorted(SCENES))
def test_against_live_oracle(name):
    builder, nsteps = SCENES[name]
    scene = builder()
    compare(run_hip(scene, nsteps), run_oracle(scene, nsteps), scene, nsteps, f"{name} vs oracle")


@pytest.mark.parametrize("name", sorted(SCENES))
def test_against_golden_fixture(name):
    builder, nsteps = SCENES[name]
    scene = builder()
    ref = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    compare(run_hip(scene, nsteps), ref, scene, nsteps, f"{name} vs golden")


def test_longer_trajectory_dam_break():
    """60 steps of a 12x16x12 block collapsing in a tank (basic3-like: DFSPH + ArtificialViscosity)."""
    s = Scene(R, 2.0, "dfsph")
    fluid, shell = scenes.tank(12, 16, 12, R, wall_cells=10)
    s.add_fluid(scenes.jitter(fluid, 0.05 * R, seed=42), None, 1000.0, forces=[("artificial", 1.0, 0.0)])
    s.add_boundary(shell)
    n = 60
    w, (fl,), _ = s.make_hip()
    o = s.make_oracle(threads=4)
    worst = 0.0
    for k in range(n):
        st = w.step(DT, GRAVITY)
        so = o.step(DT, GRAVITY)
        # identical positions give the identical contact set; once the two trajectories differ by rounding, a pair sitting
        # on d = h may fall on either side: a bounded slack, never a waiver
        assert abs(int(st.ncontacts) - int(so.ncontacts)) <= (0 if k == 0 else max(4, int(2e-5 * so.ncontacts) * (k + 1))), (k, st.ncontacts, so.ncontacts)
        assert abs(st.n_pressure_iters - so.n_press_iters) <= 1 and abs(st.n_divergence_iters - so.n_div_iters) <= 2, k
        if k in (0, 9, n - 1):
            d = max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) / R
            worst = max(worst, d / (k + 1))
    # chaotic growth is slow on this time scale: stay within 1e-3 r per step
    assert worst < 1e-3, worst
    assert fl.positions[:, 1].min() > shell[:, 1].min() - R  # nothing fell through the floor


def _basic3_worlds():
    """BASELINE config[0]: the literal scene of /root/reference/examples3d/basic3.rs on both implementations, colliders coupled
    through ColliderSampling::StaticSampling exactly as the example registers them (fixed parent body: no forces)."""
    from oracle import oracle as O
    from salva_amd import ArtificialViscosity, Boundary, DFSPHSolver, Fluid, LiquidWorld
    from salva_amd.coupling import ColliderCouplingSet, RigidBody, StaticSampling

    r = 0.05
    fluid, colliders = scenes.basic3(15, r)
    w = LiquidWorld(DFSPHSolver(), r, 2.0)
    fl = Fluid(fluid, r, 1000.0)
    fl.nonpressure_forces.append(ArtificialViscosity(1.0, 0.0))
    w.add_fluid(fl)
    coupling = ColliderCouplingSet()
    for k, (pts, t, q) in enumerate(colliders):
        b = w.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
        coupling.register_coupling(b, k, RigidBody(translation=t, rotation=q, dynamic=False), StaticSampling(pts))

    def make_oracle(f64=False):
        o = O.OracleWorld(r, 2.0, O.DFSPH, f64=f64, threads=4)
        f = o.add_fluid(fluid, 1000.0)
        o.add_artificial_viscosity(f, 1.0, 0.0)
        for pts, t, q in colliders:
            b = o.add_boundary(np.zeros((0, 3), np.float32))
            o.set_boundary_sampling(b, pts)
        return o

    def pose_oracle(o):
        for b, (pts, t, q) in enumerate(colliders):
            o.update_boundary_pose(b, t, q, (0, 0, 0), (0, 0, 0), t, True, False)

    return r, fluid, colliders, w, fl, coupling, make_oracle, pose_oracle


def test_basic3_literal_scene_200_steps():
    """examples3d/basic3.rs:16-118 — 15^3 block, five ray-sampled cuboid shells, ArtificialViscosity(1.0, 0.0), 200 steps of
    dt = 1/200: per-step contact counts and iteration counts against the oracle, positions against the stated tolerance
    while the flow is regular and against the oracle's own f32-vs-f64 distance once the splash makes it chaotic."""
    r, fluid, colliders, w, fl, coupling, make_oracle, pose_oracle = _basic3_worlds()
    o, o64 = make_oracle(), make_oracle(f64=True)
    n = 200
    checkpoints = (1, 10, 50, 100, 200)
    for k in range(n):
        coupling.update_boundaries(w)
        pose_oracle(o)
        pose_oracle(o64)
        st = w.step(DT, GRAVITY)
        so = o.step(DT, GRAVITY)
        o64.step(DT, GRAVITY)
        if k == 0:
            assert st.nparticles == 3375 and sum(len(c[0]) for c in colliders) == 5392 + 4 * 1648
            assert st.ncontacts == so.ncontacts, (st.ncontacts, so.ncontacts)
            assert (w.contact_counts(fl) == o.contact_counts(0)).all() and (w.contact_counts(fl, True) == o.contact_counts(0, True)).all()
            for b, h in enumerate(w.boundaries()):
                assert rel_err(h.volumes, o.boundary_volumes(b)) < 1e-5
        d = max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) / r
        noise = max_norm_diff(o.fluid_vec(0, "positions"), o64.fluid_vec(0, "positions")) / r
        # the two trajectories agree to rounding: contacts may differ only by pairs sitting on d = h
        slack = 0 if k == 0 else max(4, int(1e-4 * so.ncontacts))
        if d < 1e-2:
            assert abs(int(st.ncontacts) - int(so.ncontacts)) <= slack, (k, st.ncontacts, so.ncontacts)
            assert abs(st.n_pressure_iters - so.n_press_iters) <= 1 and abs(st.n_divergence_iters - so.n_div_iters) <= 2, (k, st, so)
        if k + 1 in checkpoints:
            # stated tolerance 1e-4 r per step, or what the reference's own arithmetic is worth on this flow:
            # the oracle's f32 and f64 runs are 1e-3 r apart after 50 steps, 0.08 r after 100 and decorrelated (6 r) after 200;
            # measured (profiles/r03_experiments/r03x_literal_scene_ratios.log): the HIP run stays 0.02-0.7 of that distance from
            # the oracle's f32 run at every checkpoint, so the bound is 2 x the oracle's own noise (round 2: 10 x)
            print(f"basic3 checkpoint {k + 1}: |hip - oracle| = {d:.3e} r, oracle |f32 - f64| = {noise:.3e} r, ratio {d / max(noise, 1e-30):.2f}")
            assert d < max(1e-4 * (k + 1), 2.0 * noise), f"after {k + 1} steps: {d:.3e} r vs oracle, oracle f32-f64 {noise:.3e} r"
    # bulk state after the splash: centre of mass and kinetic energy agree to a few per cent whatever the chaos did to particles
    po, vo = o.fluid_vec(0, "positions").astype(np.float64), o.fluid_vec(0, "velocities").astype(np.float64)
    pg, vg = fl.positions.astype(np.float64), fl.velocities.astype(np.float64)
    assert np.abs(pg.mean(axis=0) - po.mean(axis=0)).max() < 0.02
    assert abs((vg ** 2).sum() - (vo ** 2).sum()) < 0.05 * max((vo ** 2).sum(), 1e-6)
    assert pg[:, 1].min() > 0.1  # nothing fell through the ground (top face at y = 0.2, samples at 0.15)


def test_faucet3_literal_scene():
    """examples3d/faucet3.rs:19-109 — an empty fluid fed with a 10 x 10 sheet of particles every 12 steps (`add_particles`), XSPH +
    Akinci2013(1, 10), falling on a fixed ball whose boundary is 336 ray-sampled points coupled by StaticSampling; particles
    below y = -2 are deleted.  140 steps through the FluidsPipeline mirror against the oracle: particle counts, contacts while
    the two runs are the same flow, then the bulk state."""
    from oracle import oracle as O
    from salva_amd import Akinci2013SurfaceTension, Boundary, Fluid, XSPHViscosity
    from salva_amd.coupling import FluidsPipeline, RigidBody, StaticSampling

    sc = scenes.faucet3()
    r, sheet, g = sc["radius"], sc["sheet"], sc["gravity"]
    pipe = FluidsPipeline(r, 2.0)
    fl = Fluid(np.zeros((0, 3), np.float32), r, 1000.0)
    fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    fl.nonpressure_forces.append(Akinci2013SurfaceTension(1.0, 10.0))
    h = pipe.liquid_world.add_fluid(fl)
    bo = pipe.liquid_world.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    pipe.coupling.register_coupling(bo, "ground", RigidBody(dynamic=False), StaticSampling(sc["ball_samples"]))

    def oracle(f64):
        o = O.OracleWorld(r, 2.0, O.DFSPH, f64=f64)
        f = o.add_fluid(np.zeros((0, 3), np.float32), 1000.0)
        o.add_xsph(f, 0.5, 0.0)
        o.add_akinci2013(f, 1.0, 10.0)
        b = o.add_boundary(np.zeros((0, 3), np.float32))
        o.set_boundary_sampling(b, sc["ball_samples"])
        o.update_boundary_pose(b, (0, 0, 0), (0, 0, 0, 1), (0, 0, 0), (0, 0, 0), (0, 0, 0), True, False)
        return o

    o, o64 = oracle(False), oracle(True)
    nsteps = 140
    same_flow_steps = 0
    for k in range(nsteps):
        if k > 0 and k % 12 == 0:  # the callback of :62-103, every 0.06 s
            h.add_particles(sheet, np.zeros_like(sheet))
            for ow in (o, o64):
                ow.add_particles(0, sheet, np.zeros_like(sheet))
        st = pipe.step(g, DT)
        so = o.step(DT, g)
        o64.step(DT, g)
        assert h.num_particles() == o.fluid_len(0) == 100 * (k // 12)
        if h.num_particles():
            d = max_norm_diff(h.positions, o.fluid_vec(0, "positions")) / r
            if k + 1 in (24, 36, 48, 60, 72, 96):
                noise_k = max_norm_diff(o.fluid_vec(0, "positions"), o64.fluid_vec(0, "positions")) / r
                print(f"faucet3 checkpoint {k + 1}: |hip - oracle| = {d:.3e} r, oracle |f32 - f64| = {noise_k:.3e} r")
                # while the sheets are still one flow (the oracle's own f32 and f64 runs within 1e-3 r of each other: up to step
                # ~50): the stated tolerance or 3 x the oracle's own noise.  Past that point every pair of runs decorrelates within
                # a dozen steps — which dozen is itself chaotic: with one summation order of the density pass the HIP run sat at
                # 0.067 r against the oracle's 0.099 r at step 72, with another at 0.13 r against 0.003 r at step 60
                # (profiles/r03_experiments/r03x_*) — so only the bulk checks below apply then.
                if noise_k < 1e-3:
                    assert d < max(1e-4 * (k + 1), 3.0 * noise_k), f"after {k + 1} steps: {d:.3e} r vs oracle, oracle f32-f64 {noise_k:.3e} r"
            if d < 1e-2:
                same_flow_steps += 1
                assert abs(int(st.ncontacts) - int(so.ncontacts)) <= max(8, int(2e-4 * so.ncontacts)), (k, st.ncontacts, so.ncontacts)
    assert same_flow_steps >= 36, same_flow_steps  # (a free sheet under Akinci cohesion wrinkles chaotically: the oracle's f32 and f64 runs part as early)
    po = o.fluid_vec(0, "positions")
    pg = h.positions
    noise = max_norm_diff(po, o64.fluid_vec(0, "positions")) / r
    d = max_norm_diff(pg, po) / r
    print(f"faucet3 after {nsteps} steps: |hip - oracle| = {d:.3e} r, oracle |f32 - f64| = {noise:.3e} r, ratio {d / max(noise, 1e-30):.2f}, same-flow steps {same_flow_steps}")
    # (by now the runs are decorrelated — tens of radii apart, the oracle's f32 and f64 runs as much as the HIP run: two maxima
    # over chaotic trajectories are compared, measured ratio 2.0 — so the bound here stays a factor, 4 x, and the bulk checks
    # below carry the meaning)
    assert d < max(1e-4 * nsteps, 4.0 * noise), f"after {nsteps} steps: {d:.3e} r vs oracle, oracle f32-f64 {noise:.3e} r"
    # the sheets that reached the ball (top at y = 0.15) were deflected, not swallowed: nothing inside the sampled sphere
    assert np.linalg.norm(pg, axis=1).min() > 0.12 and np.linalg.norm(po, axis=1).min() > 0.12
    assert abs(pg[:, 1].mean() - po[:, 1].mean()) < 2 * r


def test_api_semantics_match_reference():
    """Velocity lag, dt lag, host edits between steps, add / delete particles, remove fluid (swap-remove)."""
    from salva_amd import DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity

    pos = scenes.jitter(scenes.cube_fluid_positions(6, 6, 6, R), 0.1 * R)
    vel = scenes.random_velocities(len(pos), 0.1)
    s = Scene(R, 2.0, "dfsph")
    s.add_fluid(pos, vel, 1000.0, forces=[("xsph", 0.5, 0.0)])
    w, (fl,), _ = s.make_hip()
    o = s.make_oracle()
    x0 = fl.positions.copy()
    st = w.step(DT, GRAVITY)
    so = o.step(DT, GRAVITY)
    # first step: inv_dt = 0 -> the divergence tolerance is exactly 0 (dt lag).  The loop then only stops early if
    # every clamped divergence is *exactly* zero, which depends on f32 summation order (the oracle itself flips under
    # shuffle_seed), so only the protocol bounds are asserted here; the states still agree to tolerance below.
    assert 1 <= st.n_divergence_iters <= 50 and 1 <= so.n_div_iters <= 50
    assert max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) < 1e-4 * R
    v, dv = fl.velocities, w.velocity_changes(fl)
    assert np.allclose(fl.positions, x0 + (v + dv) * np.float32(DT), atol=1e-6)  # x += (v + dv) dt, v not updated (:411-420)
    # host edit between steps (heightfield3.rs:40 overwrites velocities): assign + step on both sides
    newv = scenes.random_velocities(len(pos), 0.05, seed=7)
    fl.velocities = newv
    o.set_fluid_velocities(0, newv)
    w.step(DT, GRAVITY)
    o.step(DT, GRAVITY)
    assert max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) < 2e-4 * R
    assert max_norm_diff(w.velocity_changes(fl), o.fluid_vec(0, "velocity_changes")) < 1e-3
    # add particles (faucet3.rs:99-103) then delete some (faucet3.rs:69-73)
    extra = scenes.cube_fluid_positions(3, 3, 3, R) + np.float32([0.0, 0.6, 0.0])
    fl.add_particles(extra, np.zeros_like(extra))
    assert fl.num_particles() == len(pos) + 27
    w.step(DT, GRAVITY)
    assert fl.positions.shape == (len(pos) + 27, 3) and np.isfinite(fl.positions).all()
    for i in (0, 5, 100):
        fl.delete_particle_at_next_timestep(i)
    before = fl.positions.copy()
    w.step(DT, GRAVITY)
    assert fl.num_particles() == len(pos) + 24
    keep = np.ones(len(before), bool)
    keep[[0, 5, 100]] = False
    assert np.max(np.abs(fl.positions - before[keep])) < 0.05  # order-preserving compaction (helper.rs:4-12)
    # second fluid + swap-remove of the first
    f2 = Fluid(scenes.cube_fluid_positions(4, 4, 4, R) + np.float32([1.0, 0.0, 0.0]), R, 500.0)
    f2.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f2)
    w.step(DT, GRAVITY)
    assert w.remove_fluid(fl) is fl and len(w.fluids()) == 1 and f2._slot == 0
    w.step(DT, GRAVITY)
    assert np.isfinite(f2.positions).all() and f2.num_particles() == 64


def test_edge_cases():
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, _lib

    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    st = w.step(DT, GRAVITY)  # empty world
    assert st.nparticles == 0
    f = w.add_fluid(Fluid(np.zeros((0, 3), np.float32), R, 1000.0))  # empty fluid (faucet3.rs:39-44 starts like this)
    w.step(DT, GRAVITY)
    f.add_particles(np.float32([[0.0, 1.0, 0.0]]))  # a single particle: only the self contact, alpha = 0
    st = w.step(DT, GRAVITY)
    assert st.ncontacts == 1
    y0 = float(f.positions[0, 1])
    w.step(DT, GRAVITY)
    assert f.positions[0, 1] < y0  # free fall
    # two coincident particles: |d| = 0 -> gradient 0 (kernel.rs:18-24), W = W(0)
    g = w.add_fluid(Fluid(np.float32([[5.0, 0.0, 0.0], [5.0, 0.0, 0.0]]), R, 1000.0))
    st = w.step(DT, GRAVITY)
    assert np.isfinite(g.positions).all()
    assert (w.contact_counts(g) == 2).all()
    # NaN positions are reported, not propagated silently (the reference would panic in hgrid.rs:42)
    bad = LiquidWorld(DFSPHSolver(), R, 2.0)
    bad.add_fluid(Fluid(np.float32([[np.nan, 0.0, 0.0], [0.0, 0.0, 0.0]]), R, 1000.0))
    with pytest.raises(_lib.SalvaHipError) as e:
        bad.step(DT, GRAVITY)
    assert e.value.code == _lib.E_NUMERIC
    # WCSPHSurfaceTension with a boundary coefficient panics in the reference (wcsph_surface_tension.rs:66-83): rejected
    from salva_amd import WCSPHSurfaceTension
    wt = LiquidWorld(DFSPHSolver(), R, 2.0)
    ft = Fluid(scenes.cube_fluid_positions(3, 3, 3, R), R, 1000.0)
    ft.nonpressure_forces.append(WCSPHSurfaceTension(0.5, 0.5))
    wt.add_fluid(ft)
    with pytest.raises(_lib.SalvaHipError) as e:
        wt.step(DT, GRAVITY)
    assert e.value.code == _lib.E_INVALID
    # dt <= eps: no substep (timestep_manager.rs:56-58)
    w2 = LiquidWorld(DFSPHSolver(), R, 2.0)
    h = w2.add_fluid(Fluid(scenes.cube_fluid_positions(3, 3, 3, R), R, 1000.0))
    x = h.positions.copy()
    w2.step(0.0, GRAVITY)
    assert np.array_equal(h.positions, x)


def test_large_block_invariants():
    """Size-independent properties at 64^3 = 262k particles (the oracle needs ~10 s per step there, so only step 1 is
    compared directly): contact symmetry (sum of counts = ncontacts, even off-diagonal), momentum conservation of the
    pressure solver, run-to-run determinism."""
    n = 64
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R)
    vel = scenes.random_velocities(len(pos), 0.1)
    s = Scene(R, 2.0, "dfsph")
    s.add_fluid(pos, vel, 1000.0)
    s.solver_params.update(max_divergence_iter=3)
    w, (fl,), _ = s.make_hip()
    st = w.step(DT, (0.0, 0.0, 0.0))
    cnt = w.contact_counts(fl)
    assert int(cnt.sum()) == st.ncontacts and (st.ncontacts - len(pos)) % 2 == 0
    m = np.float64(fl.particle_mass(0))
    p0 = m * vel.astype(np.float64).sum(axis=0)
    p1 = m * (fl.velocities.astype(np.float64) + w.velocity_changes(fl).astype(np.float64)).sum(axis=0)
    assert np.max(np.abs(p1 - p0)) < 1e-5 * m * np.abs(vel).sum()
    o = s.make_oracle(threads=8)
    so = o.step(DT, (0.0, 0.0, 0.0))
    assert so.ncontacts == st.ncontacts
    assert (o.contact_counts(0) == cnt).all()
    assert rel_err(w.densities(fl), o.fluid_scalar(0, "densities")) < 1e-5
    assert max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) < 1e-4 * R
    # determinism: a second world fed the same input gives bit-identical output
    w2, (fl2,), _ = s.make_hip()
    w2.step(DT, (0.0, 0.0, 0.0))
    assert np.array_equal(fl.positions, fl2.positions) and np.array_equal(w.velocity_changes(fl), w2.velocity_changes(fl2))


def test_cpp_mirror_basic3_example():
    """examples/basic3.cpp drives the scene of examples3d/basic3.rs through include/salva_hip.hpp (the C++ mirror of the
    salva3d host API) and the C ABI; 120 steps of the dam break must run, keep the fluid above the ground and converge."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "basic3")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples")])
    out = subprocess.run([exe, "120"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("step 120: 3375 particles"), last


@pytest.mark.parametrize("name", ["two_phase", "iisph_akinci", "dfsph_tank"])
def test_contact_sets_match_oracle(name):
    """Not just the counts: the exported contact lists (salva_hip_get_fluid_contacts) name exactly the partners the
    oracle's contact manager holds — fluid-fluid across models, fluid-boundary, self contacts included."""
    builder, _ = SCENES[name]
    scene = builder()
    w, fls, _ = scene.make_hip()
    o = scene.make_oracle()
    w.step(DT, GRAVITY)
    o.step(DT, GRAVITY)
    for f, h in enumerate(fls):
        for boundary in (False, True):
            if boundary and not scene.boundaries:
                continue
            off, jm, j = w.fluid_contacts(h, boundary)
            assert len(off) == h.num_particles() + 1 and off[-1] == len(j)
            keys = (jm.astype(np.uint64) << np.uint64(32)) | j.astype(np.uint64)
            for i in range(h.num_particles()):
                got = sorted(int(k) for k in keys[int(off[i]):int(off[i + 1])])
                ref = [(m << 32) | k for m, k in o.contacts_of(f, i, boundary)]
                assert got == ref, f"{name}: fluid {f} particle {i} boundary={boundary}"
            if not boundary:
                # every particle lists itself
                self_key = (np.uint64(f) << np.uint64(32)) | np.arange(h.num_particles(), dtype=np.uint64)
                assert all(self_key[i] in keys[int(off[i]):int(off[i + 1])] for i in range(0, h.num_particles(), 37))


def _sparse_scene():
    """A dense block plus a spray of isolated particles over a 60x larger box: thousands of tiles, almost all empty or
    holding a single particle — the shape that makes fixed-stride slot tables too wasteful, so the compact ones are used."""
    s = Scene(R, 2.0, "dfsph")
    block = scenes.jitter(scenes.cube_fluid_positions(8, 8, 8, R), 0.1 * R, seed=42)
    u = scenes.lcg_uniform(3 * 1500, 99).reshape(-1, 3)
    spray = ((u - np.float32(0.5)) * np.float32(6.0)).astype(np.float32)
    pos = np.concatenate([block, spray]).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.2, seed=5)
    s.add_fluid(pos, vel, 1000.0, forces=[("xsph", 0.5, 0.0)])
    return s


@pytest.mark.parametrize("compact", [False, True])
def test_sparse_scene_and_compact_slot_tables(compact, monkeypatch):
    """Isolated particles (lists holding only the self contact, below the 20-contact divergence threshold), empty
    tiles, and both layouts of the per-tile slot tables give the oracle's trajectory."""
    if compact:
        monkeypatch.setenv("SALVA_HIP_COMPACT_HALO", "1")
    for scene, nsteps in ((_sparse_scene(), 4), (SCENES["dfsph_tank"][0](), 4)):
        compare(run_hip(scene, nsteps), run_oracle(scene, nsteps), scene, nsteps, f"sparse/compact={compact}")


def test_plane_layouts_and_the_fused_first_divergence_change_rounding_only(monkeypatch):
    """Round 4: with a uniform particle mass the solver kernels stage 24 / 16 bytes per halo slot as 8-byte planes and the first
    divergence evaluate rides in the density pass.  Both are other kernels for the same sums: against the
    32 / 20-byte layouts and the separate pass (SALVA_HIP_NO_PLANES, SALVA_HIP_NO_FUSED_DIV) the iteration counts and contact
    counts are identical and the states agree to f32 summation order; a scene whose masses differ takes the old kernels whatever
    the switches say and is bit-identical."""
    def run(scene, nsteps, env):
        for k in ("SALVA_HIP_NO_PLANES", "SALVA_HIP_NO_FUSED_DIV", "SALVA_HIP_NO_TWO_MASS"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "1")
        return run_hip(scene, nsteps)

    tank = SCENES["dfsph_tank"][0]()
    nsteps = 12
    base = run(tank, nsteps, ())
    for env in (("SALVA_HIP_NO_FUSED_DIV",), ("SALVA_HIP_NO_PLANES",)):
        other = run(tank, nsteps, env)
        assert np.array_equal(base["iters"], other["iters"]), f"{env}: iteration or contact counts differ"
        dp, dv = np.abs(base["pos_0"] - other["pos_0"]).max(), np.abs(base["vel_0"] - other["vel_0"]).max()
        assert dp < 2e-5 * R * nsteps, f"{env}: positions differ by {dp / R:.2e} r"
        assert dv < 1e-4, f"{env}: velocities differ by {dv:.2e} m/s"
    # fluids of different density0: the masses differ, and without the two-mass form of the plane layouts (round 5, next test) nothing
    # above applies
    two = SCENES["two_phase"][0]()
    a, b = run(two, 6, ("SALVA_HIP_NO_TWO_MASS",)), run(two, 6, ("SALVA_HIP_NO_TWO_MASS", "SALVA_HIP_NO_PLANES", "SALVA_HIP_NO_FUSED_DIV"))
    assert np.array_equal(a["iters"], b["iters"])
    for f in range(2):
        assert np.array_equal(a[f"pos_{f}"], b[f"pos_{f}"]) and np.array_equal(a[f"vel_{f}"], b[f"vel_{f}"])


def _two_phase_side_by_side():
    """BASELINE config 4 in small: two blocks of different density0 side by side along x over a floor — most tiles see one mass in
    their whole halo, the tiles around the interface two."""
    s = Scene(R, 2.0, "dfsph")
    a = scenes.jitter(scenes.cube_fluid_positions(24, 12, 12, R), 0.05 * R, seed=42)
    b = scenes.jitter(scenes.cube_fluid_positions(24, 12, 12, R), 0.05 * R, seed=43)
    a[:, 0] -= np.float32(24 * R)
    b[:, 0] += np.float32(24 * R)
    a[:, 1] += np.float32(12 * R + 2 * R)
    b[:, 1] += np.float32(12 * R + 2 * R)
    s.add_fluid(a, scenes.random_velocities(len(a), 0.05, seed=5), 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_fluid(b, scenes.random_velocities(len(b), 0.05, seed=6), 500.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(scenes.plane_lattice(56, 20, 0.0, R, -28 * 2 * R + R, -10 * 2 * R + R, layers=2))
    return s


def test_two_mass_worlds_take_the_plane_layouts(monkeypatch):
    """Round 5: a world with exactly two particle masses — two fluids of different density0 with `Fluid::new`'s uniform volumes,
    BASELINE config 4 — runs the plane-layout kernels too: k_nbr_tile writes the lists of the tiles that see both
    masses with the lighter class first, and the kernels add (m_b - m_a) x the sum over the heavier tail segment to m_a x the sum
    over the whole list.  Other kernels for the same sums: against a run with SALVA_HIP_NO_TWO_MASS=1 (the general kernels) the
    contact and iteration counts are identical and the states agree to f32 summation order — but not bit for bit, or the path
    never switched on — and both agree with the oracle.  A third fluid with one of the two masses changes nothing; a third MASS under
    round 5's rule (SALVA_HIP_MAX_MASSES=2; round 6 takes up to four, next test), or a fluid with non-uniform volumes, falls back to
    the general kernels (bit-identical to NO_TWO_MASS)."""
    scene = _two_phase_side_by_side()
    nsteps = 10
    monkeypatch.delenv("SALVA_HIP_NO_TWO_MASS", raising=False)
    on = run_hip(scene, nsteps)
    monkeypatch.setenv("SALVA_HIP_NO_TWO_MASS", "1")
    off = run_hip(scene, nsteps)
    monkeypatch.delenv("SALVA_HIP_NO_TWO_MASS", raising=False)
    assert np.array_equal(on["iters"], off["iters"]), "iteration or contact counts differ"
    differs = False
    for f in range(2):
        dp, dv = np.abs(on[f"pos_{f}"] - off[f"pos_{f}"]).max(), np.abs(on[f"vel_{f}"] - off[f"vel_{f}"]).max()
        assert dp < 2e-5 * R * nsteps and dv < 1e-4, (f, dp / R, dv)
        differs |= not np.array_equal(on[f"vel_{f}"], off[f"vel_{f}"])
    assert differs, "bit-identical runs: the two-mass path never switched on"
    compare(on, run_oracle(scene, nsteps), scene, nsteps, "two-phase side by side (two-mass plane layouts) vs oracle")

    # three fluids, two masses (the third shares the first one's): still the two-mass path, still the oracle's results
    three = _two_phase_side_by_side()
    top = scenes.jitter(scenes.cube_fluid_positions(10, 6, 10, R), 0.05 * R, seed=44)
    top[:, 1] += np.float32(12 * R + 2 * R + 24 * R + 4 * R)
    three.add_fluid(top, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    compare(run_hip(three, 6), run_oracle(three, 6), three, 6, "three fluids, two masses vs oracle")

    # a third mass, and non-uniform volumes: the general kernels, whatever the switch says
    for variant in ("third_mass", "volumes"):
        sc = _two_phase_side_by_side()
        if variant == "third_mass":
            sc.add_fluid(top, None, 800.0, forces=[("xsph", 0.5, 0.0)])
            monkeypatch.setenv("SALVA_HIP_MAX_MASSES", "2")
        else:
            vol = np.full(len(sc.fluids[0]["pos"]), 0.8 * (2 * R) ** 3, np.float32)
            vol[::7] *= np.float32(1.01)
            sc.fluids[0]["volumes"] = vol
        a = run_hip(sc, 4)
        monkeypatch.setenv("SALVA_HIP_NO_TWO_MASS", "1")
        b = run_hip(sc, 4)
        monkeypatch.delenv("SALVA_HIP_NO_TWO_MASS", raising=False)
        monkeypatch.delenv("SALVA_HIP_MAX_MASSES", raising=False)
        for key in a:
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key]), equal_nan=True), (variant, key)


def _phases_around_a_line(rho0s, lift=2):
    """2 x 2 blocks of different density0 (the last may repeat) meeting along a vertical line over a floor, dropped from `lift` radii:
    tiles around the line see every mass, tiles along the faces two, the others one."""
    s = Scene(R, 2.0, "dfsph")
    for k, ((sx, sz), rho0) in enumerate(zip(((-1, -1), (1, -1), (-1, 1), (1, 1)), rho0s)):
        p = scenes.jitter(scenes.cube_fluid_positions(12, 10, 12, R), 0.05 * R, seed=42 + k)
        p[:, 0] += np.float32(sx * 12 * R)
        p[:, 2] += np.float32(sz * 12 * R)
        p[:, 1] += np.float32(10 * R + lift * R)
        s.add_fluid(p, scenes.random_velocities(len(p), 0.05, seed=7 + k), rho0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(scenes.plane_lattice(30, 30, 0.0, R, -15 * 2 * R + R, -15 * 2 * R + R, layers=2))
    return s


@pytest.mark.parametrize("rho0s", [(1000.0, 800.0, 600.0, 600.0), (1000.0, 800.0, 600.0, 400.0), (400.0, 1000.0, 600.0, 800.0)])
def test_worlds_with_three_and_four_masses_take_the_plane_layouts_too(monkeypatch, rho0s):
    """Round 6: up to four particle masses — the lists of a tile whose halo holds several get one segment per
    mass class, lightest first (one walk of the candidates per class present), the kernels count everything behind the first segment
    with the second class's mass inside their loop and add (m_c - m_b) / (m_d - m_b) x the sums over the third / fourth segment on
    top (pairs.h pair_tail_*).  Same protocol as for two masses: iteration and contact counts of the general kernels
    (SALVA_HIP_NO_TWO_MASS=1), states equal to f32 summation order but not bit for bit, and the oracle's results.  With
    SALVA_HIP_MAX_MASSES one below the scene's count the world falls back to the general kernels, bit for bit.  (Opt-in through
    that switch: on four columns of 10^6 particles the general kernels were 8-10 % faster, profiles/r06_experiments/r06l_*.)"""
    scene = _phases_around_a_line(rho0s)
    nsteps = 10
    monkeypatch.delenv("SALVA_HIP_NO_TWO_MASS", raising=False)
    monkeypatch.setenv("SALVA_HIP_MAX_MASSES", "4")  # (opt-in: world.h max_masses)
    on = run_hip(scene, nsteps)
    monkeypatch.setenv("SALVA_HIP_NO_TWO_MASS", "1")
    off = run_hip(scene, nsteps)
    monkeypatch.delenv("SALVA_HIP_NO_TWO_MASS", raising=False)
    assert np.array_equal(on["iters"], off["iters"]), "iteration or contact counts differ"
    assert on["iters"][:, 0].max() >= 5  # (the blocks do land within the run)
    differs = False
    for f in range(4):
        dp, dv = np.abs(on[f"pos_{f}"] - off[f"pos_{f}"]).max(), np.abs(on[f"vel_{f}"] - off[f"vel_{f}"]).max()
        assert dp < 2e-5 * R * nsteps and dv < 1e-4, (f, dp / R, dv)
        differs |= not np.array_equal(on[f"vel_{f}"], off[f"vel_{f}"])
    assert differs, "bit-identical runs: the plane layouts never switched on"
    compare(on, run_oracle(scene, nsteps), scene, nsteps, f"{len(set(rho0s))} masses around a line (plane layouts) vs oracle")
    monkeypatch.setenv("SALVA_HIP_MAX_MASSES", str(len(set(rho0s)) - 1))
    capped = run_hip(scene, 4)
    monkeypatch.setenv("SALVA_HIP_NO_TWO_MASS", "1")
    general = run_hip(scene, 4)
    for key in capped:
        assert np.array_equal(np.asarray(capped[key]), np.asarray(general[key]), equal_nan=True), key


@pytest.mark.parametrize("name", ["dfsph_tank", "iisph_akinci", "dfsph_xsph_block", "two_phase", "four_phase"])
def test_every_lds_layout_instantiation_computes_the_same_bits(monkeypatch, name):
    """The solver kernels exist in up to four instantiations each — the second staged array / plane at one of three compile-time
    distances (three, two, one tile per CU) or at a run-time distance — picked from the launch's largest halo (pairs.h pick_ds*).
    Halos large enough for the upper ones take a 10^6-particle column under compression, so SALVA_HIP_DS_LEVEL pushes the pick up
    by one, two, three levels on a small scene.  The layout moves LDS addresses and nothing else: every level must reproduce
    level 0 bit for bit, in both kernel families (plane layouts / SALVA_HIP_NO_PLANES)."""
    builder, nsteps = SCENES[name]
    if name == "four_phase":
        monkeypatch.setenv("SALVA_HIP_MAX_MASSES", "4")  # (the instantiations for three and four masses: opt-in, world.h max_masses)
    for planes in (True, False):
        runs = []
        for level in (0, 1, 2, 3):
            monkeypatch.setenv("SALVA_HIP_DS_LEVEL", str(level))
            if planes:
                monkeypatch.delenv("SALVA_HIP_NO_PLANES", raising=False)
            else:
                monkeypatch.setenv("SALVA_HIP_NO_PLANES", "1")
            runs.append(run_hip(builder(), nsteps))
        for level in (1, 2, 3):
            for key in runs[0]:
                if key.startswith("bforce_"):  # (atomically accumulated: the last bit depends on the order the tiles arrive in)
                    continue
                a, b = np.asarray(runs[0][key]), np.asarray(runs[level][key])
                assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), f"{name} planes={planes} level {level}: {key} differs"


def test_counting_sort_by_cell_is_the_radix_sort_bit_for_bit(monkeypatch):
    """Round 4: the fluid's sort of every step is a counting sort by cell (atomic counts -> scan = the cell table -> scatter ->
    ascending source index within each cell, grid.hip cell_sort); SALVA_HIP_RADIX_SORT=1 brings back the stable radix sort +
    k_cell_start it replaced.  A stable sort from the identity is "by key, then by index", so both must give the same permutation
    — the same summation orders, the same bits — on dense scenes, with two fluids, with a bounding box of 3 x 10^7 mostly empty cells,
    and with a few hundred particles crowded into one cell (k_cell_order is linear in a cell's population per particle)."""
    def both(make, nsteps):
        monkeypatch.setenv("SALVA_HIP_RADIX_SORT", "0")  # (unset: by size — the box of the strays scene would keep the radix sort)
        a = run_hip(make(), nsteps)
        monkeypatch.setenv("SALVA_HIP_RADIX_SORT", "1")
        b = run_hip(make(), nsteps)
        monkeypatch.delenv("SALVA_HIP_RADIX_SORT", raising=False)
        for key in a:
            if key.startswith("bforce_"):  # (atomically accumulated)
                continue
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key]), equal_nan=True), f"{key} differs"

    both(SCENES["dfsph_tank"][0], 8)
    both(SCENES["two_phase"][0], 6)

    def strays():
        s = Scene(R, 2.0, "dfsph")
        block = scenes.jitter(scenes.cube_fluid_positions(8, 8, 8, R), 0.1 * R, seed=42)
        far = np.array([[30.0, 40.0, -25.0], [-12.0, 3.0, 8.0], [0.3, -35.0, 0.1], [30.02, 40.01, -25.0]], np.float32)
        pos = np.concatenate([block, far]).astype(np.float32)
        s.add_fluid(pos, scenes.random_velocities(len(pos), 0.3, seed=5), 1000.0, forces=[("xsph", 0.5, 0.0)])
        return s

    both(strays, 4)

    def crowd():
        s = Scene(R, 2.0, "dfsph")
        rng = np.random.default_rng(3)
        block = scenes.jitter(scenes.cube_fluid_positions(6, 6, 6, R), 0.1 * R, seed=42)
        knot = (np.float32([0.505, 0.505, 0.505]) + rng.uniform(0.0, 0.9 * 4 * R, size=(300, 3))).astype(np.float32)  # one cell, h = 4 r
        pos = np.concatenate([block, knot]).astype(np.float32)
        s.add_fluid(pos, scenes.random_velocities(len(pos), 0.1, seed=6), 1000.0, forces=[("xsph", 0.5, 0.0)])
        return s

    both(crowd, 3)


def test_stray_particles_far_from_the_bulk():
    """A few particles hundreds of cells away from the block (what a leaking wall produces, in the reference too) blow
    the cell bounding box up to tens of millions of empty cells: per-tile tables are compact over non-empty tiles and the
    cell table is filled by chunks, so the step still matches the oracle and stays cheap."""
    s = Scene(R, 2.0, "dfsph")
    block = scenes.jitter(scenes.cube_fluid_positions(8, 8, 8, R), 0.1 * R, seed=42)
    strays = np.array([[30.0, 40.0, -25.0], [-12.0, 3.0, 8.0], [0.3, -35.0, 0.1], [30.02, 40.01, -25.0]], np.float32)
    pos = np.concatenate([block, strays]).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.3, seed=5)
    s.add_fluid(pos, vel, 1000.0, forces=[("xsph", 0.5, 0.0)])
    got, ref = run_hip(s, 4), run_oracle(s, 4)
    compare(got, ref, s, 4, "strays")
    w, (fl,), _ = s.make_hip()
    st = w.step(DT, GRAVITY)
    assert st.step_ms < 20.0, f"a step over a mostly empty 30M-cell box took {st.step_ms:.1f} ms"


def test_strays_beyond_the_cell_table_budget_fold_the_grid_or_are_refused_with_advice(monkeypatch):
    """The dense cell table costs 4 bytes per cell of the bounding box (the reference's hash grid: per occupied cell).  Round 5: a box
    that is mostly empty is FOLDED (device_types.h TileGrid) — a stray 1500 cells away costs a table of a few 10^5 cells and the step
    agrees with the oracle.  Where folding is off (SALVA_HIP_NO_FOLD=1; decomposed runs, dynamic contact sampling) the step fails
    past the budget (8 GiB, SALVA_HIP_CELL_TABLE_GIB) with E_CAPACITY and says what to do; the world stays usable once the stray is
    gone."""
    from salva_amd import _lib

    s = Scene(R, 2.0, "dfsph")
    block = scenes.jitter(scenes.cube_fluid_positions(6, 6, 6, R), 0.1 * R, seed=42)
    pos = np.concatenate([block, np.float32([[150.0, 140.0, 160.0]])]).astype(np.float32)  # 1500 x 1400 x 1600 cells = 12.5 GiB
    s.add_fluid(pos, None, 1000.0)
    monkeypatch.delenv("SALVA_HIP_NO_FOLD", raising=False)
    compare(run_hip(s, 3), run_oracle(s, 3), s, 3, "a stray 1500 cells away (folded grid)")
    monkeypatch.setenv("SALVA_HIP_NO_FOLD", "1")
    w, (fl,), _ = s.make_hip()
    monkeypatch.delenv("SALVA_HIP_NO_FOLD", raising=False)
    with pytest.raises(_lib.SalvaHipError) as e:
        w.step(DT, GRAVITY)
    assert e.value.code == _lib.E_CAPACITY and "SALVA_HIP_CELL_TABLE_GIB" in str(e.value) and "delete strays" in str(e.value)
    fl.delete_particle_at_next_timestep(len(pos) - 1)
    st = w.step(DT, GRAVITY)
    assert st.nparticles == len(block)


def _scene_with_leaked_particles(solver="dfsph", two=False):
    """A tank with its block of fluid, and particles that have left it: singles, a pair in contact, and a small cluster — some of them
    exactly a multiple of a small power of two of cells away from the block, so that a folded grid puts them into the block's own
    cells."""
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(10, 8, 10, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=11)
    h = 4.0 * R
    rng = np.random.default_rng(3)
    cluster = scenes.jitter(scenes.cube_fluid_positions(3, 3, 3, R), 0.1 * R, seed=12) + np.float32([37.0 * h, -64.0 * h, 5.0 * h])
    images = fluid[rng.choice(len(fluid), 6, replace=False)] + np.float32([32.0 * h, 0.0, 0.0]) * np.arange(1, 7, dtype=np.float32)[:, None]
    far = np.float32([[90.0 * h, -200.0 * h, -75.0 * h], [90.0 * h + R, -200.0 * h, -75.0 * h], [-128.0 * h, 16.0 * h, 64.0 * h]])
    pos = np.concatenate([fluid, cluster, images, far]).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.2, seed=13)
    forces = [("xsph", 0.5, 0.0)] if solver == "dfsph" else [("akinci", 1.0, 10.0)]
    if two:
        half = len(pos) // 2
        s.add_fluid(pos[:half], vel[:half], 1000.0, forces=forces)
        s.add_fluid(pos[half:], vel[half:], 500.0, forces=forces)
    else:
        s.add_fluid(pos, vel, 1000.0, forces=forces)
    s.add_boundary(shell)
    return s


@pytest.mark.parametrize("variant", ["dfsph", "iisph", "two_mass"])
def test_a_folded_grid_changes_nothing_but_the_table(monkeypatch, variant):
    """device_types.h TileGrid: with the cell coordinates taken modulo a power-of-two period the table is a torus; particles of other
    images become candidates and fail the exact distance test, so the contact SETS — hence every sum, up to the order of its terms —
    are the unfolded grid's.  Three worlds of one scene (a tank, leaked particles up to 200 cells away, some of them folded right into
    the block's cells): never folded, folded by the rule (the box holds far more cells than particles), folded as hard as the
    boundary grid allows (SALVA_HIP_FOLD_CELLS=8 -> the periods the tank's own width dictates).  The first step — whose sort starts
    from the host's order in all three — is the same bit for bit; afterwards the order inside a cell is the previous step's order,
    which follows the numbering of the tiles, so later steps agree in every count and to summation order in the state.  All agree
    with the oracle."""
    s = _scene_with_leaked_particles("iisph" if variant == "iisph" else "dfsph", two=variant == "two_mass")
    nsteps = 5
    runs = {}
    for name, env in (("never", {"SALVA_HIP_NO_FOLD": "1"}), ("rule", {}), ("hard", {"SALVA_HIP_FOLD_CELLS": "8"})):
        for k in ("SALVA_HIP_NO_FOLD", "SALVA_HIP_FOLD_CELLS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        runs[name] = run_hip(s, nsteps)
    for k in ("SALVA_HIP_NO_FOLD", "SALVA_HIP_FOLD_CELLS"):
        monkeypatch.delenv(k, raising=False)
    ref = runs["never"]
    for name in ("rule", "hard"):
        got = runs[name]
        assert np.array_equal(got["iters"], ref["iters"]), (name, got["iters"], ref["iters"])
        for f in range(len(s.fluids)):
            for key in (f"s1_nff_{f}", f"s1_nfb_{f}", f"s1_density_{f}", f"s1_alpha_{f}", f"s1_pos_{f}", f"s1_vel_{f}"):
                assert np.array_equal(got[key], ref[key]), (name, key)
            dp, dv = np.abs(got[f"pos_{f}"] - ref[f"pos_{f}"]).max(), np.abs(got[f"vel_{f}"] - ref[f"vel_{f}"]).max()
            assert dp < 2e-5 * R * nsteps and dv < 1e-4, (name, f, dp / R, dv)
    compare(runs["hard"], run_oracle(s, nsteps), s, nsteps, f"leaked particles, folded grid ({variant}) vs oracle")


def test_an_isolated_particle_in_a_dense_slice_has_alpha_zero():
    """dfsph_solver.rs:208: alpha = 0 where sum |m grad W|^2 + |sum m grad W|^2 <= 1e-5 — a particle whose only contact is itself.
    Here such a particle shares its tile, hence its 64-particle slice and the slice's padded trip count, with a dense block one empty
    cell away: every padded trip evaluates the kernel at distance zero, and the gradient factor there has to be exactly 0 (sph_math.h
    kernel_wg2: a contracted a2 a2 - a1 a1 left 4e-10 x 1e15 in it, sixteen trips of which crossed the 1e-5).  Default path and the
    general kernels, DFSPH alpha and the IISPH diagonal a_ii of the same pass."""
    import os

    h = 4 * R
    block = scenes.jitter(scenes.cube_fluid_positions(4, 8, 8, R), 0.05 * R, seed=3)   # cells 0..1 x 0..3 x 0..3 of one tile
    block += np.float32([R, R, R]) - block.min(axis=0)
    lone = np.float32([[3.5 * h, 1.5 * h, 1.5 * h]])                                     # cell 3: one empty cell from the block
    pos = np.concatenate([block, lone]).astype(np.float32)
    for solver in ("dfsph", "iisph"):
        s = Scene(R, 2.0, solver)
        s.add_fluid(pos, None, 1000.0)
        for env in ({}, {"SALVA_HIP_NO_PLANES": "1", "SALVA_HIP_NO_FUSED_DIV": "1"}):
            old = {k: os.environ.pop(k, None) for k in ("SALVA_HIP_NO_PLANES", "SALVA_HIP_NO_FUSED_DIV")}
            os.environ.update(env)
            try:
                w, (fl,), _ = s.make_hip()
                w.step(DT, GRAVITY)
            finally:
                for k in env:
                    os.environ.pop(k, None)
                os.environ.update({k: v for k, v in old.items() if v is not None})
            o = s.make_oracle()
            o.step(DT, GRAVITY)
            assert w.contact_counts(fl)[-1] == 1 and w.contact_counts(fl)[:-1].min() >= 7 and w.contact_counts(fl)[:-1].max() >= 27
            if solver == "dfsph":
                a, ao = w.alphas(fl), o.fluid_scalar(0, "alphas")
                assert ao[-1] == 0.0 and a[-1] == 0.0, (env, a[-1])
                assert rel_err(a[:-1], ao[:-1], floor=float(np.abs(ao).max()) * 1e-3) < 1e-4
            assert max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) < 1e-4 * R, (solver, env)


def test_device_side_add_and_delete_match_the_host_path():
    """Fluid::add_particles / delete_particle_at_next_timestep (fluid.rs:71-150).  A faucet-like loop
    run twice: once growing / compacting the fluid on the device (salva_hip_add_particles / salva_hip_delete_particles),
    once through the host path (download, edit the arrays, re-upload everything with the survivors' velocity_changes).
    Both are the same stable compaction of the same data, so the trajectories must be bit-identical."""
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity
    from salva_amd import _lib as L

    fluid, shell = scenes.tank(10, 10, 10, R)
    fluid = scenes.jitter(fluid, 0.05 * R, seed=42)
    nozzle = scenes.cube_fluid_positions(3, 1, 3, R) + np.float32([0.0, 0.55, 0.0])

    def run(host_path):
        w = LiquidWorld(DFSPHSolver(), R, 2.0)
        f = Fluid(fluid, R, 1000.0)
        f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
        w.add_fluid(f)
        w.add_boundary(Boundary(shell))
        counts = []
        for k in range(12):
            w.step(DT, GRAVITY)
            if k % 2 == 1:
                if host_path:
                    _ = f.positions
                    f._dirty |= L.DIRTY_POSITIONS  # pretend the host edited something: forces the re-upload path
                f.add_particles(nozzle, np.tile(np.float32([0.0, -1.0, 0.0]), (len(nozzle), 1)))
            if k % 3 == 2:
                for i in range(0, f.num_particles(), 17):
                    f.delete_particle_at_next_timestep(i)
                if host_path:
                    _ = f.positions
                    f._dirty |= L.DIRTY_POSITIONS
            counts.append(f.num_particles() - f.num_deleted_particles())
        w.step(DT, GRAVITY)
        return f.positions.copy(), f.velocities.copy(), w.velocity_changes(f), counts

    pa, va, da, ca = run(False)
    pb, vb, db, cb = run(True)
    assert ca == cb and len(pa) == ca[-1] and ca[-1] != 1000
    assert np.array_equal(pa, pb) and np.array_equal(va, vb) and np.array_equal(da, db)
    assert np.isfinite(pa).all()


def test_cpp_mirror_faucet_example():
    """examples/faucet3.cpp: particles added by a nozzle and deleted below a kill plane through the C++ mirror, replayed on
    the device; the particle count must follow (added - deleted) and the jet must reach the floor."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "faucet3")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples")])
    out = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    last = out.stdout.strip().splitlines()[-1]
    m = re.match(r"step 300: (\d+) particles \(added (\d+), deleted (\d+)\), y in \[(-?[\d.]+), (-?[\d.]+)\]", last)
    assert m, last
    n, added, deleted, ymin, ymax = int(m[1]), int(m[2]), int(m[3]), float(m[4]), float(m[5])
    assert added == 30 * 25 and n == added - deleted and n > 100
    assert ymin < 0.1 and ymax <= 0.61


def test_particles_intersecting_aabb():
    """LiquidWorld::particles_intersecting_aabb (liquid_world.rs:210-243): distance to the box < particle radius, for fluid
    and boundary particles, before and after steps (the device answers from current positions)."""
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld

    fluid, shell = scenes.tank(8, 8, 8, R)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    f = w.add_fluid(Fluid(scenes.jitter(fluid, 0.1 * R, seed=1), R, 1000.0))
    f2 = w.add_fluid(Fluid(fluid + np.float32([1.0, 0.0, 0.0]), R, 800.0))
    b = w.add_boundary(Boundary(shell))
    mins, maxs = np.float32([-0.11, -0.3, -0.05]), np.float32([0.07, 0.02, 0.3])

    def expected():
        out = []
        for kind, objs in (("fluid", [f, f2]), ("boundary", [b])):
            for o in objs:
                p = np.asarray(o.positions, np.float32)
                d = np.maximum(np.maximum(mins - p, p - maxs), np.float32(0))
                hit = np.nonzero((d * d).sum(1, dtype=np.float32) < np.float32(R) * np.float32(R))[0]
                out += [(kind, o, int(i)) for i in hit]
        return out

    for steps in (0, 3):
        for _ in range(steps):
            w.step(DT, GRAVITY)
        got, ref = w.particles_intersecting_aabb(mins, maxs), expected()
        assert len(ref) > 20 and any(k == "boundary" for k, _, _ in ref)
        assert [(k, id(o), i) for k, o, i in got] == [(k, id(o), i) for k, o, i in ref]
    assert w.particles_intersecting_aabb([50, 50, 50], [51, 51, 51]) == []


@pytest.mark.parametrize("name", ["dfsph_tank", "iisph_akinci", "two_phase"])
def test_checkpoint_restart_continues_the_run(name, tmp_path):
    """a world rebuilt from `checkpoint()` (positions, velocities, volumes, velocity_changes, IISPH
    pressures, TimestepManager dt / inv_dt — through np.savez and back) continues like the original: same contacts and
    iteration counts, states equal up to f32 summation order (the restarted world sorts its cells from host order, the
    running one from last step's order, so neighbour sums add in a different sequence) — 100x below the parity tolerance.
    Leaving out any one piece (velocity_changes, pressures, dt) breaks this by orders of magnitude."""
    builder, _ = SCENES[name]
    a, fa, ba = builder().make_hip()
    for _ in range(5):
        a.step(DT, GRAVITY)
    path = tmp_path / "state.npz"
    np.savez(path, **a.checkpoint())
    b, fb, bb = builder().make_hip()
    b.restore(dict(np.load(path)))
    for k in range(5):
        sa, sb = a.step(DT, GRAVITY), b.step(DT, GRAVITY)
        assert (sa.n_divergence_iters, sa.n_pressure_iters, sa.ncontacts) == (sb.n_divergence_iters, sb.n_pressure_iters, sb.ncontacts)
    for x, y in zip(fa, fb):
        assert np.abs(x.positions - y.positions).max() < 1e-6 * R * 5
        vref = max(np.abs(x.velocities).max(), 2 * R / DT * 1e-2)
        assert np.abs(x.velocities - y.velocities).max() < 1e-5 * vref
        assert np.abs(a.velocity_changes(x) - b.velocity_changes(y)).max() < 1e-5 * vref
    # the restart needs every piece: without the timestep the first restarted step sees inv_dt = 0 in its divergence solve
    c, fc, _ = builder().make_hip()
    st = dict(np.load(path))
    st["timestep"] = np.zeros(2, np.float32)
    for k in range(len(fc)):
        st[f"fluid{k}_velocity_changes"] = np.zeros_like(st[f"fluid{k}_velocity_changes"])
    c.restore(st)
    for k in range(5):
        c.step(DT, GRAVITY)
    assert max(np.abs(x.positions - y.positions).max() for x, y in zip(fa, fc)) > 1e-4 * R


def test_borderline_pairs_are_contacts_exactly_when_the_reference_says_so():
    """Unjittered lattices are full of pairs at exactly d = h (two lattice spacings with smoothing factor 2), where
    `d^2 <= h^2` is decided by the last bit of ((dx*dx + dy*dy) + dz*dz).  The device must evaluate that expression with the
    reference's three roundings — an FMA-contracted sum (one rounding) classifies hundreds of these pairs differently.  Fluid
    lattice (fluid-fluid), floor (fluid-boundary) and two nearly coincident plates (boundary-boundary, the case that
    exposed a contracted dist2 in k_boundary_volumes): all counts must be equal, not close."""
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld
    from oracle import oracle as O

    fluid = scenes.cube_fluid_positions(9, 9, 9, R) + np.float32([0.013, 9 * R + R, 0.0071])  # off-grid origin, exact spacing
    floor = scenes.plane_lattice(16, 16, 0.0, R, -8 * 2 * R + R + 0.004, -8 * 2 * R + R, layers=1)
    plate_a = scenes.plane_lattice(6, 6, 0.0, R, np.float32(0.19809002), -3 * 2 * R + R, layers=1) + np.float32([0.0, 0.08, 0.0])
    plate_b = scenes.plane_lattice(6, 6, 0.0, R, np.float32(0.19811678), -3 * 2 * R + R, layers=1) + np.float32([0.0, 0.08, 0.0])
    o = O.OracleWorld(R, 2.0, O.DFSPH)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    k = o.add_fluid(fluid, 1000.0)
    h = w.add_fluid(Fluid(fluid, R, 1000.0))
    for b in (floor, plate_a, plate_b):
        o.add_boundary(b)
        w.add_boundary(Boundary(b))
    so, sh = o.step(DT, GRAVITY), w.step(DT, GRAVITY)
    nff, nfb = o.contact_counts(k, False), o.contact_counts(k, True)
    assert np.array_equal(w.contact_counts(h, False), nff) and np.array_equal(w.contact_counts(h, True), nfb)
    assert sh.ncontacts == so.ncontacts
    # the scene does contain what it is meant to test: many pairs within 1e-6 of h, roughly half of them contacts
    allb = np.concatenate([floor, plate_a, plate_b]).astype(np.float32)
    d = allb[:, None, :] - allb[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    hh = np.float32(R) * np.float32(2.0) * np.float32(2.0)
    near = np.abs(np.sqrt(d2.astype(np.float64)) - 0.1) < 1e-6
    fused = (d.astype(np.float64) ** 2).sum(-1).astype(np.float32) <= hh * hh
    assert near.sum() > 500 and (fused != (d2 <= hh * hh)).sum() > 20
