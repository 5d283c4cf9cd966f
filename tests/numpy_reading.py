"""A SECOND, independent reading of salva's DFSPH and IISPH steps — plain numpy, float64, dense O(N^2) neighbourhoods, at most
a few hundred particles — written straight from the Rust, not from oracle/salva_oracle.cpp:

  liquid_world.rs:62-158 (step order), timestep_manager.rs:36-95 (dt / inv_dt lag), geometry/contacts.rs:254-400 (contact
  criterion d^2 <= h^2, self contacts included, directed lists), kernel/cubic_spline_kernel.rs:12-79 + kernel/kernel.rs:13-24,
  solver/helper.rs:9-65, object/fluid.rs:105-115 (volume = 0.8 (2r)^3), solver/pressure/dfsph_solver.rs:72-708,
  solver/pressure/iisph_solver.rs:92-711, solver/viscosity/xsph_viscosity.rs:31-95, solver/viscosity/artificial_viscosity.rs:41-135,
  solver/surface_tension/akinci2013_surface_tension.rs:44-203, he2014_surface_tension.rs:41-182, wcsph_surface_tension.rs:30-92, solver/viscosity/dfsph_viscosity.rs:38-322.

Covered (tests/test_second_reading.py): both pressure solvers pass by pass; the four SPH kernels; XSPH, artificial viscosity,
Akinci2013 / He2014 / WCSPH surface tension and DFSPHViscosity (including the divergence of its loop); boundary volumes and the
reaction forces handed to boundary particles; several fluids and boundaries with InteractionGroups; particles added and deleted
between steps.  Not covered: DynamicContactSampling (its own numpy geometry in tests/test_oracle.py) and the rigid-body pose /
wrench arithmetic (tests/test_oracle.py::test_coupling_pose_velocity_and_wrench).

Purpose: the oracle and the HIP kernels were written by the same hand from the same source, so a shared
misreading passes every GPU-vs-oracle test.  This file shares no code and no data structure with either (no grid, no contact
lists, no per-pass loops over contacts: every pass is a masked dense matrix expression), and tests/test_second_reading.py
compares every intermediate field of the oracle's f64 build with it, step by step.  It does not pin anything to salva itself —
no Rust toolchain here — but a transcription error would have to be made twice, independently, in two different formulations.

Test infrastructure only."""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)  # Real::default_epsilon() of the reference's f32 build (kernel.rs:19)


def spline_w(r, h):
    """cubic_spline_kernel.rs:12-33 (dim3)."""
    normalizer = 8.0 / (np.pi * h * h * h)
    q = r / h
    q2 = q * q
    inner = 1.0 + (q2 * q - q2) * 6.0
    outer = (1.0 - q) ** 3 * 2.0
    return normalizer * np.where(q <= 0.5, inner, np.where(q <= 1.0, outer, 0.0))


def spline_dw(r, h):
    """cubic_spline_kernel.rs:55-79 (dim3): zero for q > 1 and for q <= 1e-5."""
    normalizer = 8.0 / (np.pi * h * h * h)
    q = r / h
    inner = (q * 3.0 - 2.0) * q * 6.0
    one_q = 1.0 - q
    outer = -one_q * one_q * 6.0
    rhs = np.where((q > 1.0) | (q <= 1.0e-5), 0.0, np.where(q <= 0.5, inner, outer))
    return normalizer * rhs / h


def poly6_w(r, h):
    """poly6_kernel.rs:9-21 (dim3)."""
    return np.where(r <= h, 315.0 / 64.0 / (np.pi * h ** 9) * (h * h - r * r) ** 3, 0.0)


def poly6_dw(r, h):
    """poly6_kernel.rs:24-38."""
    return np.where(r <= h, 315.0 / 64.0 / (np.pi * h ** 9) * (h * h - r * r) ** 2 * r * -6.0, 0.0)


def spiky_w(r, h):
    """spiky_kernel.rs:9-21."""
    return np.where(r <= h, 15.0 / (np.pi * h ** 6) * (h - r) ** 3, 0.0)


def spiky_dw(r, h):
    """spiky_kernel.rs:23-36."""
    return np.where(r <= h, -15.0 / (np.pi * h ** 6) * (h - r) ** 2 * 3.0, 0.0)


def visc_w(r, h):
    """viscosity_kernel.rs:9-27: zero at r = 0 as well."""
    rs = np.where(r > 0.0, r, 1.0)
    val = 15.0 / (2.0 * np.pi * h ** 3) * (rs * rs / (h * h) * (1.0 - rs / (2.0 * h)) + h / (2.0 * rs) - 1.0)
    return np.where((r > 0.0) & (r <= h), val, 0.0)


def visc_dw(r, h):
    """viscosity_kernel.rs:29-49."""
    rs = np.where(r > 0.0, r, 1.0)
    val = 15.0 / (2.0 * np.pi * h ** 3) * (-3.0 * rs * rs / (2.0 * h ** 3) + 2.0 * rs / (h * h) - h / (2.0 * rs * rs))
    return np.where((r > 0.0) & (r <= h), val, 0.0)


# the solvers' KernelDensity / KernelGradient type parameters (dfsph_solver.rs:17-20) by name
KERNELS = {"cubic": (spline_w, spline_dw), "poly6": (poly6_w, poly6_dw), "spiky": (spiky_w, spiky_dw), "viscosity": (visc_w, visc_dw)}


def pair_tables(xa, xb, h, kernel_density="cubic", kernel_gradient="cubic", f32_contacts=False):
    """For every (a, b): contact mask (contacts.rs: distance_squared <= h*h), weight (helper.rs: KernelDensity::points_apply) and
    gradient (KernelGradient::points_apply_diff1 = apply_diff(pa - pb): direction * dW/dr, zero when the norm is <= eps,
    kernel.rs:18-24).  `f32_contacts`: decide the contact in the arithmetic of the reference's f32 build (positions rounded to
    f32, (dx*dx + dy*dy) + dz*dz and h*h in f32) — lattice boundaries hold many pairs at exactly d = h, which f32 and f64
    arithmetic put on different sides; everything else stays f64."""
    d = xa[:, None, :] - xb[None, :, :]
    r2 = (d * d).sum(axis=2)
    if f32_contacts:
        d32 = xa.astype(np.float32)[:, None, :] - xb.astype(np.float32)[None, :, :]
        r2_32 = (d32[:, :, 0] * d32[:, :, 0] + d32[:, :, 1] * d32[:, :, 1]) + d32[:, :, 2] * d32[:, :, 2]
        mask = r2_32 <= np.float32(h) * np.float32(h)
    else:
        mask = r2 <= h * h
    r = np.sqrt(r2)
    w = np.where(mask, KERNELS[kernel_density][0](r, h), 0.0)
    safe = np.where(r > EPS32, r, 1.0)
    g = np.where((mask & (r > EPS32))[:, :, None], d / safe[:, :, None] * KERNELS[kernel_gradient][1](r, h)[:, :, None], 0.0)
    return mask, w, g


class DenseWorld:
    """Any number of fluids and boundaries (set_fluid / set_boundary: exactly one of each; add_fluid / add_boundary: more), each
    with its InteractionGroups, an optional XSPHViscosity and any list of the other built-in NonPressureForces per fluid.
    All fluid particles live in one set of arrays (`model` = the fluid a particle belongs to), all boundary particles in another:
    a pass over "the contacts of particle i" is a masked row of a dense pair table whatever the objects are."""

    def __init__(self, particle_radius, smoothing_factor=2.0, solver="dfsph", kernel_density="cubic", kernel_gradient="cubic",
                 f32_contacts=False):
        self.kernels = (kernel_density, kernel_gradient)
        self.f32_contacts = bool(f32_contacts)  # pair_tables: the contact criterion in f32 arithmetic (against f32 implementations)
        self.r = float(particle_radius)
        self.h = float(particle_radius) * float(smoothing_factor) * 2.0  # liquid_world.rs:44
        self.solver = solver
        # pub fields of DFSPHSolver::new / IISPHSolver::new
        self.min_pressure_iter, self.max_pressure_iter, self.max_density_error = 1, 50, 0.05
        self.min_divergence_iter, self.max_divergence_iter, self.max_divergence_error = 1, 50, 0.1
        self.min_neighbors = 20
        self.omega = 0.5
        self.dt = 0.0      # TimestepManager::new: dt = inv_dt = 0 until the first advance()
        self.inv_dt = 0.0
        self.xsph = {}     # fluid index -> (fluid coefficient, boundary coefficient)
        self.forces = {}   # fluid index -> fluid.nonpressure_forces after the XSPH entry, in order: (kind, params...)
        self.trace = {}
        self._reset_fluids()
        self._reset_boundaries()

    def _reset_fluids(self):
        self.x = np.zeros((0, 3)); self.v = np.zeros((0, 3)); self.a = np.zeros((0, 3)); self.vol = np.zeros(0)
        self.model = np.zeros(0, np.int64)       # which fluid a particle belongs to
        self.rho0 = np.zeros(0)                  # its fluid's density0
        self.groups = np.zeros((0, 2), np.uint64)  # its fluid's (memberships, filter)
        self.nfluids = 0
        self.deleted = np.zeros(0, bool)           # delete_particle_at_next_timestep marks
        self.dv = np.zeros((0, 3))   # solver.velocity_changes: persists across steps
        self.p = np.zeros(0)         # IISPH pressures: persist across steps

    def _reset_boundaries(self):
        self.xb = np.zeros((0, 3)); self.vb = np.zeros((0, 3))
        # boundary.forces (boundary.rs:59-67, `Some` buffer): accumulated by every apply_force until the caller clears it
        self.bforce = np.zeros((0, 3))
        self.bmodel = np.zeros(0, np.int64)
        self.bgroups = np.zeros((0, 2), np.uint64)
        self.nboundaries = 0

    def add_fluid(self, positions, density0=1000.0, velocities=None, memberships=1, filter=0xFFFFFFFF):
        x = np.asarray(positions, np.float64).reshape(-1, 3)
        n = len(x)
        v = np.zeros((n, 3)) if velocities is None else np.asarray(velocities, np.float64).reshape(-1, 3)
        self.x = np.concatenate([self.x, x]); self.v = np.concatenate([self.v, v]); self.a = np.concatenate([self.a, np.zeros((n, 3))])
        self.vol = np.concatenate([self.vol, np.full(n, self.r ** 3 * 8.0 * 0.8)])  # fluid.rs:105-115
        self.model = np.concatenate([self.model, np.full(n, self.nfluids, np.int64)])
        self.rho0 = np.concatenate([self.rho0, np.full(n, float(density0))])
        self.groups = np.concatenate([self.groups, np.tile(np.array([[memberships, filter]], np.uint64), (n, 1))])
        self.dv = np.concatenate([self.dv, np.zeros((n, 3))]); self.p = np.concatenate([self.p, np.zeros(n)])
        self.deleted = np.concatenate([self.deleted, np.zeros(n, bool)])
        self.nfluids += 1
        return self.nfluids - 1

    def add_boundary(self, positions, memberships=1, filter=0xFFFFFFFF):
        xb = np.asarray(positions, np.float64).reshape(-1, 3)
        self.xb = np.concatenate([self.xb, xb]); self.vb = np.concatenate([self.vb, np.zeros_like(xb)])
        self.bforce = np.concatenate([self.bforce, np.zeros_like(xb)])
        self.bmodel = np.concatenate([self.bmodel, np.full(len(xb), self.nboundaries, np.int64)])
        self.bgroups = np.concatenate([self.bgroups, np.tile(np.array([[memberships, filter]], np.uint64), (len(xb), 1))])
        self.nboundaries += 1
        return self.nboundaries - 1

    def add_particles(self, fluid, positions, velocities=None):
        """Fluid::add_particles (fluid.rs:126-150): appended with default volume and zero acceleration; the solver's per-particle
        buffers grow with zeros at the next step's init_with_fluids (dfsph_solver.rs:526-547: `resize(n, zero)`) — a new particle
        starts with no velocity change and no IISPH pressure."""
        x = np.asarray(positions, np.float64).reshape(-1, 3)
        n = len(x)
        rows = np.nonzero(self.model == fluid)[0]
        if len(rows) == 0:
            raise ValueError("add_particles: the fluid needs at least one particle to copy its density0 / groups from")
        v = np.zeros((n, 3)) if velocities is None else np.asarray(velocities, np.float64).reshape(-1, 3)
        self.x = np.concatenate([self.x, x]); self.v = np.concatenate([self.v, v]); self.a = np.concatenate([self.a, np.zeros((n, 3))])
        self.vol = np.concatenate([self.vol, np.full(n, self.r ** 3 * 8.0 * 0.8)])
        self.model = np.concatenate([self.model, np.full(n, fluid, np.int64)])
        self.rho0 = np.concatenate([self.rho0, np.full(n, self.rho0[rows[0]])])
        self.groups = np.concatenate([self.groups, np.tile(self.groups[rows[0]][None, :], (n, 1))])
        self.dv = np.concatenate([self.dv, np.zeros((n, 3))]); self.p = np.concatenate([self.p, np.zeros(n)])
        self.deleted = np.concatenate([self.deleted, np.zeros(n, bool)])

    def delete_particle_at_next_timestep(self, fluid, i):
        """fluid.rs:71-86: marked now, gone at the top of the next step — from the fluid (apply_particles_removal, :88-98) and
        from the solver's buffers (filter_from_mask with the same mask, dfsph_solver.rs:549-560, iisph_solver.rs:502-536); the
        survivors keep their order."""
        self.deleted[np.nonzero(self.model == fluid)[0][i]] = True

    def _apply_removal(self):
        if self.deleted.any():
            keep = ~self.deleted
            for name in ("x", "v", "a", "vol", "model", "rho0", "groups", "dv", "p"):
                setattr(self, name, getattr(self, name)[keep])
            self.deleted = np.zeros(int(keep.sum()), bool)

    def set_fluid_volumes(self, fluid, volumes):
        """`fluid.volumes` is a pub field (fluid.rs:24): scenes may override the default."""
        self.vol[self.model == fluid] = np.asarray(volumes, np.float64)

    def set_fluid(self, positions, density0=1000.0, velocities=None):
        self._reset_fluids()
        self.add_fluid(positions, density0, velocities)

    def set_boundary(self, positions):
        self._reset_boundaries()
        self.add_boundary(positions)

    def fluid_rows(self, f):
        return self.model == f

    def set_xsph(self, fluid_coeff, boundary_coeff, fluid=0):
        self.xsph[fluid] = (float(fluid_coeff), float(boundary_coeff))

    @staticmethod
    def _allowed(model_a, groups_a, model_b, groups_b, same_kind):
        """interaction_groups.rs:64-72 `test`, as the contact search applies it (contacts.rs:277, :316, :348, :359): pairs inside
        one object always interact; pairs of different objects only when each side's memberships meet the other's filter."""
        test = ((groups_a[:, None, 0] & groups_b[None, :, 1]) != 0) & ((groups_b[None, :, 0] & groups_a[:, None, 1]) != 0)
        if same_kind:
            return test | (model_a[:, None] == model_b[None, :])
        return test

    def _error(self, per_particle):
        """`max_error.max(err / nparts)` over the fluids (dfsph_solver.rs:151-159 and its siblings)."""
        worst = 0.0
        for f in range(self.nfluids):
            rows = self.model == f
            if rows.any():
                worst = max(worst, float(per_particle[rows].sum() / rows.sum()))
        return worst

    def add_force(self, kind, *params, fluid=0):
        """kind in "artificial" (fluid, boundary, alpha = 1, beta = 0, speed_of_sound = 10), "akinci2013" (tension, adhesion),
        "he2014" (fluid tension, boundary tension), "dfsph_viscosity" (coefficient, min_iter, max_iter, max_error), "wcsph" (fluid tension; the boundary arm of the reference indexes the
        boundary set with fluid contacts, wcsph_surface_tension.rs:69-88, and is left at 0)."""
        self.forces.setdefault(fluid, []).append((kind,) + tuple(float(p) for p in params))

    # ------------------------------------------------------------------------------------------------------------
    def step(self, dt, gravity=(0.0, -9.81, 0.0)):
        g = np.asarray(gravity, np.float64)
        self._apply_removal()   # liquid_world.rs:76-80: before the substep loop, so also when the loop does not run
        if dt <= EPS32:  # timestep_manager.is_done() before the first substep
            return
        h = self.h
        m = self.vol * self.rho0                              # Fluid::particle_mass
        ok_ff = self._allowed(self.model, self.groups, self.model, self.groups, True)
        ok_fb = self._allowed(self.model, self.groups, self.bmodel, self.bgroups, False)
        ok_bb = self._allowed(self.bmodel, self.bgroups, self.bmodel, self.bgroups, True)
        self.ff, self.wff, self.gff = pair_tables(self.x, self.x, h, *self.kernels, f32_contacts=self.f32_contacts)
        self.fb, self.wfb, self.gfb = pair_tables(self.x, self.xb, h, *self.kernels, f32_contacts=self.f32_contacts)
        bb, wbb, _ = pair_tables(self.xb, self.xb, h, *self.kernels, f32_contacts=self.f32_contacts)
        self.ff, self.wff, self.gff = self.ff & ok_ff, np.where(ok_ff, self.wff, 0.0), np.where(ok_ff[:, :, None], self.gff, 0.0)
        self.fb, self.wfb, self.gfb = self.fb & ok_fb, np.where(ok_fb, self.wfb, 0.0), np.where(ok_fb[:, :, None], self.gfb, 0.0)
        bb, wbb = bb & ok_bb, np.where(ok_bb, wbb, 0.0)
        # compute_boundary_volumes (dfsph_solver.rs:72-96)
        self.volb = 1.0 / wbb.sum(axis=1) if len(self.xb) else np.zeros(0)
        mb = self.volb[None, :] * self.rho0[:, None]          # V_b * fluid_i.density0: per (fluid particle, boundary particle)
        # compute_densities (:628-665)
        self.rho = self.wff @ m + (self.wfb * mb).sum(axis=1)
        self.ncontacts = int(self.ff.sum() + self.fb.sum() + bb.sum())
        if self.solver == "dfsph":
            self._dfsph(dt, g, m, mb)
        else:
            self._iisph(dt, g, m, mb)

    # ------------------------------------------------------------------------------------------------------------
    def _forces(self, m, mb):
        """predict_advection after `acceleration += gravity`: XSPHViscosity::solve (xsph_viscosity.rs:31-95) with the
        timestep's CURRENT inv_dt — the previous step's, advance() comes afterwards."""
        full = (self.ff, self.wff, self.gff, self.fb, self.wfb, self.gfb)
        for f in range(self.nfluids):
            if f not in self.xsph and not self.forces.get(f):
                continue
            # a fluid's forces see the contacts of ITS particles, and of those only the ones inside the fluid
            # (`if c.i_model == c.j_model` in every fluid arm) plus all boundary contacts: the same tables with the other rows and
            # the other fluids' columns blanked
            rows = self.model == f
            inside = rows[:, None] & rows[None, :]
            self._rows, self._rho0f = rows, float(self.rho0[rows][0]) if rows.any() else 0.0
            self.ff, self.wff, self.gff = full[0] & inside, np.where(inside, full[1], 0.0), np.where(inside[:, :, None], full[2], 0.0)
            self.fb, self.wfb, self.gfb = full[3] & rows[:, None], np.where(rows[:, None], full[4], 0.0), np.where(rows[:, None, None], full[5], 0.0)
            if f in self.xsph:
                cf, cb = self.xsph[f]
                add = np.zeros_like(self.a)
                if cf != 0.0:
                    coef = cf * self.wff * (m / self.rho)[None, :]     # c.weight * volumes[j] * density0 / densities[j]
                    add += (coef[:, :, None] * (self.v[None, :, :] - self.v[:, None, :])).sum(axis=1) * self.inv_dt
                if cb != 0.0:
                    coef = cb * self.wfb * mb / self.rho[:, None]
                    delta = coef[:, :, None] * (self.vb[None, :, :] - self.v[:, None, :])
                    add += delta.sum(axis=1) * self.inv_dt
                    self.bforce += (delta * (-m * self.inv_dt)[:, None, None]).sum(axis=0)   # xsph_viscosity.rs:87-88
                self.a += add
            for force in self.forces.get(f, []):
                self.a += getattr(self, "_force_" + force[0])(m, mb, *force[1:])
        self.ff, self.wff, self.gff, self.fb, self.wfb, self.gfb = full

    # ---- the other built-in NonPressureForces, as dense pair expressions.  Self contacts are in the lists (contacts.rs) and
    # contribute nothing: r_ij = 0 makes v.r = 0 (not < 0), the gradient 0 and the unit direction undefined (-> zero vector).
    def _pairs(self, other_x):
        d = self.x[:, None, :] - other_x[None, :, :]
        r2 = (d * d).sum(axis=2)
        return d, r2, np.sqrt(r2)

    def _force_artificial(self, m, mb, cf, cb, alpha=1.0, beta=0.0, speed_of_sound=10.0):
        """artificial_viscosity.rs:62-131 (Monaghan 1992): only approaching pairs (v_ij . r_ij < 0)."""
        h, rho = self.h, self.rho
        acc = np.zeros_like(self.a)
        eta2 = h * h * 0.01
        if cf != 0.0:
            d, r2, _ = self._pairs(self.x)
            vr = (d * (self.v[:, None, :] - self.v[None, :, :])).sum(axis=2)
            mu = h * vr / (r2 + eta2)
            pi_ij = cf * (speed_of_sound * alpha * mu - beta * mu * mu) * (m[None, :] / ((rho[:, None] + rho[None, :]) * 0.5))
            acc += (self.gff * np.where(self.ff & (vr < 0.0), pi_ij, 0.0)[:, :, None]).sum(axis=1)
        if cb != 0.0 and len(self.xb):
            d, r2, _ = self._pairs(self.xb)
            vr = (d * (self.v[:, None, :] - self.vb[None, :, :])).sum(axis=2)
            mu = h * vr / (r2 + eta2)
            pi_ib = cb * (speed_of_sound * alpha * mu - beta * mu * mu) * (mb / rho[:, None])
            acc += (self.gfb * np.where(self.fb & (vr < 0.0), pi_ib, 0.0)[:, :, None]).sum(axis=1)
        return acc

    def _force_akinci2013(self, m, mb, tension, adhesion):
        """akinci2013_surface_tension.rs: normals (:44-71), cohesion / adhesion splines (:74-117), forces (:146-198)."""
        h, rho, rho0 = self.h, self.rho, self._rho0f
        acc = np.zeros_like(self.a)
        if tension != 0.0:
            normals = (self.gff * (m / rho)[None, :, None]).sum(axis=1) * h
            d, _, r = self._pairs(self.x)
            inner = 2.0 * (h - r) ** 3 * r ** 3 - h ** 6 / 64.0
            outer = (h - r) ** 3 * r ** 3
            coh = 32.0 / (np.pi * h ** 9) * np.where(r <= h / 2.0, inner, np.where(r <= h, outer, 0.0))
            unit = np.where((r > EPS32)[:, :, None], d / np.where(r > EPS32, r, 1.0)[:, :, None], 0.0)  # Unit::try_new_and_get(dpos, eps)
            cohesion = unit * (coh * (-tension) * m[None, :])[:, :, None]
            curvature = (normals[:, None, :] - normals[None, :, :]) * (-tension)
            kij = 2.0 * rho0 / (rho[:, None] + rho[None, :])
            acc += ((curvature + cohesion) * np.where(self.ff, kij, 0.0)[:, :, None]).sum(axis=1)
        if adhesion != 0.0 and len(self.xb):
            d, _, r = self._pairs(self.xb)
            inside = (r > h / 2.0) & (r <= h)
            poly = np.maximum(-4.0 * r * r / h + 6.0 * r - 2.0 * h, 0.0)
            adh = np.where(inside, 0.007 / h ** 3.25 * poly ** 0.25, 0.0)
            unit = np.where((r > EPS32)[:, :, None], d / np.where(r > EPS32, r, 1.0)[:, :, None], 0.0)
            adhesion_acc = unit * np.where(self.fb, adh * adhesion * mb, 0.0)[:, :, None]
            acc -= adhesion_acc.sum(axis=1)
            self.bforce += (adhesion_acc * m[:, None, None]).sum(axis=0)                 # :187-188
        return acc

    def _force_he2014(self, m, mb, tension, boundary_tension):
        """he2014_surface_tension.rs: colour field (:41-78, boundary volumes enter unweighted by a density), squared norm of its
        normalised gradient (:80-108), forces (:137-176)."""
        rho, rho0, rows = self.rho, self._rho0f, self._rows
        volf = m / rho
        colors = self.wff @ volf + (self.wfb @ self.volb if len(self.xb) else 0.0)
        colors = np.where(rows, colors, 1.0)                    # (rows of other fluids: blank, kept finite)
        gradc = (self.gff * (colors * volf)[None, :, None]).sum(axis=1) / colors[:, None]
        gsq = (gradc * gradc).sum(axis=1)
        acc = np.zeros_like(self.a)
        if tension != 0.0:
            f = volf[:, None] * volf[None, :] * (gsq[:, None] + gsq[None, :]) / 2.0
            acc += (self.gff * f[:, :, None]).sum(axis=1) * (tension / (2.0 * m))[:, None]
        if boundary_tension != 0.0 and len(self.xb):
            f = self.gfb * (volf[:, None] * (mb / rho0) * gsq[:, None] * boundary_tension * 0.25)[:, :, None]
            acc += f.sum(axis=1) / m[:, None]
            self.bforce -= f.sum(axis=0)                                                 # :176
        return acc

    def _force_dfsph_viscosity(self, m, mb, coefficient, min_iter=1.0, max_iter=50.0, max_error=0.01):
        """dfsph_viscosity.rs: betas (:130-196, with the preconditioner that scales the FIRST THREE columns only, :166-168 and
        :191-194), target strain rates (:198-246, once), then the error / acceleration loop (:297-322).  `timestep.dt()` and
        `inv_dt()` are the previous step's here too (predict_advection runs before advance)."""
        rho = self.rho
        n = int(self._rows.sum())                               # fluid.num_particles()
        g = self.gff                                            # zero outside the contact mask and on the diagonal
        z = np.zeros_like(g[:, :, 0])
        # compute_gradient_matrix: 6 x 3 per pair
        G = np.stack([np.stack([2.0 * g[:, :, 0], z, z], axis=-1), np.stack([z, 2.0 * g[:, :, 1], z], axis=-1),
                      np.stack([z, z, 2.0 * g[:, :, 2]], axis=-1), np.stack([g[:, :, 1], g[:, :, 0], z], axis=-1),
                      np.stack([g[:, :, 2], z, g[:, :, 0]], axis=-1), np.stack([z, g[:, :, 2], g[:, :, 1]], axis=-1)], axis=-2)
        half = (m[None, :] / (2.0 * rho[:, None]))              # particle_mass(j) / (2 densities[i])
        Gi = G * half[:, :, None, None]
        squared = np.einsum("ijab,ijcb->iac", Gi, Gi) / rho[:, None, None]
        gsum = Gi.sum(axis=1)
        den = squared + np.einsum("iab,icb->iac", gsum, gsum) / rho[:, None, None]
        diag = np.einsum("iaa->ia", den)
        inv_diag = np.where(np.abs(diag) < 1.0e-6, 1.0, 1.0 / np.where(np.abs(diag) < 1.0e-6, 1.0, diag))
        den = den.copy()
        den[:, :, :3] *= inv_diag[:, :, None]                   # column_mut(c).component_mul_assign(&inv_diag), c < SPATIAL_DIM
        det = np.linalg.det(den)
        ok = np.abs(det) >= 1.0e-6
        betas = np.zeros_like(den)
        betas[ok] = np.linalg.inv(den[ok])
        betas[:, :, :3] *= inv_diag[:, None, :3]                # column c scaled by inv_diag[c], c < SPATIAL_DIM
        self.visc_betas = betas

        # The two pair sums of the loop as matrix-vector products (the loop runs up to 50 times per step):
        #   rate_i  = sum_j Gi_ij (v_j - v_i)                    = A v - (sum_j Gi_ij) v_i,       A: (6N x 3N)
        #   acc_i   = m_i inv_dt sum_j G_ij^T (u_i + u_j) m_j / 2 = m_i inv_dt (S_i^T u_i + B u),  B: (3N x 6N), S_i = sum_j G_ij m_j / 2
        nall = len(self.x)
        A = Gi.transpose(0, 2, 1, 3).reshape(6 * nall, 3 * nall)
        Hm = G * (m[None, :] / 2.0)[:, :, None, None]
        B = Hm.transpose(0, 3, 1, 2).reshape(3 * nall, 6 * nall)
        S = Hm.sum(axis=1)

        def rates(acc):
            v = self.v + acc * self.dt
            return (A @ v.ravel()).reshape(nall, 6) - np.einsum("iab,ib->ia", gsum, v)

        a = self.a.copy()
        target = rates(a) * (1.0 - coefficient)
        self.visc_iters = 0
        for i in range(int(max_iter)):
            error = rates(a) - target
            avg = float(np.abs(error).sum() / 6.0 / n) if n else 0.0
            self.visc_err = avg
            if avg <= max_error and i >= int(min_iter):
                break
            u = np.einsum("iab,ib->ia", betas, error) / (rho * rho)[:, None]
            a = a + (np.einsum("iab,ia->ib", S, u) + (B @ u.ravel()).reshape(nall, 3)) * (m * self.inv_dt)[:, None]
            self.visc_iters += 1
        return a - self.a

    def _force_wcsph(self, m, mb, tension):
        """wcsph_surface_tension.rs:46-66: a weight-proportional attraction along r_ij."""
        d, _, _ = self._pairs(self.x)
        return (d * (self.wff * (-tension) * m[None, :] / m[:, None])[:, :, None]).sum(axis=1)

    def _advance(self, dt):
        self.dt = dt
        self.inv_dt = 0.0 if dt == 0.0 else 1.0 / dt

    def _dfsph(self, dt, g, m, mb):
        rho0 = self.rho0                                        # per particle: its fluid's density0
        ncon = self.ff.sum(axis=1) + self.fb.sum(axis=1)
        # compute_alphas (:165-216)
        gi = self.gff * m[None, :, None]
        gbi = self.gfb * mb[:, :, None]
        sq = (gi * gi).sum(axis=(1, 2)) + (gbi * gbi).sum(axis=(1, 2))
        gs = gi.sum(axis=1) + gbi.sum(axis=1)
        den = sq + (gs * gs).sum(axis=1)
        self.alpha = np.where(den <= 1.0e-5, 0.0, 1.0 / np.where(den <= 1.0e-5, 1.0, den))
        # divergence_solve (:466-503) — timestep.inv_dt() is still the previous step's
        self.n_div = 0
        for i in range(self.max_divergence_iter):
            w = self.v + self.dv
            dvel = w[:, None, :] - w[None, :, :]
            div = ((dvel * self.gff).sum(axis=2) * m[None, :]).sum(axis=1) + ((w[:, None, :] * self.gfb).sum(axis=2) * mb).sum(axis=1)
            div = np.where(ncon < self.min_neighbors, 0.0, np.maximum(div, 0.0))
            self.div = div
            err = self._error(div / rho0)
            self.div_err = err
            if err <= self.max_divergence_error * self.inv_dt * 0.01 and i >= self.min_divergence_iter:
                break
            k = div * self.alpha
            kij = k[:, None] + k[None, :]
            delta_b = self.gfb * (-k[:, None] * mb)[:, :, None]
            self.dv = self.dv + (self.gff * (-(kij) * m[None, :])[:, :, None]).sum(axis=1) + delta_b.sum(axis=1)
            if len(self.xb):
                self.bforce += (delta_b * (-self.inv_dt * m)[:, None, None]).sum(axis=0)  # :403-405, the lagged inv_dt
            self.n_div += 1
        # update_velocities + zero (:689-691)
        self.v = self.v + self.dv
        self.dv = np.zeros_like(self.dv)
        # predict_advection (:565-604)
        self.a = self.a + g[None, :]
        self._forces(m, mb)
        self._advance(dt)
        # integrate_and_clear_accelerations (:505-519)
        self.dv = self.dv + self.a * self.dt
        self.a = np.zeros_like(self.a)
        # pressure_solve (:432-464)
        self.n_press = 0
        for i in range(self.max_pressure_iter):
            w = self.v + self.dv
            dvel = w[:, None, :] - w[None, :, :]
            delta = ((dvel * self.gff).sum(axis=2) * m[None, :]).sum(axis=1)
            delta += (((w[:, None, :] - self.vb[None, :, :]) * self.gfb).sum(axis=2) * mb).sum(axis=1)
            self.rho_pred = self.rho + delta * self.dt
            e = np.where(self.rho_pred < rho0, 0.0, self.rho_pred / rho0 - 1.0)
            err = self._error(e)
            self.press_err = err
            if err <= self.max_density_error and i >= self.min_pressure_iter:
                break
            k = (self.rho_pred - rho0) * self.alpha
            kp = np.maximum(k, 0.0)
            kij = kp[:, None] + kp[None, :]
            self.dv = self.dv - (self.gff * (kij * m[None, :] * self.inv_dt)[:, :, None]).sum(axis=1)
            coeff = np.where(k > 0.0, k, 0.0)[:, None] * mb * self.inv_dt
            delta_b = self.gfb * coeff[:, :, None]
            self.dv = self.dv - delta_b.sum(axis=1)
            if len(self.xb):
                self.bforce += (delta_b * (self.inv_dt * m)[:, None, None]).sum(axis=0)   # :264-272
            self.n_press += 1
        # update_positions (:411-420): velocities are NOT updated here
        self.x = self.x + (self.v + self.dv) * self.dt

    def _iisph(self, dt, g, m, mb):
        rho0 = self.rho0
        # step (:643-711)
        self.a = self.a + g[None, :]
        self._forces(m, mb)
        self._advance(dt)
        self.dv = self.dv + self.a * self.dt
        self.a = np.zeros_like(self.a)
        dt2 = self.dt * self.dt
        rho = self.rho
        # compute_dii (:144-186)
        fac = -dt2 / (rho * rho)
        self.dii = (self.gff * m[None, :, None]).sum(axis=1) * fac[:, None] + (self.gfb * mb[:, :, None]).sum(axis=1) * fac[:, None]
        self.p = self.p * 0.5
        # compute_predicted_densities (:92-142)
        w = self.v + self.dv
        dvel = w[:, None, :] - w[None, :, :]
        delta = ((dvel * self.gff).sum(axis=2) * m[None, :]).sum(axis=1)
        delta += (((w[:, None, :] - self.vb[None, :, :]) * self.gfb).sum(axis=2) * mb).sum(axis=1)
        self.rho_pred = rho + delta * self.dt
        # compute_aii (:188-233): a_ii = sum_j m_j (d_ii - d_ji) . grad W_ij, d_ji = grad W_ij dt^2 m_i / rho_i^2
        fji = dt2 * m / (rho * rho)
        dji = self.gff * fji[:, None, None]
        self.aii = (((self.dii[:, None, :] - dji) * self.gff).sum(axis=2) * m[None, :]).sum(axis=1)
        djib = self.gfb * fji[:, None, None]
        self.aii += (((self.dii[:, None, :] - djib) * self.gfb).sum(axis=2) * mb).sum(axis=1)
        # pressure_solve (:422-456)
        self.n_press = 0
        for i in range(self.max_pressure_iter):
            # compute_dij_pjl (:235-268)
            coef = -m * self.p / (rho * rho)
            self.dijpj = (self.gff * coef[None, :, None]).sum(axis=1) * dt2
            # compute_next_pressures (:270-353)
            factor = (self.dijpj[:, None, :] - self.dii[None, :, :] * self.p[None, :, None]
                      - (self.dijpj[None, :, :] - dji * self.p[:, None, None]))
            s = ((factor * self.gff).sum(axis=2) * m[None, :]).sum(axis=1)
            s += ((self.dijpj[:, None, :] * self.gfb).sum(axis=2) * mb).sum(axis=1)
            active = np.abs(self.aii) > 1.0e-9
            aii_safe = np.where(active, self.aii, 1.0)
            pn = (1.0 - self.omega) * self.p + self.omega * (rho0 - self.rho_pred - s) / aii_safe
            pos = active & (pn > 0.0)
            e = np.where(pos, (-s - self.aii * pn) / rho0, 0.0)
            self.p = np.where(pos, pn, 0.0)
            err = self._error(e)
            self.press_err = err
            self.n_press += 1
            if err <= self.max_density_error and i >= self.min_pressure_iter:
                break
        # compute_velocity_changes (:355-404)
        pr = self.p / (rho * rho)
        cij = self.dt * m[None, :] * (pr[:, None] + pr[None, :])
        self.dv = self.dv - (self.gff * cij[:, :, None]).sum(axis=1)
        acc_b = self.gfb * (mb * pr[:, None])[:, :, None]
        self.dv = self.dv - acc_b.sum(axis=1) * self.dt
        if len(self.xb):
            self.bforce += (acc_b * m[:, None, None]).sum(axis=0)                         # :394-400
        # update_velocities_and_positions (:406-420) + zero
        self.v = self.v + self.dv
        self.x = self.x + self.v * self.dt
        self.dv_last = self.dv.copy()
        self.dv = np.zeros_like(self.dv)
