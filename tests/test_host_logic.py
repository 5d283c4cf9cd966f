"""CPU-only checks of the host-side mirror code that sits above the C ABI (no device needed): the rigid-body stand-in
of salva_amd.coupling (rapier's formulas), the contact / kernel helpers handed to user-defined forces, and the
bookkeeping flags that keep the per-step host work O(1)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from salva_amd import Fluid, NonPressureForce, _lib
from salva_amd.coupling import RigidBody, quat_mul, quat_rotate
from salva_amd.world import ParticlesContacts, _cubic_spline

R = 0.025


def _rot_matrix(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_quaternion_helpers_match_rotation_matrices():
    rng = np.random.default_rng(1)
    for _ in range(20):
        q = rng.normal(size=4)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        v = rng.normal(size=3).astype(np.float32)
        np.testing.assert_allclose(quat_rotate(q, v), _rot_matrix(q) @ v, atol=2e-6)
        p = rng.normal(size=4)
        p = (p / np.linalg.norm(p)).astype(np.float32)
        np.testing.assert_allclose(_rot_matrix(quat_mul(q, p)), _rot_matrix(q) @ _rot_matrix(p), atol=2e-6)


def test_rigid_body_impulses_follow_rapier_formulas():
    b = RigidBody(translation=np.float32([1, 2, 3]), local_com=np.float32([0.1, 0, 0]), mass=2.0,
                  principal_inertia=np.float32([1.0, 2.0, 4.0]))
    ang = 0.7
    b.rotation = np.float32([0, np.sin(ang / 2), 0, np.cos(ang / 2)])
    com = b.center_of_mass()
    np.testing.assert_allclose(com, _rot_matrix(b.rotation) @ b.local_com + b.translation, atol=1e-6)
    b.linvel = np.float32([0.5, 0, 0])
    b.angvel = np.float32([0, 0, 2.0])
    pt = np.float32([1.5, 2.5, 3.0])
    np.testing.assert_allclose(b.velocity_at_point(pt), b.linvel + np.cross(b.angvel, pt - com), atol=1e-6)
    # apply_impulse_at_point(J, pt) = apply_impulse(J) + apply_torque_impulse((pt - com) x J): world inertia R I R^T
    J = np.float32([0.0, 1.0, 0.5])
    w0, v0 = b.angvel.copy(), b.linvel.copy()
    b.apply_impulse(J)
    b.apply_torque_impulse(np.cross(pt - com, J))
    Rm = _rot_matrix(b.rotation)
    inv_inertia_world = Rm @ np.diag(1.0 / b.principal_inertia) @ Rm.T
    np.testing.assert_allclose(b.linvel, v0 + J / 2.0, atol=1e-6)
    np.testing.assert_allclose(b.angvel, w0 + inv_inertia_world @ np.cross(pt - com, J), atol=1e-5)
    # a non-dynamic body ignores impulses and gravity
    k = RigidBody(dynamic=False, linvel=np.float32([0, 1, 0]))
    k.apply_impulse(J)
    k.integrate(0.1)
    np.testing.assert_allclose(k.linvel, [0, 1, 0])
    np.testing.assert_allclose(k.translation, [0, 0.1, 0], atol=1e-7)
    pose = b.pose()
    assert pose.has_body == 1 and pose.is_dynamic == 1 and abs(pose.rotation[3] - np.cos(ang / 2)) < 1e-7


def test_mirror_spline_and_contacts_match_the_oracle_kernel(oracle_lib):
    h = np.float32(4 * R)
    for r in np.linspace(0.0, 1.1 * h, 45, dtype=np.float32):
        w, dw = _cubic_spline(r, h)
        assert abs(float(w) - O.kernel_w(float(r), float(h))) <= 2e-6 * O.kernel_w(0.0, float(h))
        assert abs(float(dw) - O.kernel_dw(float(r), float(h))) <= 2e-6 * abs(O.kernel_dw(0.3 * float(h), float(h)))
    # CSR -> Contact objects: weight / gradient of the pair, zero gradient for the self contact
    pos = np.float32([[0, 0, 0], [0.05, 0, 0], [0, 0.08, 0]])
    pc = ParticlesContacts(0, np.uint64([0, 3, 5, 7]), np.uint32([0] * 7), np.uint32([0, 1, 2, 0, 1, 0, 2]), pos, lambda m: pos, float(h))
    assert pc.len() == 3
    c = pc.particle_contacts(0)
    assert [k.j for k in c] == [0, 1, 2] and not c[0].gradient.any()
    assert abs(c[0].weight - O.kernel_w(0.0, float(h))) < 1e-3
    assert abs(c[1].weight - O.kernel_w(0.05, float(h))) < 1e-3 and c[1].gradient[0] > 0 and c[1].gradient[1] == 0
    g = pc.particle_contacts(1)[0]  # contact (1 -> 0): gradient flips
    np.testing.assert_allclose(g.gradient, -c[1].gradient, rtol=1e-6)


def test_custom_force_descriptor_and_host_flags():
    class Field(NonPressureForce):
        def solve(self, *a):
            pass

    assert Field()._desc().kind == _lib.FORCE_CUSTOM
    with pytest.raises(NotImplementedError):
        NonPressureForce()._desc()
    # per-step host work stays O(1) unless the user touched the O(N) arrays
    f = Fluid(np.zeros((10, 3), np.float32), R, 1000.0)
    assert not f._acc_touched and not f._maybe_deleted
    f.accelerations[3] = 1.0
    assert f._acc_touched
    f.delete_particle_at_next_timestep(4)
    assert f._maybe_deleted and f.num_deleted_particles() == 1


def test_the_pinning_kit_still_names_things_the_reference_has():
    """bench/rust_ref has never met cargo: at least every `use salva3d::...` item and every method its main.rs calls must exist in the
    reference tree (/root/reference/src/lib.rs:86-118 and below).  Skipped where the reference is absent (the GPU boxes)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference tree here")
    r = subprocess.run([sys.executable, os.path.join(root, "bench", "rust_ref", "check_against_reference.py"), "/root/reference"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
