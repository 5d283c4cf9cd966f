"""`DFSPHSolver<KernelDensity, KernelGradient>` / `IISPHSolver<..>` with non-default type parameters (dfsph_solver.rs:17-20,
iisph_solver.rs:17-20; src/kernel/{poly6,spiky,viscosity}_kernel.rs): the HIP path against the oracle on the golden scenes —
every solver pass and every NonPressureForce takes its contact weights from KernelDensity and its gradients from KernelGradient
(solver/helper.rs:9-63).  The oracle's restatement of the three kernels is itself checked against an independent numpy reading
and closed forms in tests/test_second_reading.py (CPU)."""
import ctypes as C

import numpy as np
import pytest

from golden_scenes import SCENES, run_oracle
from salva_amd import DFSPHSolver, LiquidWorld, _lib
from test_parity_gpu import compare, run_hip

pytestmark = pytest.mark.gpu

# (scene, KernelDensity, KernelGradient): Mueller's classic pairing on every family of passes, plus mixed pairs that leave one
# of the two parameters at its default
CASES = [
    ("dfsph_xsph_block", "poly6", "spiky"),
    ("dfsph_tank", "poly6", "spiky"),          # ArtificialViscosity + boundary forces
    ("iisph_akinci", "poly6", "spiky"),        # IISPH + Akinci2013 + XSPH
    ("two_phase", "cubic", "spiky"),
    ("dfsph_viscous", "poly6", "spiky"),       # DFSPHViscosity reads the gradients (<cubic, spiky> blows this scene up in the oracle too)
    ("surface_tension", "poly6", "cubic"),     # He2014 / WCSPH tensions read the weights
    ("dfsph_tank", "spiky", "viscosity"),
]


@pytest.mark.parametrize("name,kd,kg", CASES)
def test_other_kernels_against_oracle(name, kd, kg):
    builder, nsteps = SCENES[name]
    scene = builder()
    scene.kernels = (kd, kg)
    got, ref = run_hip(scene, nsteps), run_oracle(scene, nsteps)
    compare(got, ref, scene, nsteps, f"{name} <{kd}, {kg}> vs oracle")
    # ... and the choice does change the answer (a world that silently ran the cubic spline would pass nothing above, but say so)
    default = run_oracle(builder(), nsteps)
    assert np.abs(default["density_0"] - ref["density_0"]).max() > 1e-3 * np.abs(ref["density_0"]).max() or kd == "cubic"


def test_unknown_kernel_kind_is_refused():
    w = LiquidWorld(DFSPHSolver(), 0.025, 2.0)
    p = _lib.Params()
    w._L.salva_hip_default_params(C.byref(p))
    assert (p.kernel_density, p.kernel_gradient) == (0, 0)
    p.particle_radius, p.smoothing_factor, p.kernel_gradient = 0.025, 2.0, 7
    h = C.c_void_p()
    with pytest.raises(_lib.SalvaHipError):
        _lib.check(w._L.salva_hip_create(C.byref(p), C.byref(h)))
