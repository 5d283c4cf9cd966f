"""salva_amd — an MI355X-native (gfx950 HIP) implementation of salva3d's `LiquidWorld::step` hot path.

The product is `salva_amd/csrc/libsalva_hip.so` (C ABI in include/salva_hip.h); this package is the Python mirror of
the reference's host API used by the tests and the benchmark.  Importing the API objects does not load the
library; creating a `LiquidWorld` does, and fails loudly when it is missing or no HIP device is usable.
"""
from . import coupling, dist, scenes  # noqa: F401
from .world import (  # noqa: F401
    Akinci2013SurfaceTension,
    ArtificialViscosity,
    Boundary,
    Counters,
    CubicSplineKernel,
    DFSPHSolver,
    DFSPHViscosity,
    Fluid,
    He2014SurfaceTension,
    IISPHSolver,
    InteractionGroups,
    LiquidWorld,
    NonPressureForce,
    Poly6Kernel,
    SpikyKernel,
    ViscosityKernel,
    WCSPHSurfaceTension,
    XSPHViscosity,
)

__all__ = [
    "Akinci2013SurfaceTension", "ArtificialViscosity", "Boundary", "Counters", "CubicSplineKernel", "DFSPHSolver", "DFSPHViscosity", "Fluid", "He2014SurfaceTension", "IISPHSolver",
    "InteractionGroups", "LiquidWorld", "NonPressureForce", "Poly6Kernel", "SpikyKernel", "ViscosityKernel", "WCSPHSurfaceTension", "XSPHViscosity", "coupling", "dist", "scenes",
]


def kernel_source_sha() -> str:
    """sha256 (first 16 hex digits) over the sources of the kernels a single-GPU profile can contain: salva_amd/csrc/*.hip and
    *.h except the exchange transports (comm.h, comm.hip, comm_peer.hip) and the extern "C" shims (capi.hip), which hold no
    such kernel.  A committed rocprof summary records it so that bench.py can tell whether its PMC figures still describe the
    kernels in the tree."""
    import glob
    import hashlib
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    skip = {"comm.h", "comm.hip", "comm_peer.hip", "capi.hip"}
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(here, "*.hip")) + glob.glob(os.path.join(here, "*.h"))):
        if os.path.basename(f) in skip:
            continue
        hsh.update(os.path.basename(f).encode())
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]
