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
    DFSPHSolver,
    DFSPHViscosity,
    Fluid,
    He2014SurfaceTension,
    IISPHSolver,
    InteractionGroups,
    LiquidWorld,
    NonPressureForce,
    WCSPHSurfaceTension,
    XSPHViscosity,
)

__all__ = [
    "Akinci2013SurfaceTension", "ArtificialViscosity", "Boundary", "Counters", "DFSPHSolver", "DFSPHViscosity", "Fluid", "He2014SurfaceTension", "IISPHSolver",
    "InteractionGroups", "LiquidWorld", "NonPressureForce", "WCSPHSurfaceTension", "XSPHViscosity", "coupling", "dist", "scenes",
]
