"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/salva_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "salva_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(salva_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(hip_lib):
    from salva_amd import _lib

    declared = _declared_symbols()
    assert declared, "no declarations found"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} not exported by libsalva_hip.so"


def test_struct_layouts_match_header(hip_lib):
    from salva_amd import _lib

    assert C.sizeof(_lib.Params) == 4 * 18
    assert C.sizeof(_lib.ForceDesc) == 32
    assert C.sizeof(_lib.StepStats) == 4 * 4 + 8 * 2 + 4 * 8
    assert C.sizeof(_lib.RigidPose) == 4 * (3 + 4 + 3 + 3 + 3 + 2)  # SalvaHipRigidPose
    p = _lib.Params()
    hip_lib.salva_hip_default_params(C.byref(p))
    # defaults of DFSPHSolver::new (dfsph_solver.rs:54-70)
    assert (p.min_pressure_iter, p.max_pressure_iter, p.min_divergence_iter, p.max_divergence_iter) == (1, 50, 1, 50)
    assert abs(p.max_density_error - 0.05) < 1e-7 and abs(p.max_divergence_error - 0.1) < 1e-7
    assert hip_lib.salva_hip_version().startswith(b"salva_hip")


def test_no_cpu_fallback(hip_lib):
    """Without a HIP device the world cannot be created: the product never routes through a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from salva_amd import _lib

    p = _lib.Params()
    hip_lib.salva_hip_default_params(C.byref(p))
    h = C.c_void_p()
    rc = hip_lib.salva_hip_create(C.byref(p), C.byref(h))
    assert rc == _lib.E_HIP and not h.value
    assert b"no CPU fallback" in hip_lib.salva_hip_last_error()
    from salva_amd import DFSPHSolver, LiquidWorld

    with pytest.raises(_lib.SalvaHipError):
        LiquidWorld(DFSPHSolver(), 0.05, 2.0)


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "salva_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f in ("sph_math.h",), os.path.join(dirpath, f)


def test_rust_ffi_is_generated_from_the_header():
    """bindings/rust/salva3d-hip/src/ffi.rs = tools/gen_rust_ffi.py(include/salva_hip.h): every exported entry point is declared
    there, and the committed file is what the generator writes today."""
    import re
    import subprocess
    import sys

    from salva_amd import _lib

    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]) == 0, "run python tools/gen_rust_ffi.py"
    src = open(os.path.join(ROOT, "bindings", "rust", "salva3d-hip", "src", "ffi.rs")).read()
    assert set(re.findall(r"pub fn (salva_hip_\w+)", src)) == set(_lib.EXPORTED_SYMBOLS)
    for s in ("SalvaHipParams", "SalvaHipForceDesc", "SalvaHipStepStats", "SalvaHipCounters", "SalvaHipRigidPose", "SalvaHipShape"):
        assert f"pub struct {s} " in src


def test_cpp_mirror_header_compiles():
    """include/salva_hip.hpp (header-only C++ mirror of the salva3d API, incl. FluidsPipeline and the dynamic-sampling
    boundaries) is valid C++17 on its own: host compiler only, no HIP."""
    import subprocess
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".cpp", delete=False) as f:
        f.write('#include "%s"\nint main() { salva::FluidsPipeline* p = nullptr; salva::Boundary b = salva::Boundary::dynamic_ball(0.1f); (void)p; (void)b; return 0; }\n'
                % os.path.join(ROOT, "include", "salva_hip.hpp"))
        path = f.name
    try:
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-fsyntax-only", path])
    finally:
        os.unlink(path)


def test_c_header_is_plain_c():
    """include/salva_hip.h is the boundary a Rust `bindgen` / cgo / ctypes user parses as C, not C++: it compiles as strict C99
    with the host compiler (no HIP, no C++ constructs), and every declared entry point is callable from a C translation unit."""
    import re
    import subprocess
    import tempfile

    hdr = os.path.join(ROOT, "include", "salva_hip.h")
    names = sorted(set(re.findall(r"\b(salva_hip_[a-z0-9_]+)\s*\(", open(hdr).read())))
    assert len(names) >= 60
    # taking every function's address forces the declaration to be complete and consistent C
    body = "".join(f"    sink((void (*)(void)){n});\n" for n in names)
    src = ('#include "%s"\nstatic void sink(void (*f)(void)) { (void)f; }\nint main(void) {\n%s    return 0;\n}\n' % (hdr, body))
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        f.write(src)
        path = f.name
    try:
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Wstrict-prototypes", "-pedantic", "-Werror", "-fsyntax-only", path])
    finally:
        os.unlink(path)
