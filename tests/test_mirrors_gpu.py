"""The two host mirrors of the salva3d API — Python (salva_amd/world.py) and C++ (include/salva_hip.hpp) — replay the same
seeded script of operations (add fluid / particles, deferred deletions, remove fluid / boundary, host velocity edits,
steps with varying dt) against the same library: the final states must be identical bit for bit.  Whatever differs is a
bug in how one of the mirrors stages, defers or replays the edits (both agree with the oracle in tests/test_fuzz_gpu.py
only as far as the Python one is concerned)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from parity import DT, GRAVITY
from salva_amd import Boundary, DFSPHSolver, Fluid, IISPHSolver, LiquidWorld, XSPHViscosity, scenes

pytestmark = pytest.mark.gpu

R = 0.025
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fmt(a):
    return " ".join(repr(float(x)) for x in np.asarray(a, np.float32).ravel())


def _script(solver, seed, nops=50):
    """The operations as data: (text for the C++ helper, list of tuples for the Python replay)."""
    rng = np.random.default_rng(seed)
    lines, ops = [f"{solver} {float(np.float32(R))!r}"], []
    counts, nb = [], 0  # particles per fluid slot as the reference would report them (uncompacted until a step)
    pending = []

    def block(nx, ny, nz, origin):
        p = scenes.jitter(scenes.cube_fluid_positions(nx, ny, nz, R), 0.05 * R, seed=int(rng.integers(1 << 30)))
        return (p + np.float32(origin)).astype(np.float32)

    def add_fluid(x):
        pos = block(4, 4, 4, [x, 0.25, 0.0])
        vel = scenes.random_velocities(len(pos), 0.1, seed=int(rng.integers(1 << 30))).astype(np.float32)
        density = float(rng.choice([800.0, 1000.0]))
        lines.append(f"ADD_FLUID {density!r} {len(pos)}\n" + "\n".join(_fmt(np.concatenate([p, v])) for p, v in zip(pos, vel)))
        ops.append(("add_fluid", density, pos, vel))
        counts.append(len(pos)); pending.append(set())

    def add_boundary(pos):
        nonlocal nb
        lines.append(f"ADD_BOUNDARY {len(pos)}\n" + "\n".join(_fmt(p) for p in pos))
        ops.append(("add_boundary", pos))
        nb += 1

    add_boundary(scenes.plane_lattice(30, 12, 0.0, R, -6 * 2 * R + R, -6 * 2 * R + R, layers=1))
    add_fluid(0.0)
    add_fluid(0.5)
    next_x = 1.0
    for _ in range(nops):
        op = rng.choice(["step", "step", "step", "add_particles", "delete", "remove_fluid", "add_fluid", "shift", "add_boundary", "remove_boundary"])
        if op == "step":
            dt = float(rng.choice([DT, DT, DT / 2, 0.0]))
            lines.append(f"STEP {dt!r} {GRAVITY[0]!r} {GRAVITY[1]!r} {GRAVITY[2]!r}")
            ops.append(("step", dt))
            for k in range(len(counts)):
                counts[k] -= len(pending[k]); pending[k] = set()
        elif op == "add_particles" and counts:
            k = int(rng.integers(len(counts)))
            pos = block(2, 2, 2, [0.5 * k + float(rng.uniform(-0.05, 0.05)), 0.7, 0.0])
            has_vel = bool(rng.random() < 0.5)
            vel = np.tile(np.float32([0.0, -0.5, 0.0]), (len(pos), 1))
            rows = [_fmt(np.concatenate([p, v]) if has_vel else p) for p, v in zip(pos, vel)]
            lines.append(f"ADD_PARTICLES {k} {len(pos)} {int(has_vel)}\n" + "\n".join(rows))
            ops.append(("add_particles", k, pos, vel if has_vel else None))
            counts[k] += len(pos)
        elif op == "delete" and counts:
            k = int(rng.integers(len(counts)))
            idx = [int(i) for i in rng.choice(counts[k], size=min(5, counts[k]), replace=False)]
            lines.append(f"DELETE {k} {len(idx)} " + " ".join(str(i) for i in idx))
            ops.append(("delete", k, idx))
            pending[k] |= set(idx)
        elif op == "remove_fluid" and len(counts) >= 2:
            k = int(rng.integers(len(counts)))
            lines.append(f"REMOVE_FLUID {k}")
            ops.append(("remove_fluid", k))
            counts[k], pending[k] = counts[-1], pending[-1]
            counts.pop(); pending.pop()
        elif op == "add_fluid" and len(counts) < 4:
            add_fluid(next_x)
            next_x += 0.5
        elif op == "shift" and counts:
            k = int(rng.integers(len(counts)))
            lines.append(f"SHIFT_VELOCITIES {k} 0.2")
            ops.append(("shift", k))
        elif op == "add_boundary" and nb < 3:
            add_boundary(scenes.plane_lattice(6, 6, 0.0, R, float(rng.uniform(-0.2, 1.0)), -3 * 2 * R + R, layers=1) + np.float32([0.0, 0.08, 0.0]))
        elif op == "remove_boundary" and nb >= 2:
            k = int(rng.integers(1, nb))
            lines.append(f"REMOVE_BOUNDARY {k}")
            ops.append(("remove_boundary", k))
            nb -= 1
    lines.append(f"STEP {DT!r} {GRAVITY[0]!r} {GRAVITY[1]!r} {GRAVITY[2]!r}")
    ops.append(("step", DT))
    return "\n".join(lines) + "\n", ops


def _replay_python(solver, ops):
    w = LiquidWorld(DFSPHSolver() if solver == "dfsph" else IISPHSolver(), R, 2.0)
    fluids, bounds = [], []
    for op in ops:
        if op[0] == "add_fluid":
            f = Fluid(op[2], R, op[1])
            f.velocities = op[3]
            f.nonpressure_forces.append(XSPHViscosity(0.5, 0.2))
            fluids.append(w.add_fluid(f))
        elif op[0] == "add_boundary":
            bounds.append(w.add_boundary(Boundary(op[1])))
        elif op[0] == "step":
            w.step(op[1], GRAVITY)
        elif op[0] == "add_particles":
            fluids[op[1]].add_particles(op[2], op[3])
        elif op[0] == "delete":
            for i in op[2]:
                fluids[op[1]].delete_particle_at_next_timestep(i)
        elif op[0] == "remove_fluid":
            w.remove_fluid(fluids[op[1]])
            fluids[op[1]] = fluids[-1]
            fluids.pop()
        elif op[0] == "shift":
            v = fluids[op[1]].velocities.copy()
            v[:, 0] += np.float32(0.2)
            fluids[op[1]].velocities = v
        elif op[0] == "remove_boundary":
            w.remove_boundary(bounds[op[1]])
            bounds[op[1]] = bounds[-1]
            bounds.pop()
    return [(f.positions.copy(), f.velocities.copy()) for f in fluids]


def _read_dump(path):
    raw = open(path, "rb").read()
    nf, = struct.unpack_from("<Q", raw, 0)
    off, out = 8, []
    for _ in range(nf):
        n, = struct.unpack_from("<Q", raw, off)
        off += 8
        pos = np.frombuffer(raw, np.float32, 3 * n, off).reshape(n, 3); off += 12 * n
        vel = np.frombuffer(raw, np.float32, 3 * n, off).reshape(n, 3); off += 12 * n
        out.append((pos, vel))
    return out


@pytest.mark.parametrize("solver,seed", [("dfsph", 11), ("iisph", 12), ("dfsph", 13), ("iisph", 14)])
def test_cpp_and_python_mirrors_replay_identically(solver, seed, tmp_path):
    exe = tmp_path / "mirror_replay"
    lib = os.path.join(ROOT, "salva_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "mirror_replay.cpp"),
                           f"-L{lib}", "-lsalva_hip", f"-Wl,-rpath,{lib}"])
    text, ops = _script(solver, seed)
    script, dump = tmp_path / "script.txt", tmp_path / "dump.bin"
    script.write_text(text)
    r = subprocess.run([str(exe), str(script), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    cpp = _read_dump(dump)
    py = _replay_python(solver, ops)
    assert len(cpp) == len(py)
    for k, ((cp, cv), (pp, pv)) in enumerate(zip(cpp, py)):
        assert cp.shape == pp.shape, (k, cp.shape, pp.shape)
        assert np.array_equal(cp, pp) and np.array_equal(cv, pv), (k, float(np.abs(cp - pp).max()), float(np.abs(cv - pv).max()))
