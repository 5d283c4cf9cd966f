"""`LiquidWorld: Send + Sync` (the reference's compile-time test, /root/reference/src/liquid_world.rs:283-287) at the C boundary:
every entry point that takes a world holds the world's lock (salva_amd/csrc/capi.hip), so the `&self` methods of the Rust wrapper —
queries, read-backs, counters — may run on other threads while one thread steps."""
import os
import subprocess
import threading
import time

import numpy as np
import pytest

from parity import DT, GRAVITY, Scene
from salva_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_world_three_reader_threads_and_a_stepping_thread_cpp(tmp_path):
    exe = tmp_path / "two_threads"
    lib = os.path.join(ROOT, "salva_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "two_threads.cpp"),
                           f"-L{lib}", "-lsalva_hip", f"-Wl,-rpath,{lib}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 bad" in r.stdout, r.stdout


def test_python_threads_query_while_the_world_steps():
    """ctypes releases the GIL around every call into the library: two Python threads really are inside it at once."""
    s = Scene(0.025, 2.0, "dfsph")
    fluid, shell = scenes.tank(16, 16, 16, 0.025)
    s.add_fluid(scenes.jitter(fluid, 0.1 * 0.025, seed=1), None, 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(shell)
    w, (fl,), _ = s.make_hip()
    w.step(DT, GRAVITY)
    stop, errors, counts = threading.Event(), [], []

    def reader():
        try:
            while not stop.is_set():
                c = w.contact_counts(fl)
                assert c.min() >= 1 and c.max() < 200
                counts.append(int(c.sum()))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=reader)
    t.start()
    totals = []
    for _ in range(60):
        st = w.step(DT, GRAVITY)
        totals.append(int(w.contact_counts(fl).sum()))
        time.sleep(0.0005)  # (the lock is not fair: give the reader a chance between two steps)
    stop.set()
    t.join()
    assert not errors, errors
    assert len(counts) > 3 and set(counts) <= set(totals) | {counts[0]}  # every answer is some completed step's
