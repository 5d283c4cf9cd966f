#!/usr/bin/env python
"""tools/r06/hostile_host.py — how much of a step is the host's.

Runs the bench scene's 5 + 20 protocol (no torch, no extra legs) in child processes under different host placements and prints
one line per placement: ms per step over the 20 timed steps, the mean of the free-fall steps among them (divergence iterations
<= 2) and of the rest.  Placements:
  quiet        the child as the scheduler places it
  near         pinned to one core of the GPU's NUMA node
  far          pinned to one core of the NUMA node farthest from the GPU's (another socket when there is one)
  far+burners  the same, with busy-loop processes on the other cores of the CPU quota (what a loaded host looks like)
A step whose launches and read-backs sit on the GPU's critical path slows down from line to line; one that is enqueued ahead
does not.  No stress-ng in the image: the burners are this script's own busy loops."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    cpus = os.environ.get("HH_CPUS")
    if cpus:
        os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
    sys.path.insert(0, ROOT)
    import numpy as np
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes

    R, DT, G = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)
    side = int(os.environ.get("HH_SIDE", "100"))
    fluid, shell = scenes.tank(side, side, side, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    f = Fluid(fluid, R, 1000.0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(shell))
    for _ in range(5):
        w.step(DT, G)
    ms, it = [], []
    t0 = time.perf_counter()
    tp = t0
    for _ in range(int(os.environ.get("HH_STEPS", "20"))):
        st = w.step(DT, G)
        tn = time.perf_counter()
        ms.append((tn - tp) * 1e3)
        tp = tn
        it.append(int(st.n_divergence_iters))
    ms = np.asarray(ms)
    it = np.asarray(it)
    ff = ms[it <= 2]
    rest = ms[it > 2]
    print("HH " + json.dumps({"ms_per_step": float(ms.mean()), "free_fall_ms": float(ff.mean()) if len(ff) else None, "n_free_fall": int(len(ff)),
                              "rest_ms": float(rest.mean()) if len(rest) else None, "cpu": sorted(os.sched_getaffinity(0))[:4]}), flush=True)


def burner():
    os.sched_setaffinity(0, {int(os.environ["HH_BURN_CPU"])})
    x = 1.0
    while True:
        for _ in range(1000000):
            x = x * 1.0000001 + 1e-9


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def main():
    allowed = sorted(os.sched_getaffinity(0))
    nodes = {}
    base = "/sys/devices/system/node"
    if os.path.isdir(base):
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                try:
                    nodes[int(d[4:])] = [c for c in cpulist(open(f"{base}/{d}/cpulist").read()) if c in allowed]
                except OSError:
                    pass
    gpu_node = None
    for card in sorted(os.listdir("/sys/class/drm")) if os.path.isdir("/sys/class/drm") else []:
        p = f"/sys/class/drm/{card}/device/numa_node"
        if card.startswith("renderD") and os.path.exists(p):
            try:
                gpu_node = int(open(p).read())
                break
            except (OSError, ValueError):
                pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        pass
    print(f"host: {len(allowed)} allowed CPUs, NUMA nodes {{{', '.join(f'{k}: {len(v)} cpus' for k, v in nodes.items())}}}, GPU on node {gpu_node}, cpu quota {quota}", flush=True)
    near = (nodes.get(gpu_node) or allowed)[:1] if gpu_node is not None and gpu_node >= 0 else allowed[:1]
    far_node = None
    if len(nodes) > 1 and gpu_node is not None and gpu_node >= 0:
        # the node with the largest distance from the GPU's
        try:
            dist = [int(x) for x in open(f"{base}/node{gpu_node}/distance").read().split()]
            far_node = max((k for k in nodes if nodes[k] and k != gpu_node), key=lambda k: dist[k] if k < len(dist) else 0)
        except (OSError, ValueError):
            far_node = max(k for k in nodes if nodes[k] and k != gpu_node)
    far = nodes[far_node][-1:] if far_node is not None else allowed[-1:]
    nburn = int(os.environ.get("HH_BURNERS", str(max(int((quota or 8)) - 1, 1))))
    env0 = dict(os.environ, HH_ROLE="child")
    reps = int(os.environ.get("HH_REPS", "2"))

    def run(tag, cpus, burners=0):
        procs = []
        others = [c for c in allowed if c not in cpus]
        for k in range(burners):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, HH_ROLE="burn", HH_BURN_CPU=str(others[(k * 7919) % len(others)]))))
        try:
            for _ in range(reps):
                env = dict(env0)
                if cpus:
                    env["HH_CPUS"] = ",".join(map(str, cpus))
                out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
                line = [l for l in out.stdout.splitlines() if l.startswith("HH ")]
                print(f"{tag:12s} {line[-1][3:] if line else 'FAILED: ' + out.stderr[-400:]}", flush=True)
        finally:
            for p in procs:
                p.kill()
            for p in procs:
                p.wait()

    run("quiet", [])
    run("near", near)
    run("far", far)
    run("far+burners", far, nburn)
    run("quiet+burners", [], nburn)


if __name__ == "__main__":
    role = os.environ.get("HH_ROLE")
    if role == "child":
        child()
    elif role == "burn":
        burner()
    else:
        main()
