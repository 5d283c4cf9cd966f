import numpy as np

from salva_amd import scenes


def test_cube_fluid_matches_helper_rs():
    """examples3d/helper.rs:4-20: i-major / k-minor, spacing 2r, centred on the origin."""
    r = np.float32(0.05)
    p = scenes.cube_fluid_positions(3, 2, 4, float(r))
    assert p.shape == (24, 3) and p.dtype == np.float32
    assert np.allclose(p.mean(axis=0), 0.0, atol=1e-6)
    assert np.allclose(p[1] - p[0], [0, 0, 2 * r])          # k is the fastest index
    assert np.allclose(p[4] - p[0], [0, 2 * r, 0])
    assert np.allclose(p[8] - p[0], [2 * r, 0, 0])
    assert np.allclose(p[0], [r - 3 * r, r - 2 * r, r - 4 * r])


def test_lcg_is_the_numerical_recipes_sequence():
    x, ref = 42, []
    for _ in range(1000):
        x = (1664525 * x + 1013904223) & 0xFFFFFFFF
        ref.append((x >> 8) / float(1 << 24))
    got = scenes.lcg_uniform(1000, 42)
    assert np.array_equal(got, np.asarray(ref, np.float32))
    v = scenes.random_velocities(10, 0.1)
    assert v.shape == (10, 3) and np.abs(v).max() <= 0.1


def test_box_shell_and_tank():
    sh = scenes.box_shell([0, 0, 0], [1, 1, 1], 0.05, faces="y")
    assert len(sh) == 11 * 11 and np.allclose(sh[:, 1], 0.0)
    fluid, shell = scenes.tank(4, 4, 4, 0.05)
    assert len(fluid) == 64
    assert shell[:, 1].min() < fluid[:, 1].min()
    assert len(np.unique(np.round((shell - shell.min(axis=0)) / 0.1).astype(int), axis=0)) == len(shell)


def test_committed_scene_files_match_the_python_builders(tmp_path):
    """bench/rust_ref/scenes/*.scene (what the real salva3d is run on, tests/golden/compare_rust_dump.py) are the golden scenes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tests", "golden", "export_scenes.py"), "--out", str(tmp_path)],
                          stdout=subprocess.DEVNULL)
    names = sorted(os.listdir(tmp_path))
    assert len(names) == 7
    for n in names:
        assert open(os.path.join(tmp_path, n), "rb").read() == open(os.path.join(root, "bench", "rust_ref", "scenes", n), "rb").read(), n


def test_ball_ray_sampling_is_a_lattice_shell():
    """ball_surface_ray_sample (ray_sampling.rs:27-88 with the ball's closed-form ray cast): points of the sampler's lattice
    (spacing 2 r, origin aabb.mins - 2 r + r), entry impacts rounded inwards (ceil / floor), so every sample lies inside the
    sphere within one lattice diagonal of its surface, and the set has the symmetries of the lattice."""
    r, R = 0.0125, 0.15
    pts = scenes.ball_surface_ray_sample(R, r)
    assert len(pts) == len({tuple(p) for p in pts.tolist()}) == 336
    d = np.linalg.norm(pts.astype(np.float64), axis=1)
    assert d.max() <= R + 1e-6 and d.min() > R - 2 * r * np.sqrt(3)
    origin = np.float32(-R - 2 * r + r)
    idx = (pts - origin) / np.float32(2 * r)
    assert np.abs(idx - np.round(idx)).max() < 1e-4
    as_set = {tuple(np.round(p / r).astype(int)) for p in pts}
    assert {(-a, b, c) for a, b, c in as_set} == as_set and {(b, c, a) for a, b, c in as_set} == as_set
