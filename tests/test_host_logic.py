"""CPU: host-side logic added in round 4 that needs no GPU -- the mesh depth complexity that seeds the intersector's overflow
policy (mesh_utils.mesh_depth_complexity; Cauchy-Crofton), the per-triangle random UV charts of SURVEY.md 8(d), the
banded tile count of the windowed coherent layout (qf_banded_tile_count: a host-only entry of the C ABI)."""
import numpy as np
import pytest

from quadraturefields_amd import synthetic
from quadraturefields_amd.mesh_utils import mesh_depth_complexity


def test_depth_complexity_is_the_mean_crossing_count_of_a_line_through_the_box():
    v, f = synthetic.icosphere(4)
    # one unit sphere in its own bounding box: 2 * 4 pi / (6 * 2^2) = pi / 3 crossings per line through the BOX
    # (every line through the sphere crosses it twice; a fraction pi/6 of the box-crossing line measure meets it)
    one = mesh_depth_complexity(v, f)
    assert abs(one - np.pi / 3) < 0.01
    # n concentric shells of radii r_i inside the outermost one's box: sum r_i^2 / r_max^2 as many
    radii = np.array([0.25, 0.5, 0.75, 1.0])
    vs = np.concatenate([v * r for r in radii])
    fs = np.concatenate([f + i * len(v) for i in range(len(radii))])
    many = mesh_depth_complexity(vs, fs)
    assert abs(many / one - float((radii ** 2).sum())) < 1e-6
    # the bench scenes: the 12-shell Lego stand-in stays in the plain mode at K = 25, configs[2]'s 36 shells start dense
    lego = synthetic.shell_mesh(n_shells=12, subdivisions=3)
    dense = synthetic.shell_mesh(n_shells=36, subdivisions=3)
    assert mesh_depth_complexity(lego.vertices, lego.faces) < 0.5 * 25 <= mesh_depth_complexity(dense.vertices, dense.faces)
    assert mesh_depth_complexity(np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)) == 0.0
    flat = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float64)           # a single flat triangle: box area 2 * 1
    assert mesh_depth_complexity(flat, np.array([[0, 1, 2]])) == pytest.approx(2 * 0.5 / 2.0)


def test_per_triangle_charts_unshare_the_vertices_and_keep_the_triangles():
    mesh = synthetic.shell_mesh(n_shells=2, subdivisions=2)
    m2, uv = synthetic.per_triangle_charts(mesh, 512, seed=3)
    f = mesh.faces.shape[0]
    assert m2.faces.shape == (f, 3) and np.array_equal(m2.faces.reshape(-1), np.arange(3 * f))
    assert np.array_equal(m2.vertices, mesh.vertices[mesh.faces.reshape(-1)])      # same triangles, same ids
    assert uv.shape == (3 * f, 2) and uv.dtype == np.float32 and uv.min() >= 0 and uv.max() <= 511
    tri = uv.reshape(f, 3, 2)
    legs = np.stack([tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]], axis=1)
    assert np.allclose(legs[:, 0], [4.0, 0.0], atol=1e-3) and np.allclose(legs[:, 1], [0.0, 4.0], atol=1e-3)
    # the charts are scattered: neighbouring triangles land far apart
    assert np.abs(np.diff(tri[:, 0, 0])).mean() > 50


def test_banded_tile_count(lib):
    from quadraturefields_amd import _C
    n = lambda w, h, b: int(_C.lib().qf_banded_tile_count(w, h, b))                 # noqa: E731
    assert n(800, 800, 0) == 100 * 100 == n(800, 800, 800) == n(800, 800, 4000)       # one band = the plain grid
    assert n(800, 800, 200) == 4 * 100 * 25                                         # the reference's windows at 800 wide
    assert n(1600, 1600, 100) == 16 * 200 * 13                                      # ... at up_sample 2: 100 rows = 12.5 tiles
    assert n(96, 96, 20) == 12 * (4 * 3 + 2)                                        # four bands of 3 tile rows + one of 2
    assert n(17, 43, 8) == 3 * 6 and n(0, 5, 0) == -1 and n(5, 5, -1) == -1
