"""The reference's native intersector module under its own shape (``build.lib.intersector``).

The reference's ``--optix`` route imports a pybind module that is not in its repository
(``from build.lib import intersector``, mesh_utils.py:77,165; train_finetune.py:215-216) and uses three things of it:

* ``Intersector(vertices: float[F*9], max_hits: int, device: int)`` -- triangle soup, one triangle per 9 floats;
* ``.find_intersections(rays: float[R*6]) -> int[R*max_hits]`` -- triangle ids per ray, -1 padded;
* ``.update_vertices(vertices: float[F*9])`` -- same triangles, new positions (mesh_utils.py:83-84); the finetune
  loop also replaces the whole object after a vertex update (train_finetune.py:716-718).

This class is that shape over ``RayIntersector`` (8-wide BVH traversal in HIP, ``qf_bvh_*``).  Ids come nearest first.
What the absent module did with coincident hits is unknown (parity unpinned): ``min_separation``
defaults to 0 here -- every hit counts -- and can be set to the trimesh rule of the Embree route.

    from quadraturefields_amd import intersector          # instead of: from build.lib import intersector
"""
import numpy as np

from .mesh_io import TriMesh
from .mesh_utils import RayIntersector


class Intersector:
    def __init__(self, vertices, max_hits: int, device: int = 0, min_separation=0.0):
        tri = np.ascontiguousarray(np.asarray(vertices, dtype=np.float32).reshape(-1))
        if tri.size == 0 or tri.size % 9 != 0:
            raise ValueError("vertices: a flat float array of 9 values per triangle")
        n_tri = tri.size // 9
        mesh = TriMesh(tri.reshape(-1, 3), np.arange(n_tri * 3, dtype=np.int64).reshape(-1, 3))
        self.max_hits = int(max_hits)
        self.device = int(device)
        self.core = RayIntersector(mesh, max_hits=self.max_hits, device=f"cuda:{self.device}",
                                   min_separation=min_separation)

    @property
    def num_triangles(self) -> int:
        return int(self.core.mesh.faces.shape[0])

    def find_intersections(self, rays) -> np.ndarray:
        r = np.asarray(rays, dtype=np.float32).reshape(-1)
        if r.size % 6 != 0:
            raise ValueError("rays: a flat float array of 6 values (origin, direction) per ray")
        return self.core.find_intersections(r)

    def update_vertices(self, vertices) -> None:
        v = np.asarray(vertices, dtype=np.float32).reshape(-1)
        if v.size != self.num_triangles * 9:
            raise ValueError(f"update_vertices: expected {self.num_triangles * 9} floats, got {v.size}")
        self.core.update_intersector(v)
        self.core.mesh.vertices = v.reshape(-1, 3).astype(np.float64)
