"""Minimal triangle-mesh container and PLY / OBJ readers (numpy only).

The reference loads its quadrature mesh with ``trimesh.load(path, force='mesh', process=False)``
(``examples/mesh_utils.py:193``) and only touches ``.vertices`` (float64), ``.faces`` (int64),
``.visual.uv`` and ``.export``.  trimesh is not available here, so this module provides that small surface.
"""
import struct
from types import SimpleNamespace
from typing import Optional

import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
    "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
    "double": "f8", "float64": "f8",
}


class TriMesh:
    """vertices float64 [V,3], faces int64 [F,3], optional per-vertex uv float64 [V,2] (as ``visual.uv``)."""

    def __init__(self, vertices, faces, uv: Optional[np.ndarray] = None):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.int64).reshape(-1, 3)
        self.visual = SimpleNamespace(uv=None if uv is None else np.ascontiguousarray(uv, dtype=np.float64))

    @property
    def triangles(self) -> np.ndarray:
        return self.vertices[self.faces]

    @property
    def face_normals(self) -> np.ndarray:
        t = self.triangles
        n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        return n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-300)

    def export(self, path: str) -> None:
        """Binary little-endian PLY (vertices + faces [+ s,t])."""
        has_uv = self.visual.uv is not None
        with open(path, "wb") as f:
            hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {len(self.vertices)}",
                   "property float x", "property float y", "property float z"]
            if has_uv:
                hdr += ["property float s", "property float t"]
            hdr += [f"element face {len(self.faces)}", "property list uchar int vertex_indices", "end_header"]
            f.write(("\n".join(hdr) + "\n").encode())
            cols = [self.vertices.astype("<f4")]
            if has_uv:
                cols.append(self.visual.uv.astype("<f4"))
            f.write(np.concatenate(cols, axis=1).tobytes())
            rec = np.empty(len(self.faces), dtype=[("n", "u1"), ("i", "<i4", 3)])
            rec["n"] = 3
            rec["i"] = self.faces
            f.write(rec.tobytes())


def _load_ply(path: str) -> TriMesh:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    elements[-1]["props"].append(("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        verts = faces = uv = None
        if fmt == "ascii":
            rows = f.read().decode("ascii", "replace").split("\n")
            pos = 0
            for el in elements:
                block = [r.split() for r in rows[pos:pos + el["count"]]]
                pos += el["count"]
                if el["name"] == "vertex":
                    names = [p[2] for p in el["props"]]
                    arr = np.array(block, dtype=np.float64)
                    verts = arr[:, [names.index(k) for k in ("x", "y", "z")]]
                    for a, b in (("s", "t"), ("u", "v"), ("texture_u", "texture_v")):
                        if a in names and b in names:
                            uv = arr[:, [names.index(a), names.index(b)]]
                elif el["name"] == "face":
                    tri = []
                    for r in block:
                        k = int(r[0])
                        idx = [int(v) for v in r[1:1 + k]]
                        tri += [[idx[0], idx[i], idx[i + 1]] for i in range(1, k - 1)]
                    faces = np.array(tri, dtype=np.int64)
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            for el in elements:
                if all(p[0] == "scalar" for p in el["props"]):
                    dt = np.dtype([(p[2], end + _PLY_TYPES[p[1]]) for p in el["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                    if el["name"] == "vertex":
                        verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
                        for a, b in (("s", "t"), ("u", "v"), ("texture_u", "texture_v")):
                            if a in arr.dtype.names and b in arr.dtype.names:
                                uv = np.stack([arr[a], arr[b]], axis=1).astype(np.float64)
                else:
                    tri = []
                    for _ in range(el["count"]):
                        lists = {}
                        for p in el["props"]:
                            if p[0] == "list":
                                cnt_t, idx_t = np.dtype(end + _PLY_TYPES[p[1]]), np.dtype(end + _PLY_TYPES[p[2]])
                                k = int(np.frombuffer(f.read(cnt_t.itemsize), dtype=cnt_t)[0])
                                lists[p[3]] = np.frombuffer(f.read(idx_t.itemsize * k), dtype=idx_t)
                            else:
                                f.read(np.dtype(_PLY_TYPES[p[1]]).itemsize)
                        if el["name"] == "face":
                            idx = lists.get("vertex_indices", lists.get("vertex_index"))
                            tri += [[idx[0], idx[i], idx[i + 1]] for i in range(1, len(idx) - 1)]
                    if el["name"] == "face":
                        faces = np.array(tri, dtype=np.int64)
        if verts is None or faces is None:
            raise ValueError("PLY has no vertex/face elements")
        return TriMesh(verts, faces, uv)


def _load_obj(path: str) -> TriMesh:
    v, vt, corners = [], [], []
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                v.append([float(x) for x in tok[1:4]])
            elif tok[0] == "vt":
                vt.append([float(x) for x in tok[1:3]])
            elif tok[0] == "f":
                c = []
                for item in tok[1:]:
                    parts = item.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    c.append((vi - 1 if vi > 0 else len(v) + vi, (ti - 1 if ti > 0 else len(vt) + ti) if ti else -1))
                corners += [[c[0], c[i], c[i + 1]] for i in range(1, len(c) - 1)]
    v = np.array(v, dtype=np.float64)
    if not vt:
        return TriMesh(v, np.array([[c[0] for c in tri] for tri in corners], dtype=np.int64))
    # one vertex per distinct (position, texcoord) pair, in order of first use (trimesh's unmerge)
    vt = np.array(vt, dtype=np.float64)
    remap, faces = {}, []
    for tri in corners:
        faces.append([remap.setdefault(c, len(remap)) for c in tri])
    keys = list(remap)
    verts = v[[k[0] for k in keys]]
    uv = np.array([vt[k[1]] if k[1] >= 0 else (0.0, 0.0) for k in keys])
    return TriMesh(verts, np.array(faces, dtype=np.int64), uv)


def load_mesh(path: str) -> TriMesh:
    """``trimesh.load(path, force='mesh', process=False)`` for .ply / .obj."""
    lower = path.lower()
    if lower.endswith(".ply"):
        return _load_ply(path)
    if lower.endswith(".obj"):
        return _load_obj(path)
    raise ValueError(f"unsupported mesh format: {path}")
