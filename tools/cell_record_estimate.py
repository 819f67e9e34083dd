"""Sizing of the "cell record" layout DESIGN.md section 7 proposes for the finest hash-grid levels (CPU only, numpy).

Quadrature points only fall on mesh triangles, so only the grid cells a triangle crosses are ever interpolated.  For
the bench scene this counts, per level, the distinct cells that points sampled on the surface land in, the distinct
4^3 / 8^3 bricks holding them, and what a 64-byte record per cell (8 corners x 2 fp32 features) plus a dense
brick-pointer grid would occupy -- against the T-row hashed table the level has today.

    python tools/cell_record_estimate.py --points 8000000
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=8_000_000)
    ap.add_argument("--log2-t", type=int, default=19)
    ap.add_argument("--shells", type=int, default=12)
    ap.add_argument("--subdiv", type=int, default=6)
    args = ap.parse_args()
    from quadraturefields_amd import _C, synthetic
    mesh = synthetic.shell_mesh(n_shells=args.shells, subdivisions=args.subdiv, seed=42)
    tri = mesh.vertices[mesh.faces].astype(np.float64)                       # [F,3,3]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    rng = np.random.default_rng(0)
    pick = rng.choice(len(tri), size=args.points, p=area / area.sum())
    u, v = rng.random(args.points), rng.random(args.points)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    p = tri[pick, 0] + u[:, None] * (tri[pick, 1] - tri[pick, 0]) + v[:, None] * (tri[pick, 2] - tri[pick, 0])
    x01 = (p + 1.5) / 3.0
    # the level table of the bench field: 16 levels, base 16, per-level scale as NGPRadianceField sets it (max res 4096)
    n_levels, base, max_res = 16, 16, 4096
    desc = _C.make_grid_desc(n_levels, args.log2_t, base, float(np.exp(np.log(max_res / base) / (n_levels - 1))))
    rows = []
    for level in range(n_levels):
        scale, res = float(desc.scale[level]), int(desc.resolution[level])
        cell = np.floor(x01 * scale + 0.5).astype(np.int64)
        key = (cell[:, 2] * res + cell[:, 1]) * res + cell[:, 0]
        cells = np.unique(key).size
        out = {"level": level, "resolution": res, "hashed": bool((desc.hashed_mask >> level) & 1),
               "table_rows": int(desc.offset[level + 1] - desc.offset[level]), "surface_cells_seen": int(cells)}
        for b in (4, 8):
            bc = cell // b
            nb = (res + b - 1) // b + 1
            bricks = np.unique((bc[:, 2] * nb + bc[:, 1]) * nb + bc[:, 0]).size
            out[f"bricks_{b}"] = int(bricks)
            out[f"record_MB_bricks_{b}"] = round(bricks * b ** 3 * 64 / 2 ** 20, 1)
            out[f"pointer_grid_MB_{b}"] = round(nb ** 3 * 4 / 2 ** 20, 1)
        out["record_MB_exact_cells"] = round(cells * 64 / 2 ** 20, 1)
        out["table_MB"] = round(out["table_rows"] * 8 / 2 ** 20, 1)
        rows.append(out)
        print(json.dumps(out), flush=True)
    fine = [r for r in rows if r["hashed"]][-6:]
    print(json.dumps({"six_finest_levels": [r["level"] for r in fine],
                      "record_GB_bricks_4": round(sum(r["record_MB_bricks_4"] for r in fine) / 1024, 2),
                      "record_GB_bricks_8": round(sum(r["record_MB_bricks_8"] for r in fine) / 1024, 2),
                      "pointer_grid_GB_4": round(sum(r["pointer_grid_MB_4"] for r in fine) / 1024, 2),
                      "pointer_grid_GB_8": round(sum(r["pointer_grid_MB_8"] for r in fine) / 1024, 2),
                      "note": f"{args.points} area-weighted surface samples: cell counts are lower bounds that saturate "
                              "once the sample spacing is below the cell size"}))


if __name__ == "__main__":
    main()
