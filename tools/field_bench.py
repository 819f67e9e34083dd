"""Field-kernel microbenchmark on the real quadrature points of bench frame 0 (used under rocprofv3 --pmc).
    python tools/field_bench.py [--iters N] [--order ray|tile]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--stage", default="field", choices=["field", "traverse", "pack", "composite", "frame"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from quadraturefields_amd import synthetic
    mesh, mi, field = bench.build_scene(dev)
    o, d = synthetic.camera_rays(synthetic.orbit_cameras(1, seed=42)[0], synthetic.lego_focal(bench.W), bench.W, bench.H, device=dev)
    st = bench.Stages(mi, field)
    hits = mi.rayintersector.hits(o, d, bench.MAX_HITS, image_width=bench.W)
    data = st._pack(hits)
    xyz, dirs, index_ray, ts, index_tri, org = data
    n = xyz.shape[0]

    def run():
        if args.stage == "field":
            return field(xyz, dirs)
        if args.stage == "traverse":
            return mi.rayintersector.hits(o, d, bench.MAX_HITS, image_width=bench.W)
        if args.stage == "pack":
            return st._pack(hits)
        if args.stage == "frame":
            return st.frame(o, d)
        from quadraturefields_amd import utils
        return utils.derive_properties(rgbs, sig.reshape(-1), ts, bench.STEP, None, index_ray, N=o.shape[0])

    rgbs, sig = field(xyz, dirs)
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.iters
    print(f"stage={args.stage} n_points={n} rays={o.shape[0]} ms={ms:.4f} points/s={n / ms * 1e3:.4e} rays/s={o.shape[0] / ms * 1e3:.4e}")


if __name__ == "__main__":
    main()
