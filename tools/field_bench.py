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
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--order", default="coherent", choices=["ray", "tile", "tile4", "random", "coherent"])
    ap.add_argument("--physical", action="store_true", help="permute the inputs into the processing order instead of indexing through it")
    ap.add_argument("--deform-log2-t", type=int, default=24)
    ap.add_argument("--stage", default="field", choices=["field", "traverse", "raster", "pack", "composite", "frame", "deform"])
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    from quadraturefields_amd import synthetic
    mesh, mi, field = bench.build_scene(dev)
    o, d = synthetic.camera_rays(synthetic.orbit_cameras(1, seed=42)[0], synthetic.lego_focal(bench.W), bench.W, bench.H, device=dev)
    field.compute_dtype = args.dtype
    st = bench.Stages(mi, field)
    from quadraturefields_amd.mesh_utils import make_camera
    cam = make_camera(synthetic.orbit_cameras(1, seed=42)[0], synthetic.lego_focal(bench.W), bench.W, bench.H)
    ri = mi.rayintersector
    hits = ri._hits_bvh(o, d, bench.MAX_HITS, bench.W) + (None, o, d)
    data = st._pack(hits)
    xyz, dirs, index_ray, ts, index_tri, org = data
    n = xyz.shape[0]
    order = None
    if args.order == "coherent":
        hc = hits[2]
        cs = torch.cumsum(hc.to(torch.int64), 0)
        order = ri.coherent_order(hc, (cs - hc).contiguous(), int(cs[-1]), bench.W)
    elif args.order != "ray":
        ray = index_ray
        first = torch.zeros(o.shape[0] + 1, dtype=torch.int64, device=dev)
        first[1:] = torch.cumsum(torch.bincount(ray, minlength=o.shape[0]), 0)
        k = torch.arange(n, device=dev) - first[ray]
        px, py = ray % bench.W, ray // bench.W
        if args.order == "tile":      # (8x8 tile, hit rank, pixel in tile)
            key = ((py // 8) * (bench.W // 8) + px // 8) * (64 * 64) + k * 64 + (py % 8) * 8 + px % 8
        elif args.order == "tile4":   # (4x4 tile, hit rank, pixel in tile): one wave pass = one 4x4 patch at one rank
            key = ((py // 4) * (bench.W // 4) + px // 4) * (64 * 16) + k * 16 + (py % 4) * 4 + px % 4
        else:
            key = torch.randperm(n, device=dev)
        order = torch.argsort(key).to(torch.int32).contiguous()

    if args.physical and order is not None:
        xyz, dirs = xyz[order.long()].contiguous(), dirs[order.long()].contiguous()
        order = None
    net = None
    if args.stage == "deform":      # the deformation field of train_finetune.py:387-399 (T = 2^24: a 1 GB fp32 table)
        from quadraturefields_amd.field import Field
        net = Field(scale=1.5, precision=16, log2_T=args.deform_log2_t, L=16, max_res=512, min_res=16, output_dim=1,
                    hidden_size=32, num_features=2, back_prop=False, nl="relu")
        net.load_state_dict(synthetic.seeded_deform_state(net.xyz_encoder.grid.n_params), strict=False)
        net = net.to(dev)

    def run():
        if args.stage == "deform":
            return net(xyz, return_grad=False, order=order)[0]
        if args.stage == "field":
            return field(xyz, dirs, order=order)
        if args.stage == "traverse":
            return ri._hits_bvh(o, d, bench.MAX_HITS, bench.W)
        if args.stage == "raster":
            return ri._hits_raster(o, d, bench.MAX_HITS, cam)
        if args.stage == "pack":
            return st._pack(hits)
        if args.stage == "frame":
            return st.frame(o, d, cam)
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
