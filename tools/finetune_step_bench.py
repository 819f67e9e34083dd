"""Times one optimisation step of the finetune stage (examples/train_finetune.py:465-533) on the bench scene:
2^17 random rays of random cameras -> BVH quadrature points -> render_image_finetune_with_occgrid (deformation field
T = 2^24 as the reference builds it + NGP field, training) -> smooth-L1 + regulariser -> backward -> Adam.  Prints one JSON object.

    python tools/finetune_step_bench.py --iters 10
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1 << 17)
    ap.add_argument("--deform-log2-t", type=int, default=24,
                    help="log2 table size of the deformation field; the reference builds Field(log2_T=24), train_finetune.py:387-399")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused: quadraturefields_amd.optim.Adam (one launch per tensor); torch: torch.optim.Adam (foreach)")
    args = ap.parse_args()
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune
    dev = torch.device("cuda:0")
    mesh, mi, field = bench.build_scene(dev)
    net = Field(scale=1.5, precision=16, log2_T=args.deform_log2_t, L=16, max_res=512, min_res=16, output_dim=1,
                hidden_size=32, num_features=2, back_prop=False, nl="relu").to(dev)
    finetune = MeshFinetune(mesh.vertices, mesh.faces, 0.0434, device=dev)
    from quadraturefields_amd.optim import Adam
    opt = (Adam if args.optimizer == "fused" else torch.optim.Adam)(list(field.parameters()) + list(net.parameters()),
                                                                    lr=1e-3, eps=1e-15)
    cams = synthetic.orbit_cameras(8, seed=1)
    focal = synthetic.lego_focal(bench.W)
    pool_o, pool_d = zip(*[synthetic.camera_rays(c, focal, bench.W, bench.H, device=dev) for c in cams])
    pool_o, pool_d = torch.cat(pool_o), torch.cat(pool_d)
    target = torch.rand(args.rays, 3, device=dev)
    ev = {k: [] for k in ("intersect", "forward", "backward", "optimizer")}

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def step(record):
        pick = torch.randint(0, pool_o.shape[0], (args.rays,), device=dev)
        o, d = pool_o[pick].contiguous(), pool_d[pick].contiguous()
        t0 = mark()
        with torch.no_grad():
            data = mi.sampling_raytrace_device(d, o)
        t1 = mark()
        out = utils.render_image_finetune_with_occgrid(field, net, None, Rays(origins=o, viewdirs=d), data,
                                                       render_step_size=5e-3, mesh_intersect=mi, mesh_finetune=finetune,
                                                       scaling=0.0434)
        loss = torch.nn.functional.smooth_l1_loss(out[0], target) + out[7].sum()
        t2 = mark()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        t3 = mark()
        opt.step()
        t4 = mark()
        if record:
            for k, a, b in (("intersect", t0, t1), ("forward", t1, t2), ("backward", t2, t3), ("optimizer", t3, t4)):
                ev[k].append((a, b))
        return data[0].shape[0]

    for _ in range(3):
        step(False)
    torch.cuda.synchronize()
    pts = sum(step(True) for _ in range(args.iters))
    torch.cuda.synchronize()
    ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in ev.items()}
    print(json.dumps({"rays": args.rays, "deform_log2_T": args.deform_log2_t, "deform_table_rows": int(net.xyz_encoder.grid.n_rows),
                      "optimizer": args.optimizer, "points_per_step": pts / args.iters, "stage_ms": ms,
                      "step_ms": sum(ms.values()), "rays_per_s": args.rays / (sum(ms.values()) * 1e-3)}))


if __name__ == "__main__":
    main()
