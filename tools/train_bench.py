"""Times the differentiable (training) route of NGPRadianceField: forward, backward and an Adam step on n random
points, and the two hash-grid backward kernels alone.  Not the headline bench (that is bench.py, inference);
this is the measurement for SURVEY.md section 8f item 1.

    python tools/train_bench.py --n 1048576 --log2-t 19
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch


def _time(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--log2-t", type=int, default=19)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--head", default="ngp", choices=["ngp", "sg"], help="ngp: SH head (finetune stage); sg: 6-lobe SG head")
    ap.add_argument("--sorted", action="store_true", help="points in Morton order (spatially coherent batches)")
    args = ap.parse_args()
    from quadraturefields_amd import _C, synthetic
    from quadraturefields_amd import tinycudann as tcnn
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew
    dev = torch.device("cuda:0")
    if args.head == "sg":
        field = NGPRadianceFieldSGNew(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=args.log2_t, use_viewdirs=False,
                                      num_g_lobes=6)
        field.load_state_dict(synthetic.seeded_ngp_state(args.log2_t, field.mlp_base.grid.n_rows, sg_lobes=6), strict=False)
    else:
        field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=args.log2_t)
        field.load_state_dict(synthetic.seeded_ngp_state(args.log2_t, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = ((torch.rand(args.n, 3, generator=g) * 2 - 1) * 1.45).to(dev)
    if args.sorted:
        q = ((x + 1.5) / 3.0 * 1023).long().clamp(0, 1023)

        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        x = x[torch.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2))].contiguous()
    d = torch.nn.functional.normalize(torch.randn(args.n, 3, generator=g), dim=-1).to(dev)
    target = torch.rand(args.n, 3, generator=g).to(dev)
    opt = torch.optim.Adam(field.parameters(), lr=1e-3, eps=1e-15)

    def step():
        opt.zero_grad(set_to_none=True)
        rgb, sigma = field(x, d)
        loss = ((rgb - target) ** 2).mean() + 1e-4 * sigma.mean()
        loss.backward()
        opt.step()

    def fwd_train():
        with torch.enable_grad():
            return field(x, d)

    def fwd_fused():
        with torch.no_grad():
            return field(x, d)

    out = {"n": args.n, "log2_T": args.log2_t, "head": args.head, "sorted": args.sorted}
    out["fused_forward_ms"] = _time(fwd_fused, args.iters)
    out["train_forward_ms"] = _time(fwd_train, args.iters)
    out["train_step_ms"] = _time(step, args.iters)
    # the two grid backward kernels alone
    mb = field.mlp_base
    table = mb.params.detach()[mb.n_network_params:].contiguous()
    x01 = ((x + 1.5) / 3.0).contiguous()
    dfeat = torch.randn(args.n, 32, device=dev)
    gt, gx = torch.zeros_like(table), torch.empty_like(x01)
    lib = _C.lib()
    out["grid_backward_table_ms"] = _time(lambda: _C.check(lib.qf_grid_encode_backward(
        mb.grid.desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), args.n, _C.ptr(gt), None, _C.stream()), "bwd"), args.iters)
    ws = torch.empty((int(lib.qf_grid_backward_workspace_bytes(args.n)),), dtype=torch.uint8, device=dev)
    out["grid_backward_table_lds_ms"] = _time(lambda: _C.check(lib.qf_grid_encode_backward_ws(
        mb.grid.desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), args.n, _C.ptr(gt), None, _C.ptr(ws), ws.numel(),
        _C.stream()), "bwd"), args.iters)
    ga, gb = torch.zeros_like(table), torch.zeros_like(table)
    _C.check(lib.qf_grid_encode_backward(mb.grid.desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), args.n, _C.ptr(ga), None,
                                         _C.stream()), "bwd")
    _C.check(lib.qf_grid_encode_backward_ws(mb.grid.desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), args.n, _C.ptr(gb), None,
                                            _C.ptr(ws), ws.numel(), _C.stream()), "bwd")
    out["table_grad_max_abs_diff_quad_vs_lds"] = float((ga - gb).abs().max())
    out["table_grad_max_abs"] = float(ga.abs().max())
    out["grid_backward_input_ms"] = _time(lambda: _C.check(lib.qf_grid_encode_backward(
        mb.grid.desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), args.n, None, _C.ptr(gx), _C.stream()), "bwd"), args.iters)
    out["train_points_per_s"] = args.n / (out["train_step_ms"] * 1e-3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
