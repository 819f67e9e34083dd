"""What a ray's BVH traversal is made of: runs the ``trav_stats`` experiment build of the library
(tools/experiments/variants.py: counters in bvh8_traverse_kernel, nothing else changed) on the bench scene's 800x800
frame and on a 2^17-ray random training batch and prints the counts per ray.

    python tools/experiments/variants.py trav_stats            # build container
    QF_HIP_LIBRARY=tools/experiments/_build/libqf_trav_stats.so QF_HIP_LIBRARY_EXPERIMENT=1 python tools/trav_stats.py
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import bench

NAMES = ["rays", "node_steps", "leaf_steps", "triangles_tested", "wave_node_steps", "wave_leaf_steps", "extra_pages",
         "hits_offered"]


def main():
    torch.set_grad_enabled(False)
    from quadraturefields_amd import _C, synthetic
    dev = torch.device("cuda:0")
    mesh, mi, field = bench.build_scene(dev)
    ri = mi.rayintersector
    lib = _C.lib()
    read = lib.qf_trav_stats_read
    read.restype, read.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()

    def stats(fn):
        torch.cuda.synchronize()
        assert read(buf, 1) == 0
        fn()
        torch.cuda.synchronize()
        assert read(buf, 1) == 0
        return dict(zip(NAMES, [int(v) for v in buf]))

    K, W, H = bench.MAX_HITS, bench.W, bench.H
    cams = synthetic.orbit_cameras(8, seed=1)
    focal = synthetic.lego_focal(W)
    rays = [synthetic.camera_rays(c, focal, W, H, device=dev) for c in cams]
    out = {"triangles": int(mesh.faces.shape[0]), "wide_nodes": ri.num_wide_nodes, "max_stack": ri.max_stack}
    o, d = rays[0]
    pool_o, pool_d = torch.cat([r[0] for r in rays]), torch.cat([r[1] for r in rays])
    g = torch.Generator(device=dev).manual_seed(0)
    pick = torch.randint(0, pool_o.shape[0], (1 << 17,), device=dev, generator=g)
    bo, bd = pool_o[pick].contiguous(), pool_d[pick].contiguous()
    for name, fn, n_rays in (("frame_800x800", lambda: ri._hits_bvh(o, d, K, W), W * H),
                             ("batch_2e17_random", lambda: ri._hits_bvh(bo, bd, K, 0), 1 << 17)):
        s = stats(fn)
        waves = n_rays / 8.0
        s["per_ray"] = {k: s[k] / max(s["rays"], 1) for k in ("node_steps", "leaf_steps", "triangles_tested", "hits_offered")}
        s["per_wave"] = {"node_steps": s["wave_node_steps"] / waves, "leaf_steps": s["wave_leaf_steps"] / waves}
        # lockstep efficiency: octet-steps actually needed / (wave steps x 8 octets)
        s["octet_utilisation"] = {"node": s["node_steps"] / max(8.0 * s["wave_node_steps"], 1.0),
                                  "leaf": s["leaf_steps"] / max(8.0 * s["wave_leaf_steps"], 1.0)}
        s["triangles_per_leaf_visit"] = s["triangles_tested"] / max(s["leaf_steps"], 1)
        out[name] = s
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
