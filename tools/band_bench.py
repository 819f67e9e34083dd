"""Strong-scaling rehearsal on ONE GPU: the bands an N-rank ShardedFrameRenderer would render, one after the other.

For N in --ranks: the frame is cut into N cost-balanced row bands (parallel.band_cuts on the previous frame's row
profile, as ShardedFrameRenderer does), every band is rendered through FrameRenderer.render with its band camera, and
the per-band HIP-event times are reported.  max(band) + the gather is what a frame costs on N GPUs; sum(band) / N
would be perfect scaling.  No collective runs here (one process).  Prints one JSON object.

    python tools/band_bench.py --ranks 1 2 4 8 --frames 6
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--frames", type=int, default=6)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from quadraturefields_amd import parallel, synthetic
    from quadraturefields_amd.render import FrameRenderer
    dev = torch.device("cuda:0")
    mesh, mi, field = bench.build_scene(dev)
    fr = FrameRenderer(mi, field, render_step_size=bench.STEP)
    w = h = bench.W
    focal = synthetic.lego_focal(w)
    cams = synthetic.orbit_cameras(args.frames + 2, seed=42)
    rays = [synthetic.camera_rays(c, focal, w, h, device=dev) for c in cams]
    out = {"workload": f"{w}x{h} frame of the bench scene cut into N row bands, bands rendered one after the other on one GPU",
           "ranks": {}}
    for n in args.ranks:
        sh = [parallel.ShardedFrameRenderer(fr, r, n) for r in range(n)]
        per_band = []
        for i, (c2w, (o, d)) in enumerate(zip(cams, rays)):
            cuts = sh[0].cuts_for(h)
            times, bands, samples = [], [], []
            for r in range(n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                bands.append(sh[r].render_band(o, d, c2w, focal, w, h, cuts[r], cuts[r + 1]))
                b.record()
                torch.cuda.synchronize()
                times.append(a.elapsed_time(b))
                rs = sh[r]._band_samples
                samples.append(rs if rs is not None else torch.full((cuts[r + 1] - cuts[r],), -1.0, device=dev))
            frame, row_samples = torch.cat(bands, dim=0), torch.cat(samples)
            for s in sh:                                  # every rank sees the same gathered frame: same next cuts
                s._push_profile(frame, w, h, row_samples)
            if i >= 2:
                per_band.append(times)
        # the same bands enqueued back to back (no sync in between): what the host needs per band, and what the GPU does
        import time
        cuts = sh[0].cuts_for(h)
        host_us, gpu_ms = [], []
        for c2w, (o, d) in list(zip(cams, rays))[2:]:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            t0 = time.perf_counter()
            keep = [sh[r].render_band(o, d, c2w, focal, w, h, cuts[r], cuts[r + 1]) for r in range(n)]
            host_us.append((time.perf_counter() - t0) * 1e6 / n)
            b.record()
            torch.cuda.synchronize()
            gpu_ms.append(a.elapsed_time(b) / n)
            del keep
        t = np.array(per_band)
        out["ranks"][str(n)] = {"band_ms_mean": t.mean(axis=0).round(4).tolist(), "max_band_ms": float(t.max(axis=1).mean()),
                                "sum_band_ms": float(t.sum(axis=1).mean()), "last_cuts": cuts,
                                "back_to_back_ms_per_band": float(np.mean(gpu_ms)),
                                "host_enqueue_us_per_band": float(np.mean(host_us)),
                                "speedup_if_gather_free": float(out["ranks"]["1"]["max_band_ms"] / t.max(axis=1).mean())
                                if "1" in out["ranks"] else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
