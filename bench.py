"""Headline benchmark: rays/s of the mesh-quadrature render at 800x800 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One step = one full 800x800 frame through the hot path on synthetic inputs already resident in HBM:
BVH multi-hit traversal -> sample packing -> fused field evaluation (fp32 hash grid + MLPs) -> per-ray
compositing.  N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank renders its own
frame (weak scaling, no collective on the data path) and the finished tiles are exchanged with one
all_gather_into_tensor per step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

W = H = 800
MAX_HITS = 25
LOG2_T = 19
N_SHELLS, SUBDIV = 12, 6           # 12 x 81,920 = 983,040 triangles
STEP = 5e-3
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_POINT = 16 * 8 * 2 * 4   # 16 levels x 8 corners x 2 features x 4 B


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def build_scene(device, seed=42):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    mesh = synthetic.shell_mesh(n_shells=N_SHELLS, subdivisions=SUBDIV, seed=seed)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=MAX_HITS, render_step_size=STEP,
                          device=device)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=LOG2_T)
    field.load_state_dict(synthetic.seeded_ngp_state(LOG2_T, field.mlp_base.grid.n_rows, seed=seed), strict=False)
    return mesh, mi, field.to(device)


class Stages:
    """One frame, stage by stage, with a HIP event pair around each stage on the launch stream."""
    NAMES = ("traverse", "pack", "field", "composite")

    def __init__(self, mi, field, coherent=True):
        self.mi, self.field, self.coherent, self.order = mi, field, coherent, None
        self.ev = {k: [] for k in self.NAMES}

    def _timed(self, name, fn, record):
        """record: False, True (an event pair around every stage) or a stage name (only that stage).  An event
        record costs a ~10 us bubble on the stream, so the timed region only brackets the dominant kernel; the
        other stages are timed in a separate pass after it."""
        if not (record is True or record == name):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.ev[name].append((a, b))
        return out

    def frame(self, o, d, cam=None, record=False):
        """cam (mesh_utils.make_camera): the rays are that camera's pixel grid -> camera-coherent intersector,
        with the exact K-nearest BVH traversal as fallback when a pixel collects more than MAX_HITS candidates."""
        return self.finish(self.begin(o, d, cam, record))

    def begin(self, o, d, cam=None, record=False):
        """First half of a frame on the current stream, no host wait: intersection, offsets scan, the 16-byte
        readback, pack and ordering kernels."""
        ri = self.mi.rayintersector
        if ri.want_raster(cam):
            hits = self._timed("traverse", lambda: ri._hits_raster_frame(o, d, MAX_HITS, cam), record)
        else:
            hits = self._timed("traverse", lambda: ri._hits_bvh(o, d, MAX_HITS, W) + (None,), record)
        hit_tri, hit_t, hit_count, overflow = hits
        pending = self._timed("pack", lambda: ri.pack_hits_begin(o, d, MAX_HITS, hit_tri, hit_t, hit_count, overflow, W,
                                                                 lean=self.coherent), record)
        return pending, overflow is not None, record, o.shape[0]

    def finish(self, begun):
        """Second half, on the stream ``begin`` ran on: wait for the sample count, field, compositing."""
        from quadraturefields_amd import utils
        pending, rastered, record, n_rays = begun
        ri = self.mi.rayintersector
        before = ri._raster_backoff
        data, order = ri.pack_hits_end(pending)
        if rastered and ri._raster_backoff > before:
            self.fallbacks = getattr(self, "fallbacks", 0) + 1
        xyz, dirs, index_ray, ts, index_tri, org = data
        layout = ri.last_layout if self.coherent else None
        if layout is not None:      # stream the coherent copies; compositing picks colour / density up through the inverse map
            inverse, xyz_c, dirs_c = layout
            rgbs, sigmas = self._timed("field", lambda: self.field(xyz_c, dirs_c), record)
        else:
            inverse = None
            rgbs, sigmas = self._timed("field", lambda: self.field(xyz, dirs, order=order if self.coherent else None), record)
        out = self._timed("composite", lambda: utils.derive_properties(
            rgbs, sigmas.reshape(-1), ts, STEP, None, index_ray, bg_color="white", N=n_rays, sample_index=inverse), record)
        rgb, alpha, _, depth, _ = out
        return rgb, alpha, depth, index_ray.shape[0]

    def _pack(self, hits):
        """(tools/field_bench.py) hits = (hit_tri, hit_t, hit_count, overflow, o, d) -> packed samples; sets .order."""
        hit_tri, hit_t, hit_count, overflow, o, d = hits
        data, order = self.mi.rayintersector.pack_hits(o, d, MAX_HITS, hit_tri, hit_t, hit_count, overflow, W)
        self.order = order if self.coherent else None
        return data

    def stage_ms(self, reduce=np.mean):
        """HIP-event time per stage, averaged (the timed region) or as a median (the untimed stage pass, where a
        one-off stall must not skew a stage)."""
        torch.cuda.synchronize()
        return {k: (float(reduce([a.elapsed_time(b) for a, b in v])) if v else None) for k, v in self.ev.items()}


def cpu_baseline(mesh, field, cam_o, cam_d, crop=200):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded sample:
    the centre crop x crop pixels of frame 0 through brute-force multi-hit intersection (OpenMP C), torch-CPU
    field evaluation and compositing."""
    from oracle import meshpath as om
    from tests import helpers
    cores = om.host_cores()
    torch.set_num_threads(cores)
    y0 = x0 = (W - crop) // 2
    idx = (torch.arange(y0, y0 + crop)[:, None] * W + torch.arange(x0, x0 + crop)[None, :]).reshape(-1)
    o, d = cam_o[idx].numpy(), cam_d[idx].numpy()
    wts = helpers.oracle_ngp_weights(field)
    bf = om.BruteForceIntersector(mesh.vertices, mesh.faces)
    t0 = time.perf_counter()
    sample = om.sampling_raytrace_numpy(bf, d, o, MAX_HITS)
    t1 = time.perf_counter()
    data = om.to_loader_tensors(sample)
    rgb = om.render_image_finetune(wts, None, data, crop * crop)[0]
    t2 = time.perf_counter()
    n_pts = data[0].shape[0]
    return {
        "value": crop * crop / (t2 - t0), "unit": "rays/s", "cores": cores, "kind": "port",
        "sample": f"centre {crop}x{crop} crop of frame 0 ({crop * crop} rays, {n_pts} quadrature points): "
                  f"brute-force intersection over {mesh.faces.shape[0]} triangles {t1 - t0:.1f} s (OpenMP C, "
                  f"{cores} threads) + torch-CPU field/compositing {t2 - t1:.2f} s "
                  f"({n_pts / max(t2 - t1, 1e-9):.0f} points/s)",
    }, rgb, idx, (np.asarray(sample[2]), np.asarray(sample[4]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spin-up", type=float, default=1.0,
                    help="seconds of untimed frames before the warm-up steps, to bring the GPU clocks up")
    ap.add_argument("--up-sample", type=int, default=1, choices=[1, 2],
                    help="render at up_sample x 800 per side as the reference's eval does with up_sample 2 "
                         "(train_finetune.py:620-627); the headline configuration is 1")
    ap.add_argument("--pipeline", type=int, default=1, choices=[1, 2, 3],
                    help="frames in flight (one HIP stream each); 1 = strictly one frame after the other.  Measured: "
                         "2 or 3 frames in flight are 4-5 %% SLOWER (the fabric-bound field kernels of two frames "
                         "overlap each other and the small kernels gain nothing), so the default stays 1")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: all_gather every finished frame (rgb, alpha, depth) to all ranks.  Off by default: the "
                         "frames are independent units dealt to the ranks, the path has no exchange step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL).  gloo + --single-device rehearses the multi-rank "
                         "control flow on a one-GPU box (all ranks on cuda:0)")
    ap.add_argument("--single-device", action="store_true", help="map every rank to cuda:0 (rehearsal only)")
    ap.add_argument("--intersector", default="raster", choices=["raster", "bvh"],
                    help="raster: camera-coherent intersector (BVH fallback on overflow); bvh: BVH traversal only")
    args = ap.parse_args()
    torch.set_grad_enabled(False)          # inference: the fields take the fused kernels
    global W, H
    W = H = 800 * args.up_sample

    from quadraturefields_amd import parallel, synthetic
    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = parallel.init_from_env(args.backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    log("building scene (mesh, BVH, field)")
    mesh, mi, field = build_scene(device)
    log(f"scene ready: {mesh.faces.shape[0]} triangles, {mi.rayintersector.num_nodes} BVH nodes")
    n_frames = args.steps + args.warmup
    cams = synthetic.orbit_cameras(n_frames * world, seed=42)
    focal = synthetic.lego_focal(W)
    from quadraturefields_amd.mesh_utils import make_camera
    rays = [synthetic.camera_rays(cams[i * world + rank], focal, W, H, device=device) for i in range(n_frames)]
    cameras = [None if args.intersector == "bvh" else make_camera(cams[i * world + rank], focal, W, H) for i in range(n_frames)]
    stages = Stages(mi, field)

    # --pipeline N keeps N frames in flight: frame i's intersection / pack kernels are enqueued on one stream before
    # the host waits for frame i-1's sample count and launches its field + compositing kernels on another.
    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.pipeline))]
    gather = args.gather and world > 1
    gather_bufs = {s: torch.empty((world * W * H, 5), dtype=torch.float32, device=device) for s in streams} if gather else {}

    def complete(begun, s):
        with torch.cuda.stream(s):
            rgb, alpha, depth, n_pts = stages.finish(begun)
            if gather:
                torch.distributed.all_gather_into_tensor(gather_bufs[s], torch.cat([rgb, alpha, depth], dim=1))
        return rgb, n_pts

    def run(first, last, record):
        """Frames [first, last) with len(streams) of them in flight; returns (last rgb, total points)."""
        inflight, pts, rgb = [], 0, None
        for i in range(first, last):
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                begun = stages.begin(rays[i][0], rays[i][1], cameras[i], record)
            inflight.append((begun, s))
            if len(inflight) == len(streams):
                rgb, n_pts = complete(*inflight.pop(0))
                pts += n_pts
        while inflight:
            rgb, n_pts = complete(*inflight.pop(0))
            pts += n_pts
        return rgb, pts

    # Device spin-up (untimed, before the W warm-up steps): the scene build leaves the GPU idle for seconds and its
    # clocks take tens of milliseconds of sustained work to come back -- longer than the whole default timed region.
    t_spin, i_spin = time.perf_counter(), 0
    while time.perf_counter() - t_spin < args.spin_up:      # cycling through the cameras, so that a profiler's
        i = i_spin % n_frames                               # per-kernel averages see the timed region's frame mix
        stages.frame(rays[i][0], rays[i][1], cameras[i])
        torch.cuda.synchronize()
        i_spin += 1
    run(0, args.warmup, False)
    torch.cuda.synchronize()
    log(f"{args.warmup} warmup frames done")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rgb, pts = run(args.warmup, n_frames, "field")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    t = torch.tensor([elapsed, float(pts)], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tsum = t.clone()
        torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
        elapsed, pts_total = float(tmax[0]), float(tsum[1])
    else:
        pts_total = float(pts)

    if rank != 0:
        parallel.shutdown()
        return
    ms = stages.stage_ms()                      # field: HIP events inside the timed region
    for i in range(min(5, n_frames)):           # the other stages: an extra, untimed pass with events everywhere
        stages.frame(rays[i][0], rays[i][1], cameras[i], True)
    field_ms = ms["field"]
    stages.ev["field"] = []
    ms = stages.stage_ms(np.median)
    ms["field"] = field_ms
    # HBM-side bytes of the dominant kernel come from PMC counters (separate rocprofv3 --pmc passes over this same
    # command, see profiles/r1/README.md); bench.py cannot sample them itself, so the committed measurement is scaled
    # to this run's points per launch.
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "r1", "field_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic_src = "profiles/r1/field_traffic.json (FETCH_SIZE + WRITE_SIZE, bytes per point x points per launch)"
        traffic = (tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]) / tj["points_per_launch"]
    rays_total = W * H * args.steps * world
    pts_per_launch = pts / args.steps
    field_s = (ms["field"] or 0.0) * 1e-3
    achieved = pts_per_launch * ALG_BYTES_PER_POINT / field_s / 1e9 if field_s > 0 else 0.0
    result = {
        "metric": f"rays/sec at {W}x{H} Lego (mesh-quadrature render: BVH traversal + hash-grid/MLP field + compositing)",
        "value": rays_total / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"Lego {W}x{H} (configs[1]), 1xMI355X per frame, fp32 hash-grid + tiny-MLP HIP kernels",
            "rays_per_frame": W * H, "max_hits": MAX_HITS, "triangles": int(mesh.faces.shape[0]),
            "log2_hashmap_size": LOG2_T, "render_step_size": STEP, "up_sample": args.up_sample,
            "intersector": args.intersector, "bvh_fallback_frames": getattr(stages, "fallbacks", 0),
            "overflow_repaired_frames": mi.rayintersector.repaired_frames,
            "frames_in_flight": len(streams),
            "parallelism": f"{world} rank(s), the frames dealt round-robin to the ranks, one frame per rank per step"
                           + (", all_gather of the finished frames" if gather else ", no data-path collective"),
        },
        "quadrature_points_per_frame": pts_per_launch,
        "field_evals_per_s": pts_total / elapsed,
        "field_evals_per_s_in_kernel": pts_per_launch / field_s if field_s > 0 else None,
        "stage_ms": ms,
        "roofline": {
            "kernel": "field_kernel<NGP> (hash-grid gather + MLPs)", "bound": "hbm", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None if traffic is None else traffic * pts_per_launch, "traffic_unit": "bytes per launch",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_point": ALG_BYTES_PER_POINT, "points_per_launch": pts_per_launch,
            "avg_launch_ms": ms["field"],
        },
    }
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        log("cpu baseline (oracle on host cores)")
        base, rgb_o, idx, (ray_o, tri_o) = cpu_baseline(mesh, field, rays[0][0].cpu(), rays[0][1].cpu())
        rgb0 = stages.frame(rays[0][0], rays[0][1], cameras[0])[0].cpu()[idx]
        crop = mi.sampling_raytrace_device(rays[0][1][idx.to(device)].contiguous(), rays[0][0][idx.to(device)].contiguous())
        base["hit_ids_identical"] = bool(np.array_equal(crop[2].cpu().numpy(), ray_o)
                                         and np.array_equal(crop[4].cpu().numpy(), tri_o))
        base["max_abs_err_vs_hip"] = float((rgb0 - rgb_o).abs().max())
        mse = float(((rgb0.double() - rgb_o.double()) ** 2).mean())
        base["psnr_hip_vs_oracle_db"] = float("inf") if mse == 0 else -10.0 * float(np.log10(mse))
        result["cpu_baseline"] = base
    print(json.dumps(result), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
