"""Headline benchmark: rays/s of the mesh-quadrature render at 800x800 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One step = one full 800x800 frame per rank through the hot path on synthetic inputs already resident in HBM:
ray/mesh intersection (camera-coherent pass, exact BVH traversal as its fallback) -> sample packing -> fused field
evaluation (fp32 hash grid + MLPs) -> per-ray compositing.  N > 1 (launched by torch.distributed.run, one rank per
GPU, RCCL):

* before anything is timed, every data-path collective runs once on ~1 KB of known data on device tensors
  (``parallel.selftest_collectives``): the modes that deliver the right bytes on every rank are the ones used, the run
  exits non-zero with the backend's error text when none does, and the JSON line carries the evidence
  (``distributed``: backend, world size, the device of every rank, the chosen modes);
* ``value`` (weak scaling): the frames of an evaluation run are dealt to the ranks, one frame per rank per step, and
  every finished frame goes to rank 0 -- the rank that scores an evaluation run -- with one ``dist.gather``
  (``--frame-gather all``: all_gather_into_tensor to every rank, the default of rounds 1-3; ``none``: no collective);
* ``sharded_frame`` (BASELINE configs[3]): ONE frame at a time, cut into cost-balanced row bands over the N ranks
  (quadraturefields_amd/parallel.py), each band rendered through the same HIP kernels and the bands exchanged with one
  collective per frame (``all_to_all_single`` with split sizes; padded all_gather / per-rank broadcasts as fallbacks);
  looped over eight seeded scenes.  Reported as latency (host waits for every frame) and as pipelined throughput,
  next to the same loop's 1-rank figure when N = 1.

At N = 1 rank 0 also appends ``configs``: BASELINE configs[2] (1080p, bf16, T = 2^21, dense shells) and configs[4]
(baked SG textures 4096^2, L = 6), each with the roofline of its own dominant kernel measured with HIP events, and
the ``cpu_baseline`` (the oracle on the host cores).  Rank 0 prints ONE JSON line.
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
PROFILE_ROUNDS = ("r4", "r3", "r2", "r1")      # the newest committed PMC reduction wins


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def build_scene(device, seed=42, n_shells=None, subdiv=None, log2_t=None):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    n_shells = N_SHELLS if n_shells is None else n_shells
    subdiv = SUBDIV if subdiv is None else subdiv
    log2_t = LOG2_T if log2_t is None else log2_t
    mesh = synthetic.shell_mesh(n_shells=n_shells, subdivisions=subdiv, seed=seed)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=MAX_HITS, render_step_size=STEP,
                          device=device)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=log2_t)
    field.load_state_dict(synthetic.seeded_ngp_state(log2_t, field.mlp_base.grid.n_rows, seed=seed), strict=False)
    return mesh, mi, field.to(device)


class Stages:
    """One frame, stage by stage, with a HIP event pair around each stage on the launch stream."""
    NAMES = ("traverse", "pack", "field", "composite")

    def __init__(self, mi, field, coherent=True, width=None):
        self.mi, self.field, self.coherent, self.order = mi, field, coherent, None
        self.width = W if width is None else width
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
            hits = self._timed("traverse", lambda: ri._hits_bvh(o, d, MAX_HITS, self.width) + (None,), record)
        hit_tri, hit_t, hit_count, overflow = hits
        pending = self._timed("pack", lambda: ri.pack_hits_begin(o, d, MAX_HITS, hit_tri, hit_t, hit_count, overflow,
                                                                 self.width, lean=self.coherent), record)
        packed = torch.cuda.Event()
        packed.record()                   # a back-half stream waits for this before it reads the samples
        return pending, overflow is not None, record, o.shape[0], (o, d, cam), packed

    def finish(self, begun, _retry=False):
        """Second half, on the stream ``begin`` ran on: wait for the sample count, field, compositing."""
        from quadraturefields_amd import utils
        pending, rastered, record, n_rays, frame_in, packed = begun
        ri = self.mi.rayintersector
        here = torch.cuda.current_stream()
        here.wait_event(packed)           # no-op when begin() ran on this stream
        before = ri._raster_backoff
        # (ray-major frames only: their optimistic re-origin check is read after the field / compositing launches --
        # rule_violated below; the tile pack of a coherent frame applies the rule itself)
        data, order = ri.pack_hits_end(pending, defer_rule_check=True)
        if rastered and ri._raster_backoff > before:
            self.fallbacks = getattr(self, "fallbacks", 0) + 1
        xyz, dirs, index_ray, ts, index_tri, org = data
        layout = ri.last_layout if self.coherent else None
        for t in (xyz, dirs, index_ray, ts) + (tuple(layout) if layout is not None else ()) + (order,):
            if t is not None:
                t.record_stream(here)     # allocated on the front-half stream, read here: keep the allocator from reusing them early
        if layout is not None:      # stream the coherent copies
            _, xyz_c, dirs_c = layout
            rgbs, sigmas = self._timed("field", lambda: self.field(xyz_c, dirs_c), record)
        else:
            rgbs, sigmas = self._timed("field", lambda: self.field(xyz, dirs, order=order if self.coherent else None), record)
        if layout is not None:      # ... and compositing streams the field's outputs in that same order
            frame = ri.last_frame
            for t in (frame.depth_c, frame.hit_count, frame.tile_base):
                t.record_stream(here)
            rgb, alpha, depth, _ = self._timed("composite", lambda: utils.composite_frame(rgbs, sigmas, frame, STEP), record)
        else:
            rgb, alpha, _, depth, _ = self._timed("composite", lambda: utils.derive_properties(
                rgbs, sigmas.reshape(-1), ts, STEP, None, index_ray, bg_color="white", N=n_rays), record)
        if ri.rule_violated():          # some ray had hits closer than the re-origin distance: this frame again, exactly
            # ONE retry, as the product's FrameRenderer._again (render.py): the intersector has switched to the keep
            # masks by now, so a second violation means the rule's bookkeeping is broken -- stop, do not loop
            if _retry:
                raise RuntimeError("bench: the re-origin rule was violated again on the exact retry of a frame")
            self.rule_redone = getattr(self, "rule_redone", 0) + 1
            o, d, cam = frame_in
            return self.finish(self.begin(o, d, cam, record), _retry=True)
        return rgb, alpha, depth, ri.frame_samples() if layout is not None else index_ray.shape[0]

    def frame_nowait(self, o, d, cam, record=False):
        """A whole frame WITHOUT a host wait (DESIGN.md sections 3.1 / 6): the sample count stays on the device -- the tile
        pack leaves it next to the tile bases, the field kernel reads it there -- so nothing between the rays and the
        pixels returns to the host.  The frame's point count is NOT reported (last element 0): bench.py counts the points
        of the timed frames afterwards, by rendering them once more through ``frame`` (the count is a property of the
        frame, not of the timing).  Same pixels as ``frame``."""
        from quadraturefields_amd import utils
        ri = self.mi.rayintersector
        frame = self._timed("traverse", lambda: ri.sample_frame_device(o, d, MAX_HITS, cam), record)
        _, xyz_c, dirs_c = ri.last_layout
        rgbs, sigmas = self._timed("field", lambda: self.field(xyz_c, dirs_c, n_device=frame.total_dev), record)
        rgb, alpha, depth, _ = self._timed("composite", lambda: utils.composite_frame(rgbs, sigmas, frame, STEP), record)
        return rgb, alpha, depth, 0

    def _pack(self, hits):
        """(tools/field_bench.py) hits = (hit_tri, hit_t, hit_count, overflow, o, d) -> packed samples; sets .order."""
        hit_tri, hit_t, hit_count, overflow, o, d = hits
        data, order = self.mi.rayintersector.pack_hits(o, d, MAX_HITS, hit_tri, hit_t, hit_count, overflow, self.width)
        self.order = order if self.coherent else None
        return data

    def stage_ms(self, reduce=np.mean):
        """HIP-event time per stage, averaged (the timed region) or as a median (the untimed stage pass, where a
        one-off stall must not skew a stage)."""
        torch.cuda.synchronize()
        return {k: (float(reduce([a.elapsed_time(b) for a, b in v])) if v else None) for k, v in self.ev.items()}


def cpu_baseline(mesh, field, cam_o, cam_d, crop=600, single_crop=160):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded sample: the
    centre crop x crop pixels of frame 0 through multi-hit intersection on the oracle's host BVH (OpenMP C; the
    reference walks Embree's BVH on the CPU, mesh_utils.py:350-354), torch-CPU field evaluation and compositing.
    The intersection and the field + compositing legs are reported separately.  SURVEY.md 8(d) asks for n = cores AND
    n = 1: the same path on ONE thread (a smaller centre crop, same BVH) rides along as ``single_core``."""
    from oracle import meshpath as om
    from tests import helpers
    cores = om.host_cores()
    torch.set_num_threads(cores)
    y0 = x0 = (W - crop) // 2
    idx = (torch.arange(y0, y0 + crop)[:, None] * W + torch.arange(x0, x0 + crop)[None, :]).reshape(-1)
    o, d = cam_o[idx].numpy(), cam_d[idx].numpy()
    wts = helpers.oracle_ngp_weights(field)
    tb = time.perf_counter()
    bvh = om.BVHIntersector(mesh.vertices, mesh.faces, min_separation=float(_min_sep(mesh)))
    t0 = time.perf_counter()
    sample = om.sampling_raytrace_numpy(bvh, d, o, MAX_HITS)
    t1 = time.perf_counter()
    data = om.to_loader_tensors(sample)
    rgb = om.render_image_finetune(wts, None, data, crop * crop)[0]
    t2 = time.perf_counter()
    n_pts = data[0].shape[0]
    # n = 1: one OpenMP thread, one torch thread, on the centre single_crop x single_crop pixels
    ys = xs = (W - single_crop) // 2
    idx1 = (torch.arange(ys, ys + single_crop)[:, None] * W + torch.arange(xs, xs + single_crop)[None, :]).reshape(-1)
    o1, d1 = cam_o[idx1].numpy(), cam_d[idx1].numpy()
    torch.set_num_threads(1)
    s0 = time.perf_counter()
    sample1 = om.sampling_raytrace_numpy(_OneThread(bvh), d1, o1, MAX_HITS)
    s1 = time.perf_counter()
    data1 = om.to_loader_tensors(sample1)
    om.render_image_finetune(wts, None, data1, single_crop * single_crop)
    s2 = time.perf_counter()
    torch.set_num_threads(cores)
    single = {"value": single_crop * single_crop / (s2 - s0), "unit": "rays/s", "cores": 1,
              "intersection_rays_per_s": single_crop * single_crop / max(s1 - s0, 1e-9),
              "field_composite_points_per_s": data1[0].shape[0] / max(s2 - s1, 1e-9),
              "sample": f"centre {single_crop}x{single_crop} crop of frame 0, {data1[0].shape[0]} quadrature points, one thread"}
    return {
        "value": crop * crop / (t2 - t0), "unit": "rays/s", "cores": cores, "kind": "port",
        "intersection_rays_per_s": crop * crop / max(t1 - t0, 1e-9),
        "field_composite_points_per_s": n_pts / max(t2 - t1, 1e-9),
        "host_bvh_build_s": t0 - tb, "single_core": single,
        "sample": f"centre {crop}x{crop} crop of frame 0 ({crop * crop} rays, {n_pts} quadrature points): multi-hit "
                  f"intersection on the oracle's host BVH over {mesh.faces.shape[0]} triangles {t1 - t0:.2f} s (OpenMP C, "
                  f"{cores} threads; BVH build {t0 - tb:.1f} s, not counted) + torch-CPU field/compositing {t2 - t1:.2f} s",
    }, rgb, idx, (np.asarray(sample[2]), np.asarray(sample[4]))


class _OneThread:
    """The oracle's host BVH intersector with its OpenMP team pinned to one thread (cpu_baseline's n = 1 leg)."""

    def __init__(self, inner):
        self._inner = inner

    def intersects_id(self, origins, vectors, multiple_hits=True, return_locations=True, max_hits=10):
        inner = self._inner
        run = inner._run
        inner._run = lambda o, d, n, k, n_threads, tri, t, cnt: run(o, d, n, k, 1, tri, t, cnt)
        try:
            return inner.intersects_id(origins, vectors, multiple_hits=multiple_hits, return_locations=return_locations,
                                       max_hits=max_hits)
        finally:
            inner._run = run


def _min_sep(mesh):
    from quadraturefields_amd.mesh_utils import trimesh_ray_offset
    return trimesh_ray_offset(mesh.vertices)


def timed_field_events(stages, frames, warm):
    """frames: list of (o, d, cam).  Runs them with HIP events around the field stage only; returns
    (elapsed_s, points, mean field ms)."""
    for f in frames[:warm]:
        stages.frame(*f)
    torch.cuda.synchronize()
    stages.ev["field"] = []
    t0 = time.perf_counter()
    pts = 0
    for f in frames[warm:]:
        pts += stages.frame(*f, "field")[3]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return el, pts, stages.stage_ms()["field"]


def _committed_traffic(name: str):
    """(bytes per point, source) of a dominant kernel from the committed PMC reduction profiles/<round>/<name>, or
    (None, None): bench.py cannot sample PMC counters itself (separate rocprofv3 --pmc passes, tools/config_traffic.sh)."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            tj = json.load(open(path))
            if tj.get("fetch_bytes_per_launch") and tj.get("points_per_launch"):
                per_point = (tj["fetch_bytes_per_launch"] + (tj.get("write_bytes_per_launch") or 0.0)) / tj["points_per_launch"]
                return per_point, f"profiles/{rnd}/{name} (FETCH_SIZE + WRITE_SIZE, bytes per point x points per launch)"
    return None, None


def config3_line(device, steps=5, warm=4):
    """BASELINE configs[2]: 1920x1080, T = 2^21, ~3 M triangles of thin concentric shells (most object rays collect
    more than K = 25 candidates), bf16 tables + MLPs with fp32 accumulate."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    w, h, log2_t = 1920, 1080, 21
    t0 = time.perf_counter()
    mesh, mi, field = build_scene(device, n_shells=36, subdiv=6, log2_t=log2_t)
    field.compute_dtype = "bf16"
    log(f"config 3 scene: {mesh.faces.shape[0]} triangles, built in {time.perf_counter() - t0:.1f} s")
    cams = synthetic.orbit_cameras(steps + warm, seed=42)
    focal = synthetic.lego_focal(w)
    frames = [synthetic.camera_rays(c, focal, w, h, device=device) + (make_camera(c, focal, w, h),) for c in cams]
    stages = Stages(mi, field, width=w)
    el, pts, field_ms = timed_field_events(stages, frames, warm)
    for f in frames[:3]:
        stages.frame(*f, True)
    ms = stages.stage_ms(np.median)
    ms["field"] = field_ms
    ppl = pts / steps
    achieved = ppl * 512 / (field_ms * 1e-3) / 1e9
    t3 = _committed_traffic("config3_traffic.json")
    return {
        "workload": "configs[2]: 1920x1080, bf16 tables + MLPs (fp32 accumulate), T=2^21, dense thin shells, K=25",
        "dtype": "bf16", "triangles": int(mesh.faces.shape[0]), "rays_per_frame": w * h,
        "ms_per_frame": el / steps * 1e3, "rays_per_s": w * h * steps / el, "quadrature_points_per_frame": ppl,
        "mean_hits_per_ray": ppl / (w * h), "stage_ms": ms, "raster_wide": int(mi.rayintersector.raster_wide),
        "dominant_kernel": "field_kernel_bf16<NGP>",
        "roofline": {"kernel": "field_kernel_bf16<NGP>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None if t3[0] is None else t3[0] * ppl, "traffic_unit": "bytes per launch",
                     "traffic_source": t3[1],
                     "algorithmic_bytes_per_point": 512, "points_per_launch": ppl, "avg_launch_ms": field_ms},
    }


def config5_line(device, scene, steps=8, warm=3, texture_size=4096, lobes=6):
    """BASELINE configs[4]: render from the baked SG textures (4096^2 uint8 texture set, L = 6) through the frame path on
    the bench mesh and cameras -- on the mesh's own atlas (one contiguous (azimuth, elevation) chart per shell: what an
    xatlas output looks like; the line's top-level numbers, as in rounds 2-3) and on SURVEY.md 8(d)'s random per-triangle
    charts (no texel locality at all between neighbouring triangles: the worst case), both in ``uv_sets``."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.texture_utils import FeatureCompression
    mesh, mi, field = scene
    tex = synthetic.random_textures(texture_size, lobes, seed=42)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="sigmoid", lambda_thres=7.5, device=device)
    del tex
    uv = torch.from_numpy(synthetic.scaled_uv(mesh, texture_size)).to(device)
    line = _config5_uv_set(device, mi, uv, comp, steps, warm, texture_size, lobes)
    charted = {k: line[k] for k in ("ms_per_frame", "rays_per_s", "quadrature_points_per_frame")}
    charted["shade_ms"] = line["roofline"]["avg_launch_ms"]
    charted["roofline_frac"] = line["roofline"]["frac"]
    charted["what"] = "one contiguous (azimuth, elevation) chart per shell in a 4 x 3 atlas (synthetic.shell_mesh)"
    # the same triangles with their vertices unshared, every triangle its own 4-texel chart at a random place
    mesh_r, uv_r = synthetic.per_triangle_charts(mesh, texture_size, seed=42)
    mi_r = MeshIntersection(mesh_r, simplify_mesh=False, scale=1.0, num_intersections=MAX_HITS, render_step_size=STEP,
                            device=device)
    rnd = _config5_uv_set(device, mi_r, torch.from_numpy(uv_r).to(device), comp, steps, warm, texture_size, lobes)
    del mi_r
    line["uv_sets"] = {
        "charted": charted,
        "per_triangle_random": {"ms_per_frame": rnd["ms_per_frame"], "rays_per_s": rnd["rays_per_s"],
                                "quadrature_points_per_frame": rnd["quadrature_points_per_frame"],
                                "shade_ms": rnd["roofline"]["avg_launch_ms"], "roofline_frac": rnd["roofline"]["frac"],
                                "what": "SURVEY.md 8(d): a 4-texel chart per triangle at a uniformly random place of the "
                                        "atlas (synthetic.per_triangle_charts; vertices unshared)"}}
    return line


def _config5_uv_set(device, mi, uv, comp, steps, warm, texture_size, lobes):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    from quadraturefields_amd.render import FrameRenderer
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    sg = NGPRadianceFieldSGNew(aabb=[-1.5] * 3 + [1.5] * 3, use_viewdirs=False, num_g_lobes=lobes,
                               log2_hashmap_size=14).to(device)          # only features_to_rgb's owner (B-17)
    fr = FrameRenderer(mi, sg, render_step_size=STEP)
    cams = synthetic.orbit_cameras(steps + warm, seed=42)
    focal = synthetic.lego_focal(W)
    frames = [synthetic.camera_rays(c, focal, W, H, device=device) + (make_camera(c, focal, W, H),) for c in cams]
    from quadraturefields_amd import utils
    events, shade = [], utils.shade_baked_points

    def timed_shade(*a_, **k_):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = shade(*a_, **k_)
        b.record()
        events.append((a, b, out[1].shape[0]))
        return out

    utils.shade_baked_points = timed_shade            # the frame path's shading launch, bracketed by HIP events
    for o, d, cam in frames[:warm]:
        fr.render_baked_async(o, d, uv, comp, cam)
    torch.cuda.synchronize()
    events.clear()
    t0 = time.perf_counter()
    for o, d, cam in frames[warm:]:             # no host wait inside a frame: the count stays on the device
        fr.render_baked_async(o, d, uv, comp, cam)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    utils.shade_baked_points = shade
    events = [(a, b, n) for a, b, n in events]
    pts = sum(fr.render_baked(o, d, uv, comp, camera=cam)[3] for o, d, cam in frames[warm:])     # counted afterwards, untimed
    shade_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in events]))
    ppl = pts / steps
    alg = 1 + 3 + 6 * lobes                       # uint8 codes a sample decodes (40 B at L = 6)
    t5 = _committed_traffic("config5_traffic.json")
    achieved = ppl * alg / (shade_ms * 1e-3) / 1e9
    return {
        "workload": f"configs[4]: baked SG textures {texture_size}^2 uint8 x (2+2L) planes, L={lobes}, 800x800 frames, "
                    "FrameRenderer.render_baked_async (tile pack with triangle ids -> texel lookup + decode + SG shading in "
                    "one launch -> tile compositor, no host wait; pixels equal render_image_bake_texture_images_with_occgrid bit for bit)",
        "dtype": "u8 codes -> f32", "rays_per_frame": W * H, "ms_per_frame": el / steps * 1e3,
        "rays_per_s": W * H * steps / el, "quadrature_points_per_frame": ppl,
        "dominant_kernel": "texture_shade_packed_kernel<lookup>",
        "roofline": {"kernel": "texture_shade_packed_kernel<lookup>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None if t5[0] is None else t5[0] * ppl, "traffic_unit": "bytes per launch",
                     "traffic_source": t5[1],
                     "algorithmic_bytes_per_point": alg, "fetched_bytes_per_point": 64 + 12 + 8 + 12,
                     "points_per_launch": ppl, "avg_launch_ms": shade_ms,
                     "note": "one 64-B texel record + position, triangle id and direction per sample (the 128-B triangle "
                             "records are shared by neighbouring samples); the gather of isolated 64-B sectors out of a "
                             "1.07 GB record array is request-bound, not byte-bound: 79 % of the measured HBM sector-gather "
                             "rate (profiles/r4/texture_variants.md)"},
    }


def reference_route_line(device, scene, frames=10, warm=3, texture_size=4096, lobes=6, deform_log2_t=24):
    """What a maintainer who ONLY swaps imports gets (VERDICT r2 missing 3): the reference's own eval loops restated
    over the reference-named entry points, next to the FrameRenderer headline.

    * ``finetune_eval``: the ``test()`` closure of train_finetune.py:575-629 -- loader item (SubjectLoader.__getitem__:
      ray generation, background blend, intersection, the six ray-major sample arrays) -> ``generate_splits`` in
      160 000-ray windows -> ``render_image_finetune_with_occgrid`` per split with ``field_net`` (T = 2^24 as
      train_finetune.py:387-399), ``mesh_finetune`` and ``mesh_intersect`` as the harness passes them -> the harness's
      ``rgb[split[2]] = color[split[2]]`` assembly.  "after": ``scaling = 0`` (the evaluation after the vertex update,
      comparable with the headline frame); "before": ``scaling = 0.0434`` (deformed evaluation).
    * ``baking_eval``: the ``test()`` loop of test_baking_texture_images.py:340-372 -- loader item ->
      ``render_image_bake_texture_images_with_occgrid``.
    ms/frame from the host clock around the whole loop (device synchronised at both ends)."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    from quadraturefields_amd.texture_utils import FeatureCompression
    mesh, mi, field = scene
    n = frames + warm
    cams = np.stack([np.asarray(c, dtype=np.float32) for c in synthetic.orbit_cameras(n, seed=42)])
    images = np.zeros((n, H, W, 4), dtype=np.uint8)          # pixels only feed the PSNR of the harness, not the render
    ds = SubjectLoader.from_arrays(images, cams, synthetic.lego_focal(W), split="test", mesh_intersect=mi, device=device)
    field_net = Field(scale=1.5, precision=16, log2_T=deform_log2_t, L=16, max_res=512, min_res=16, output_dim=1,
                      hidden_size=32, num_features=2, back_prop=False, nl="relu").to(device)
    mesh_finetune = MeshFinetune(mi.mesh.vertices, mi.mesh.faces, 0.0434, device=device)
    n_rays = W * H

    def finetune_frame(i, scaling):
        item = ds[i]
        rays = item["rays"]
        splits = utils.generate_splits(item["data"], rays.origins.shape[0])
        rgb = torch.ones((rays.origins.shape[0], 3), device=device)
        depth = torch.zeros((rays.origins.shape[0],), device=device)
        pts = 0
        for split in splits:
            color, _, d, n_r_s, _, _, _, _, _ = utils.render_image_finetune_with_occgrid(
                field, field_net, None, rays, split, near_plane=0.0, render_step_size=STEP,
                render_bkgd=item["color_bkgd"], cone_angle=0.0, alpha_thre=0.0, mesh_intersect=mi,
                mesh_finetune=mesh_finetune, scaling=scaling)
            rgb[split[2]] = color[split[2]]
            depth[split[2]] = d.squeeze()[split[2]]
            pts += n_r_s
        return rgb, depth, pts

    def timed(fn):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts = 0
        for i in range(warm, n):
            pts += fn(i)[2]
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3, pts / frames

    after_ms, after_pts = timed(lambda i: finetune_frame(i, 0.0))
    before_ms, before_pts = timed(lambda i: finetune_frame(i, 0.0434))
    # the scripts evaluate at up_sample 2 (run_nerfsynthetic_finetune.sh:9): 1600x1600 rays per item, 16 windows, then the
    # INTER_AREA down-sample of train_finetune.py:624-627 (a 2x2 box average)
    from quadraturefields_amd.render import area_downsample
    ds1, n_rays1 = ds, n_rays
    ds = SubjectLoader.from_arrays(images[:6], cams[:6], synthetic.lego_focal(W), split="test", mesh_intersect=mi,
                                   device=device, upsample=2)

    def up2_frame(i):
        rgb, depth, pts = finetune_frame(i % 6, 0.0)
        area_downsample(rgb.reshape(2 * H, 2 * W, 3), 2)
        area_downsample(depth.reshape(2 * H, 2 * W), 2)
        return rgb, depth, pts

    frames1, warm1, n1 = frames, warm, n
    frames, warm, n = 4, 2, 6
    up2_ms, up2_pts = timed(up2_frame)
    frames, warm, n, ds, n_rays = frames1, warm1, n1, ds1, n_rays1
    mesh_finetune.reset_d()
    del field_net
    torch.cuda.empty_cache()

    tex = synthetic.random_textures(texture_size, lobes, seed=42)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="sigmoid", lambda_thres=7.5, device=device)
    del tex
    uv = torch.from_numpy(synthetic.scaled_uv(mesh, texture_size)).to(device)
    sg = NGPRadianceFieldSGNew(aabb=[-1.5] * 3 + [1.5] * 3, use_viewdirs=False, num_g_lobes=lobes,
                               log2_hashmap_size=14).to(device)

    def baking_frame(i):
        item = ds[i]
        rgb, _, depth, n_s, _, _, _, _ = utils.render_image_bake_texture_images_with_occgrid(
            sg, item["rays"], item["data"], texture=None, uv=uv, near_plane=0.0, render_step_size=STEP,
            render_bkgd=item["color_bkgd"], cone_angle=0.0, alpha_thre=0.0, mesh_intersect=mi, mesh_finetune=None,
            scaling=0, discretize=False, compressor=comp)
        return rgb, depth, n_s

    bake_ms, bake_pts = timed(baking_frame)
    return {
        "what": "the reference's eval loops over the reference-named entry points (import swap only, no FrameRenderer): "
                "SubjectLoader item -> generate_splits (160000-ray windows) -> render_image_finetune_with_occgrid per split "
                "-> rgb[split[2]] = color[split[2]]; and SubjectLoader item -> render_image_bake_texture_images_with_occgrid",
        "frames": frames, "rays_per_frame": n_rays,
        "finetune_eval_after": {"scaling": 0.0, "ms_per_frame": after_ms, "rays_per_s": n_rays / (after_ms * 1e-3),
                                "quadrature_points_per_frame": after_pts,
                                "reference": "train_finetune.py:575-629 (test(0, test_dataset))"},
        "finetune_eval_after_up_sample_2": {"scaling": 0.0, "rays_per_frame": 4 * n_rays, "ms_per_frame": up2_ms,
                                            "rays_per_s": 4 * n_rays / (up2_ms * 1e-3), "quadrature_points_per_frame": up2_pts,
                                            "reference": "the scripts' eval: up_sample 2 (run_nerfsynthetic_finetune.sh:9), 16 windows "
                                                         "of 160000 rays, INTER_AREA down-sample (train_finetune.py:620-627)"},
        "finetune_eval_before": {"scaling": 0.0434, "deform_log2_T": deform_log2_t, "ms_per_frame": before_ms,
                                 "rays_per_s": n_rays / (before_ms * 1e-3), "quadrature_points_per_frame": before_pts,
                                 "reference": "train_finetune.py:696 (test(args.scaling, train_whole_dataset))"},
        "baking_eval": {"texture_size": texture_size, "lobes": lobes, "ms_per_frame": bake_ms,
                        "rays_per_s": n_rays / (bake_ms * 1e-3), "quadrature_points_per_frame": bake_pts,
                        "reference": "test_baking_texture_images.py:340-372"},
    }


def sharded_frames(device, rank, world, scene0, n_scenes, frames_per_scene, backend):
    """BASELINE configs[3]: every frame cut into row bands over the ranks, gathered with one collective; eight seeded
    scenes.  Returns the dict for the JSON line (identical on every rank up to timing; rank 0's is printed)."""
    from quadraturefields_amd import parallel, synthetic
    from quadraturefields_amd.render import FrameRenderer
    scenes = [scene0] + [build_scene(device, seed=42 + s) for s in range(1, n_scenes)]
    log(f"sharded_frame: {len(scenes)} scenes ready")
    cams = synthetic.orbit_cameras(frames_per_scene, seed=7)
    focal = synthetic.lego_focal(W)
    rays = [synthetic.camera_rays(c, focal, W, H, device=device) for c in cams]
    shards = [parallel.ShardedFrameRenderer(FrameRenderer(mi, field, render_step_size=STEP), rank, world)
              for _, mi, field in scenes]

    def loop(sync_each):
        """sync_each: the host waits for every gathered frame (latency).  Otherwise frame i's gather is waited for
        after frame i+1's band has been launched (throughput: the exchange overlaps the next render)."""
        pending = frame = None
        for sr in shards:
            for i in range(frames_per_scene):
                started = sr.render_async(rays[i][0], rays[i][1], cams[i], focal, W, H)
                if sync_each:
                    frame = started()
                    torch.cuda.synchronize()
                else:
                    if pending is not None:
                        frame = pending()
                    pending = started
        if pending is not None:
            frame = pending()
        return frame

    def timed(sync_each):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(sync_each)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t[0])
        return el

    loop(True)                      # warm-up: allocations, band profiles, RCCL channels
    n = len(shards) * frames_per_scene
    lat = timed(True)
    thr = timed(False)
    cuts = shards[0].last_cuts
    return {
        "workload": f"configs[3]: {len(scenes)} seeded scenes x {frames_per_scene} frames, each {W}x{H} frame cut into "
                    f"{world} cost-balanced row band(s), one collective of the bands per frame",
        "scenes": len(scenes), "frames": n, "ranks": world, "scaling": "strong",
        "latency_ms_per_frame": lat / n * 1e3, "ms_per_frame_pipelined": thr / n * 1e3,
        "rays_per_s": W * H * n / thr, "rays_per_s_latency_mode": W * H * n / lat,
        "gather": "none (1 rank)" if world == 1 else f"parallel.gather_bands mode {parallel.GATHER_MODE!r}: one collective per "
                  "frame, 20 B/ray (exact = all_to_all_single with split sizes, nothing padded)",
        "band_rows": [cuts[r + 1] - cuts[r] for r in range(world)],
    }


def launch_guard(args):
    """``--gpus N`` must run as N ranks or not at all -- never print a 1-rank line for an N-GPU request.

    * launched by torch.distributed.run (WORLD_SIZE set): it has to equal ``--gpus``, else exit 2;
    * N > 1 and WORLD_SIZE unset (``python bench.py --gpus 8``): this process starts the N ranks itself with the
      driver's own command line (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
      127.0.0.1 ...``) as a CHILD process before any HIP call, and exits with its code; if the box shows fewer than N
      devices (``torch.cuda.device_count()`` does not initialise the GPU) and the run is not a ``--single-device``
      rehearsal, exit 2."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with "
                             f"--nproc-per-node {args.gpus} (refusing to report a {env_world}-rank number for it)")
        return
    if args.gpus == 1:
        return
    if args.gpus < 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus}")
    have = torch.cuda.device_count()
    if have < args.gpus and not args.single_device:
        print(f"bench.py: --gpus {args.gpus} but this box shows {have} GPU(s) and WORLD_SIZE is unset: refusing to "
              f"print a 1-rank line for a {args.gpus}-GPU request", file=sys.stderr, flush=True)
        raise SystemExit(2)
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without a launcher: starting the ranks myself: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[2] / configs[4] lines (N = 1)")
    ap.add_argument("--no-reference-route", action="store_true",
                    help="skip the reference_route leg (the reference-named entry points timed beside the headline, N = 1)")
    ap.add_argument("--scenes", type=int, default=8, help="seeded scenes of the sharded_frame loop (0 = skip it)")
    ap.add_argument("--sharded-frames", type=int, default=4, help="frames per scene in the sharded_frame loop")
    ap.add_argument("--spin-up", type=float, default=1.0,
                    help="seconds of untimed frames before the warm-up steps, to bring the GPU clocks up")
    ap.add_argument("--up-sample", type=int, default=1, choices=[1, 2],
                    help="render at up_sample x 800 per side as the reference's eval does with up_sample 2 "
                         "(train_finetune.py:620-627); the headline configuration is 1")
    ap.add_argument("--pipeline", type=int, default=1, choices=[1, 2],
                    help="1 (default): one frame after the other on one stream.  2: front / back pipeline -- the next frame's "
                         "intersection + pack on one stream beside this frame's field kernel + compositing on another.  "
                         "Measured in round 2: 1.98 ms/frame against 1.87 -- the field kernel runs 1.80 ms instead of 1.32 "
                         "beside the next frame's atomics and streaming writes (it is bound by the memory system, and so "
                         "are they); round 1's whole-frame pipelining lost 4-5 %% for the same reason")
    ap.add_argument("--host-wait", action="store_true",
                    help="round-2 frame: the host waits for every frame's sample count before it launches the field kernel "
                         "(default since round 3: the count stays on the device, no wait -- 1.650 -> 1.628 ms per frame)")
    ap.add_argument("--frame-gather", default=None, choices=["rank0", "all", "none"],
                    help="N > 1, frame-parallel loop: where the finished frames go.  rank0 (default): dist.gather to rank 0, "
                         "the rank that scores / stores an evaluation run; all: all_gather_into_tensor to every rank (the "
                         "default of rounds 1-3); none: no data-path collective")
    ap.add_argument("--no-gather", action="store_true", help="same as --frame-gather none")
    ap.add_argument("--repeats", type=int, default=9,
                    help="after the timed region: this many more untimed-for-value repetitions of the same K-step loop; their "
                         "median / min / max ms_per_step ride along in the JSON line (`repeat`), so that a 2 %% change is "
                         "distinguishable from run-to-run noise (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL).  gloo + --single-device rehearses the multi-rank "
                         "control flow on a one-GPU box (all ranks on cuda:0)")
    ap.add_argument("--single-device", action="store_true", help="map every rank to cuda:0 (rehearsal only)")
    ap.add_argument("--intersector", default="raster", choices=["raster", "bvh"],
                    help="raster: camera-coherent intersector (BVH fallback on overflow); bvh: BVH traversal only")
    args = ap.parse_args()
    launch_guard(args)                     # before anything touches a GPU
    torch.set_grad_enabled(False)          # inference: the fields take the fused kernels
    global W, H
    W = H = 800 * args.up_sample

    from quadraturefields_amd import parallel, synthetic
    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = parallel.init_from_env(args.backend)
    assert world == args.gpus, (world, args.gpus)      # launch_guard
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.no_gather:
        args.frame_gather = "none"
    if args.frame_gather is not None:
        parallel.FRAME_GATHER_MODE = args.frame_gather
    # N > 1: before anything is built or timed, every data-path collective runs once on ~1 KB of known data on device
    # tensors; the modes that deliver the right bytes on EVERY rank are selected (parallel.GATHER_MODE /
    # FRAME_GATHER_MODE), and the run stops with the backend's error text when none does
    dist_info = None
    if world > 1:
        try:
            dist_info = parallel.selftest_collectives(rank, world, device)
            dist_info["devices"] = parallel.device_report(rank, world, device)
        except parallel.CollectiveSelfTestError as e:
            print(f"bench.py: rank {rank}: collective self-test failed: {e}", file=sys.stderr, flush=True)
            raise SystemExit(3)
        log(f"collectives ok: {dist_info['backend']}, {dist_info['world_size']} ranks on "
            f"{dist_info['devices']['distinct_devices']} distinct device(s); bands by {dist_info['band_gather_mode']!r}, "
            f"frames by {dist_info['frame_gather_mode']!r}; {dist_info['tested']}")
        if dist_info["devices"]["distinct_devices"] != world and not args.single_device:
            print(f"bench.py: {world} ranks share {dist_info['devices']['distinct_devices']} device(s): "
                  f"{dist_info['devices']['ranks']}", file=sys.stderr, flush=True)
            raise SystemExit(3)

    log("building scene (mesh, BVH, field)")
    t_build = time.perf_counter()
    mesh, mi, field = build_scene(device)
    log(f"scene ready in {time.perf_counter() - t_build:.1f} s: {mesh.faces.shape[0]} triangles, "
        f"{mi.rayintersector.num_nodes} BVH nodes")
    n_frames = args.steps + args.warmup
    cams = synthetic.orbit_cameras(n_frames * world, seed=42)
    focal = synthetic.lego_focal(W)
    from quadraturefields_amd.mesh_utils import make_camera
    rays = [synthetic.camera_rays(cams[i * world + rank], focal, W, H, device=device) for i in range(n_frames)]
    cameras = [None if args.intersector == "bvh" else make_camera(cams[i * world + rank], focal, W, H) for i in range(n_frames)]
    stages = Stages(mi, field)
    frame_mode = parallel.FRAME_GATHER_MODE if world > 1 else "none"
    gather = frame_mode != "none"
    staged = gather and args.backend == "gloo"          # rehearsal: gloo moves host memory
    receives = gather and (frame_mode == "all" or rank == 0)
    gather_bufs = [torch.empty((world * W * H, 5), dtype=torch.float32, device="cpu" if staged else device)
                   for _ in range(2)] if receives else [None, None]

    # --pipeline 2 (measured, not the default -- see its help text): two frames in flight as a FRONT / BACK pipeline:
    # frame i+1's intersection, offsets, ordering and pack (~0.45 ms) on one stream while frame i's field kernel and
    # compositing (~1.4 ms) run on another; the field kernels of consecutive frames never overlap each other.
    front, back = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    ri0 = mi.rayintersector
    nowait = not args.host_wait and args.intersector == "raster" and args.pipeline == 1

    def run(first, last, record):
        """Frames [first, last); returns (last rgb, total points).  N > 1: every finished frame goes to rank 0 with one
        ``dist.gather`` (``--frame-gather all``: to every rank with one all_gather_into_tensor); the collective of frame
        i (RCCL's own stream) overlaps the render of frame i+1 and is waited for before frame i+1's exchange starts (two
        receive buffers alternate)."""
        pts, rgb, pending, prev = 0, None, None, None

        def exchange(rgb, alpha, depth, i):
            nonlocal pending
            mine = torch.cat([rgb, alpha, depth], dim=1)
            if pending is not None:
                pending()                   # frame i-1's exchange has landed before frame i's starts
            pending = parallel.gather_frames(mine, rank, world, mode=frame_mode, out=gather_bufs[i & 1], async_op=True)

        def complete(begun, i):
            nonlocal pts, rgb
            rgb, alpha, depth, n_pts = stages.finish(begun)
            if gather:
                exchange(rgb, alpha, depth, i)
            pts += n_pts

        if args.pipeline == 1 and nowait:
            for i in range(first, last):
                rgb, alpha, depth, _ = stages.frame_nowait(rays[i][0], rays[i][1], cameras[i], record)
                if gather:
                    exchange(rgb, alpha, depth, i)
        elif args.pipeline == 1:
            for i in range(first, last):
                complete(stages.begin(rays[i][0], rays[i][1], cameras[i], record), i)
        else:
            for i in range(first, last):
                ri0.scratch_slot = i & 1
                with torch.cuda.stream(front):
                    begun = stages.begin(rays[i][0], rays[i][1], cameras[i], record)
                if prev is not None:
                    with torch.cuda.stream(back):
                        complete(*prev)
                prev = (begun, i)
            if prev is not None:
                with torch.cuda.stream(back):
                    complete(*prev)
            ri0.scratch_slot = 0
        if pending is not None:
            pending()
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
    if nowait:              # the no-wait frames do not report their point counts: count the same frames now, untimed
        pts = sum(stages.frame(rays[i][0], rays[i][1], cameras[i])[3] for i in range(args.warmup, n_frames))
        torch.cuda.synchronize()
    t = torch.tensor([elapsed, float(pts)], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    rank_ms = [elapsed / args.steps * 1e3] * 2          # (min, max) of the ranks' own ms_per_step
    if world > 1:
        tmax, tmin, tsum = t.clone(), t.clone(), t.clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tmin, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
        rank_ms = [float(tmin[0]) / args.steps * 1e3, float(tmax[0]) / args.steps * 1e3]
        elapsed, pts_total = float(tmax[0]), float(tsum[1])
    else:
        pts_total = float(pts)

    # The timed region is K = 20 steps of ~1.6 ms: 32 ms.  `value` comes from it alone (the contract), but one such
    # sample cannot tell a 2 % change from noise -- so the same K-step loop runs `--repeats` more times, each bracketed
    # exactly like the timed region, and the spread rides along.
    repeat = None
    if args.repeats > 0:
        samples = []
        for _ in range(args.repeats):
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            run(args.warmup, n_frames, False)
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - r0)
        rt = torch.tensor(samples, dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        if world > 1:
            torch.distributed.all_reduce(rt, op=torch.distributed.ReduceOp.MAX)
        per_step = (rt.cpu().numpy() / args.steps * 1e3).tolist()
        repeat = {"what": f"{args.repeats} more repetitions of the timed region's {args.steps}-step loop (same frames, same "
                          "barriers, max over ranks each; no HIP events inside); `value` does not use them",
                  "ms_per_step_median": float(np.median(per_step)), "ms_per_step_min": float(np.min(per_step)),
                  "ms_per_step_max": float(np.max(per_step)),
                  "value_at_median": W * H * world / (float(np.median(per_step)) * 1e-3)}

    sharded = None
    if args.scenes > 0:             # every rank takes part
        sharded = sharded_frames(device, rank, world, (mesh, mi, field), args.scenes, args.sharded_frames, args.backend)

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
    # the general intersector (exact BVH traversal of the same frames; the camera-coherent pass is the default above)
    ri = mi.rayintersector
    ev = []
    for i in range(min(6, n_frames)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ri._hits_bvh(rays[i][0], rays[i][1], MAX_HITS, W)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    bvh_ms = float(np.median([a.elapsed_time(b) for a, b in ev[1:]]))
    # HBM-side bytes of the dominant kernel come from PMC counters (separate rocprofv3 --pmc passes over this same
    # command, see profiles/<round>/README.md); bench.py cannot sample them itself, so the committed measurement is
    # scaled to this run's points per launch.
    traffic = traffic_src = None
    for rnd in PROFILE_ROUNDS:
        tpath = os.path.join(ROOT, "profiles", rnd, "field_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic_src = f"profiles/{rnd}/field_traffic.json (FETCH_SIZE + WRITE_SIZE, bytes per point x points per launch)"
            traffic = (tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]) / tj["points_per_launch"]
            break
    rays_total = W * H * args.steps * world
    pts_per_launch = pts / args.steps
    field_s = (ms["field"] or 0.0) * 1e-3
    achieved = pts_per_launch * ALG_BYTES_PER_POINT / field_s / 1e9 if field_s > 0 else 0.0
    result = {
        "metric": f"rays/sec at {W}x{H} Lego (mesh-quadrature render: ray/mesh intersection + hash-grid/MLP field + "
                  "compositing)",
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
        "spin_up_s": args.spin_up,
        "config": {
            "workload": f"Lego {W}x{H} (configs[1]), 1xMI355X per frame, fp32 hash-grid + tiny-MLP HIP kernels",
            "rays_per_frame": W * H, "max_hits": MAX_HITS, "triangles": int(mesh.faces.shape[0]),
            "log2_hashmap_size": LOG2_T, "render_step_size": STEP, "up_sample": args.up_sample,
            "intersector": ("camera-coherent pass (exact BVH traversal for overflowing rays)" if args.intersector == "raster"
                            else "BVH traversal"),
            "min_hit_separation": float(mi.rayintersector.min_separation),
            "bvh_fallback_frames": getattr(stages, "fallbacks", 0),
            "overflow_repaired_frames": mi.rayintersector.repaired_frames,
            "reorigin_rule_redone_frames": mi.rayintersector.rule_redone_frames,
            "camera_mismatch_frames": mi.rayintersector.camera_mismatch_frames,     # frames the pass's ray check sent to the BVH
            "mesh_depth_complexity": mi.rayintersector.depth_complexity,
            "frames_in_flight": args.pipeline,
            "host_wait_per_frame": not nowait,
            "parallelism": f"{world} rank(s), the frames dealt round-robin to the ranks, one frame per rank per step"
                           + {"rank0": ", dist.gather of every finished frame to rank 0 (20 B/ray)",
                              "all": ", all_gather_into_tensor of every finished frame to every rank",
                              "none": ", no data-path collective"}[frame_mode],
        },
        "ms_per_step_ranks": {"min": rank_ms[0], "max": rank_ms[1]},
        "quadrature_points_per_frame": pts_per_launch,
        "field_evals_per_s": pts_total / elapsed,
        "field_evals_per_s_in_kernel": pts_per_launch / field_s if field_s > 0 else None,
        "stage_ms": ms,
        "intersect_ms": {"camera_coherent": ms["traverse"] if args.intersector == "raster" else None,
                         "bvh_traversal": bvh_ms},
        "roofline": {
            "kernel": "field_kernel<NGP> (hash-grid gather + MLPs)", "bound": "hbm", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None if traffic is None else traffic * pts_per_launch, "traffic_unit": "bytes per launch",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_point": ALG_BYTES_PER_POINT, "points_per_launch": pts_per_launch,
            "avg_launch_ms": ms["field"],
        },
    }
    if repeat is not None:
        result["repeat"] = repeat
    result["distributed"] = dist_info if dist_info is not None else {
        "backend": None, "world_size": 1, "note": "one rank: no process group, no collective",
        "devices": parallel.device_report(0, 1, device)}
    if sharded is not None:
        result["sharded_frame"] = sharded
    if world == 1 and not args.no_reference_route and args.up_sample == 1:
        log("reference_route (loader item -> generate_splits -> render_image_finetune_with_occgrid; baked-texture loop)")
        result["reference_route"] = reference_route_line(device, (mesh, mi, field))
        torch.cuda.empty_cache()
    if world == 1 and not args.no_configs and args.up_sample == 1:
        log("configs[4] (baked textures)")
        cfg5 = config5_line(device, (mesh, mi, field))
        torch.cuda.empty_cache()
        log("configs[2] (1080p bf16 dense shells)")
        cfg3 = config3_line(device)
        torch.cuda.empty_cache()
        result["configs"] = [cfg3, cfg5]
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        log("cpu baseline (oracle on host cores)")
        base, rgb_o, idx, (ray_o, tri_o) = cpu_baseline(mesh, field, rays[0][0].cpu(), rays[0][1].cpu())
        rgb0 = stages.frame(rays[0][0], rays[0][1], cameras[0])[0].cpu()[idx]
        # ids of the path that was timed: the samples of the full frame, restricted to the crop's rays
        full = mi.rayintersector.sample_device(rays[0][0], rays[0][1], MAX_HITS, W, cameras[0])
        remap = torch.full((W * H,), -1, dtype=torch.int64, device=device)
        remap[idx.to(device)] = torch.arange(idx.shape[0], device=device)
        local = remap[full[2]]
        keep = local >= 0
        base["hit_ids_identical"] = bool(np.array_equal(local[keep].cpu().numpy(), ray_o)
                                         and np.array_equal(full[4][keep].cpu().numpy(), tri_o))
        base["max_abs_err_vs_hip"] = float((rgb0 - rgb_o).abs().max())
        mse = float(((rgb0.double() - rgb_o.double()) ** 2).mean())
        base["psnr_hip_vs_oracle_db"] = float("inf") if mse == 0 else -10.0 * float(np.log10(mse))
        result["cpu_baseline"] = base
    print(json.dumps(result), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
