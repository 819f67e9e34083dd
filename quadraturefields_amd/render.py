"""Whole-frame driver of the mesh-quadrature path: rays -> BVH quadrature points -> field -> composite.

This is the loop body of ``test()`` in ``examples/train_finetune.py:574-629`` (and of
``examples/test_baking_texture_images.py:341-399`` for baked textures) with the DataLoader-side
intersection folded in: every stage is a device launch on the current stream, nothing returns to the host
between the rays and the finished image except one 8-byte sample count.
"""
import copy
import ctypes
from typing import Optional

import torch

from . import _C, utils
from .datasets.utils import Rays
from .field import Field as _Field
from .radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew


class FrameRenderer:
    def __init__(self, mesh_intersect, radiance_field, field_net=None, render_step_size: Optional[float] = None,
                 bg_color: str = "white"):
        self.mesh_intersect = mesh_intersect
        self.radiance_field = radiance_field
        self.field_net = field_net
        # The reference composites with MeshIntersection.render_step_size whatever the renderer's argument says
        # (utils.py:574-585 goes through find_deltas), so there is ONE step size: the mesh intersector's.
        step = float(mesh_intersect.render_step_size)
        if render_step_size is not None and abs(float(render_step_size) - step) > 1e-12 * max(step, 1.0):
            raise ValueError(f"render_step_size {render_step_size} differs from mesh_intersect.render_step_size {step}: "
                             "the deformed and undeformed routes would composite differently")
        self.render_step_size = step
        self.bg_color = bg_color

    @torch.no_grad()
    def row_samples(self):
        """float32 [H]: quadrature points per pixel row of the frame (band) ``render`` just drew with a camera, or None
        (no camera frame / nothing hit).  ``parallel.ShardedFrameRenderer`` balances its row bands with it."""
        frame = self.mesh_intersect.rayintersector.last_frame
        if frame is None:
            return None
        out = torch.empty((frame.height,), dtype=torch.float32, device=frame.hit_count.device)
        _C.check(_C.lib().qf_row_sample_counts(_C.ptr(frame.hit_count), frame.max_hits, frame.width, frame.height,
                                               _C.ptr(out), _C.stream()), "qf_row_sample_counts")
        return out

    @torch.no_grad()
    def quadrature_points(self, origins: torch.Tensor, viewdirs: torch.Tensor, image_width: int = 0, camera=None):
        """[xyzs, dirs, index_ray, ts, index_tri, origins] on the device (None if no hit)."""
        return self.mesh_intersect.sampling_raytrace_device(viewdirs, origins, image_width=image_width, camera=camera)

    def _background(self, n_rays, dev):
        fill = 0.0 if self.bg_color == "black" else 1.0
        return (torch.full((n_rays, 3), fill, device=dev), torch.zeros((n_rays, 1), device=dev),
                torch.zeros((n_rays, 1), device=dev), 0)

    def _again(self, origins, viewdirs, image_width, scaling, render_bkgd, camera, _retry):
        """The optimistic re-origin check of a ray-major pack failed: sample again (the intersector now decides the rule
        up front, so the retry is exact).  ONE retry: a flag that is still raised afterwards is a bug, not a frame."""
        if _retry:
            raise RuntimeError("re-origin rule: the exact retry was refuted as well (RayIntersector._rule_upfront not armed?)")
        return self.render(origins, viewdirs, image_width, scaling, render_bkgd, camera, _retry=True)

    @torch.no_grad()
    def render(self, origins: torch.Tensor, viewdirs: torch.Tensor, image_width: int = 0, scaling: float = 0.0,
               render_bkgd: Optional[torch.Tensor] = None, camera=None, _retry: bool = False):
        """(rgb [R,3], alpha [R,1], depth [R,1], n_samples) for R rays."""
        n_rays = origins.shape[0]
        if camera is not None:
            image_width = camera.width
        ri = self.mesh_intersect.rayintersector
        if self.field_net is not None and scaling != 0 and camera is not None and isinstance(self.field_net, _Field):
            # the "before" evaluation of a camera frame, in the tile order throughout: tile pack -> deformation field
            # (streams) -> displacement + per-ray re-sort inside the tiles -> field -> tile compositor.  Same pixels
            # as render_image_finetune_with_occgrid on the ray-major samples, bit for bit (tested).
            data = ri.sample_device(origins, viewdirs, self.mesh_intersect.num_intersections, image_width, camera, lean=True)
            if data is None:                      # nothing hit: the background, without intersecting a second time
                return self._background(n_rays, origins.device)
            if ri.last_frame is not None:
                frame = ri.last_frame
                _, xyz_c, dirs_c = ri.last_layout
                f = self.field_net(xyz_c, return_grad=False)[0]
                xyz_s, depth_s = utils.deform_frame(f, scaling, xyz_c, dirs_c, frame)
                rgbs, sigmas = self.radiance_field(xyz_s, dirs_c)
                moved = copy.copy(frame)
                moved.depth_c = depth_s
                rgb, alpha, depth, _ = utils.composite_frame(rgbs, sigmas, moved, self.render_step_size,
                                                             render_bkgd=render_bkgd, bg_color=self.bg_color)
                return rgb, alpha, depth, ri.frame_samples()
        # without deformation only the streamed copies are read: skip the ray-major position arrays
        lean = self.field_net is None or scaling == 0
        # a lean (render-only) frame's tile pack applies the re-origin rule itself; only the ray-major route below can
        # come back with rule_violated() (its optimistic pack is verified after the field / compositing launches)
        data = ri.sample_device(origins, viewdirs, self.mesh_intersect.num_intersections, image_width, camera, lean=lean,
                                defer_rule_check=lean)
        if data is None:
            ri.rule_violated()
            return self._background(n_rays, origins.device)
        if (self.field_net is None or scaling == 0) and ri.last_layout is not None:
            # No deformation: the samples are already sorted by (ray, depth), the re-sort of sampling_indexing is the
            # identity, and the field can stream the copies laid out in its processing order (same bits).
            _, xyz_c, dirs_c = ri.last_layout
            rgbs, sigmas = self.radiance_field(xyz_c, dirs_c)
            # ... and compositing streams the field's outputs in that same order (qf_composite_tiles)
            rgb, alpha, depth, _ = utils.composite_frame(rgbs, sigmas, ri.last_frame, self.render_step_size,
                                                         render_bkgd=render_bkgd, bg_color=self.bg_color)
            if ri.rule_violated():      # rare (near-coincident faces): these samples are not the reference's; again, exactly
                return self._again(origins, viewdirs, image_width, scaling, render_bkgd, camera, _retry)
            return rgb, alpha, depth, ri.frame_samples()
        if ri.rule_violated():
            return self._again(origins, viewdirs, image_width, scaling, render_bkgd, camera, _retry)
        rays = Rays(origins=origins, viewdirs=viewdirs)
        rgb, alpha, depth, n_samples, *_ = utils.render_image_finetune_with_occgrid(
            self.radiance_field, self.field_net, None, rays, data, render_step_size=self.render_step_size,
            render_bkgd=render_bkgd, mesh_intersect=self.mesh_intersect, mesh_finetune=None, scaling=scaling,
            bg_color=self.bg_color, order=ri.last_order,
            order_inverse=ri.last_layout[0] if ri.last_layout is not None else None)
        return rgb, alpha, depth, n_samples

    @torch.no_grad()
    def render_async(self, origins: torch.Tensor, viewdirs: torch.Tensor, camera, scaling: float = 0.0,
                     render_bkgd: Optional[torch.Tensor] = None, packed: bool = False):
        """``render`` for a camera frame WITHOUT any host wait: (rgb, alpha, depth, frame) -- with ``packed``
        (rgb | alpha | depth in ONE [R,5] array, None, None, frame).  The frame's sample count
        stays on the device -- the tile pack leaves it next to the tile bases and the field / deformation kernels read
        it there (``n_device``) -- so the whole frame is a fixed sequence of ~10 launches into worst-case buffers and
        the host is free to enqueue the next frame (or band) at once; ``frame`` answers
        ``mesh_intersect.rayintersector.frame_samples()`` later, if anyone asks.  Same pixels as ``render``."""
        ri = self.mesh_intersect.rayintersector
        fused = getattr(self.radiance_field, "discretize", False) is False
        if not fused:
            rgb, alpha, depth, _ = self.render(origins, viewdirs, scaling=scaling, render_bkgd=render_bkgd, camera=camera)
            if packed:
                return torch.cat([rgb, alpha, depth], dim=1), None, None, ri.last_frame
            return rgb, alpha, depth, ri.last_frame
        k = self.mesh_intersect.num_intersections
        deform = self.field_net is not None and scaling != 0
        if (not deform and type(self.radiance_field) in (NGPRadianceField, NGPRadianceFieldSGNew)
                and self.radiance_field.compute_dtype == "fp32" and ri.fused_frame_ready(camera, k)):
            return self._render_async_one_call(origins, viewdirs, camera, k, render_bkgd, packed)
        frame = ri.sample_frame_device(origins, viewdirs, k, camera)
        _, xyz_c, dirs_c = ri.last_layout
        nd = frame.total_dev
        if self.field_net is not None and scaling != 0 and isinstance(self.field_net, _Field):
            f = self.field_net(xyz_c, return_grad=False, n_device=nd)[0]
            xyz_c, depth_s = utils.deform_frame(f, scaling, xyz_c, dirs_c, frame, total_device=nd)
            frame = copy.copy(frame)
            frame.depth_c = depth_s
        elif self.field_net is not None and scaling != 0:
            raise ValueError("render_async: a deformation module other than Field has no device-count route")
        rgbs, sigmas = self.radiance_field(xyz_c, dirs_c, n_device=nd)
        rgb, alpha, depth, _ = utils.composite_frame(rgbs, sigmas, frame, self.render_step_size, render_bkgd=render_bkgd,
                                                     bg_color=self.bg_color, packed=packed)
        return rgb, alpha, depth, frame

    def _render_async_one_call(self, origins, viewdirs, camera, k, render_bkgd, packed, _switched=False):
        """``render_async`` when the whole frame is the library's fixed sequence (``qf_frame_render``: intersection,
        repair, tile offsets, tile pack, the field -- NGP or spherical-Gaussian head --, tile compositor): the buffers are allocated here, the launches are
        enqueued by ONE bound call instead of six -- the host cost of a frame drops from ~0.28 ms to what the
        allocations take, which is what a 0.3 ms row band of a frame sharded over 8 GPUs needs.  Same kernels, same
        arguments, same pixels."""
        ri = self.mesh_intersect.rayintersector
        rf = self.radiance_field
        # launch on the intersector's device (its index is resolved at construction, _C.resolve_device; one switch at most)
        if not _switched and torch._C._cuda_getDevice() != ri.device.index:
            with torch.cuda.device(ri.device):
                return self._render_async_one_call(origins, viewdirs, camera, k, render_bkgd, packed, _switched=True)
        prepared = ri.fused_frame_job(origins, viewdirs, k, camera)
        if prepared is None:                  # the intersector's policy moved while settling an earlier frame
            return self.render_async(origins, viewdirs, camera, 0.0, render_bkgd, packed)
        job, frame, token = prepared
        dev = ri.device
        n, cap = frame.width * frame.height, frame.total
        rgbs = torch.empty((cap, 3), dtype=torch.float32, device=dev)
        sigmas = torch.empty((cap,), dtype=torch.float32, device=dev)
        table, base_w = rf.mlp_base.grid_params(), rf.mlp_base.network_params()
        head_w = sg = sg_params = None
        if type(rf) is NGPRadianceFieldSGNew:         # spherical-Gaussian head: six parameter tensors behind a host struct
            desc = rf._field_desc(_C.HEAD_SG, rf.num_g_lobes)
            sg_params = rf._sg_params()
            sg = _C.SGHead(*[_C.ptr(t) for t in sg_params])
        else:
            desc = rf._field_desc(_C.HEAD_NGP, 0)
            head_w = rf.mlp_head.params.detach()
        mode = utils._BG.get(self.bg_color, _C.BG_CUSTOM)
        bk = _C.f32c(render_bkgd.detach().reshape(3).to(dev)) if mode == _C.BG_CUSTOM else None
        rgb = alpha = depth = out5 = None
        if packed:
            out5 = torch.empty((n, 5), dtype=torch.float32, device=dev)
            job.out_packed = out5.data_ptr()
        else:
            rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
            alpha = torch.empty((n, 1), dtype=torch.float32, device=dev)
            depth = torch.empty((n, 1), dtype=torch.float32, device=dev)
            job.out_rgb, job.out_alpha, job.out_depth = rgb.data_ptr(), alpha.data_ptr(), depth.data_ptr()
        job.field = ctypes.addressof(desc)
        job.table, job.base_w = _C.ptr(table).value, _C.ptr(base_w).value
        job.head_ngp_w = _C.ptr(head_w).value if head_w is not None else None
        job.head_sg = ctypes.addressof(sg) if sg is not None else None
        job.rgb_c, job.sigma_c = rgbs.data_ptr(), sigmas.data_ptr()
        job.delta_const, job.bg_mode = float(self.render_step_size), mode
        job.bkgd = bk.data_ptr() if bk is not None else None
        _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
        ri.fused_frame_done(frame, token)
        frame._keep = frame._keep + (rgbs, sigmas, table, base_w, head_w, sg_params, bk)   # referenced until their readers ran
        return (out5, None, None, frame) if packed else (rgb, alpha, depth, frame)

    @torch.no_grad()
    def render_baked_async(self, origins, viewdirs, uv, compressor, camera):
        """``render_baked`` for a camera frame without a host wait (see ``render_async``): (rgb, alpha, depth, frame).
        The texel lookup + shading launch reads the sample count from device memory (``n_device``)."""
        ri = self.mesh_intersect.rayintersector
        k = self.mesh_intersect.num_intersections
        if ri.fused_frame_ready(camera, k) and torch._C._cuda_getDevice() == ri.device.index:
            # the sampling half as one bound call (qf_frame_render with no field: intersection ... tile pack)
            prepared = ri.fused_frame_job(origins, viewdirs, k, camera, want_tri=True)
        else:
            prepared = None
        if prepared is not None:
            job, frame, token = prepared
            _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
            ri.fused_frame_done(frame, token)
        else:
            frame = ri.sample_frame_device(origins, viewdirs, k, camera, want_tri=True)
        _, xyz_c, dirs_c = ri.last_layout
        rgbs, sigmas = utils.shade_baked_points(self.mesh_intersect, uv, compressor, xyz_c, frame.tri_c, dirs_c,
                                                n_device=frame.total_dev)
        rgb, alpha, depth, _ = utils.composite_frame(rgbs, sigmas, frame, self.render_step_size, bg_color=self.bg_color)
        return rgb, alpha, depth, frame

    @torch.no_grad()
    def render_baked(self, origins, viewdirs, uv, compressor, image_width: int = 0, camera=None):
        """Baked-texture variant (test_baking_texture_images.py:355-371).  With a ``camera`` (the rays are its pixel
        grid) the frame never leaves the intersector's tile order: tile pack with triangle ids -> texel lookup +
        decode + SG shading in one launch -> tile compositor, every stage elementwise or tile-local; same pixels as
        ``render_image_bake_texture_images_with_occgrid`` on the ray-major samples, bit for bit."""
        n_rays = origins.shape[0]
        if camera is not None:
            ri = self.mesh_intersect.rayintersector
            data = ri.sample_device(origins, viewdirs, self.mesh_intersect.num_intersections, camera.width, camera,
                                    lean=True, want_tri=True)
            if data is not None and ri.last_frame is not None and ri.last_frame.tri_c is not None:
                frame = ri.last_frame
                _, xyz_c, dirs_c = ri.last_layout
                rgbs, sigmas = utils.shade_baked_points(self.mesh_intersect, uv, compressor, xyz_c, frame.tri_c, dirs_c)
                rgb, alpha, depth, _ = utils.composite_frame(rgbs, sigmas, frame, self.render_step_size,
                                                             bg_color=self.bg_color)
                return rgb, alpha, depth, ri.frame_samples()
            if data is None:
                dev = origins.device
                return (torch.ones((n_rays, 3), device=dev), torch.zeros((n_rays, 1), device=dev),
                        torch.zeros((n_rays, 1), device=dev), 0)
        data = self.mesh_intersect.sampling_raytrace_device(viewdirs, origins, image_width=image_width, camera=camera,
                                                            layout=False)     # no field evaluation: no processing order
        if data is None:
            dev = origins.device
            return (torch.ones((n_rays, 3), device=dev), torch.zeros((n_rays, 1), device=dev),
                    torch.zeros((n_rays, 1), device=dev), 0)
        rays = Rays(origins=origins, viewdirs=viewdirs)
        rgb, alpha, depth, n_samples, *_ = utils.render_image_bake_texture_images_with_occgrid(
            self.radiance_field, rays, data, uv=uv, render_step_size=self.render_step_size,
            mesh_intersect=self.mesh_intersect, compressor=compressor, bg_color=self.bg_color)
        return rgb, alpha, depth, n_samples


def area_downsample(img: torch.Tensor, factor: int) -> torch.Tensor:
    """cv2.resize(..., INTER_AREA) by an integer factor (train_finetune.py:624-627) == box average."""
    if factor == 1:
        return img
    h, w = img.shape[:2]
    rest = img.shape[2:]
    return img.reshape(h // factor, factor, w // factor, factor, *rest).mean(dim=(1, 3))


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """-10 log10(mse) (train_finetune.py:631-632)."""
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float("inf") if mse == 0 else float(-10.0 * torch.log10(mse))
