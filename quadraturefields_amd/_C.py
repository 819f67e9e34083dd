"""ctypes binding of libqf_hip.so (the C ABI declared in include/qf_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.  torch is only
used for device memory and streams; every tensor crosses the boundary as a raw device pointer.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint32, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# QF_HIP_LIBRARY: load another build of the C ABI instead of the in-tree one (experiment builds of tools/, which live
# outside the package directory).  A library whose qf_abi_version() differs is refused unless
# QF_HIP_LIBRARY_EXPERIMENT=1 accepts the experiment offset (+1000) on top of the right version.
LIB_PATH = os.environ.get("QF_HIP_LIBRARY") or os.path.join(_PKG, "libqf_hip.so")

QF_MAX_LEVELS = 16
QF_MAX_LOBES = 8
QF_BVH_MAX_HITS = 64
QF_TEXEL_RECORD_BYTES = 64
QF_TEXEL_TRIANGLE_RECORD_BYTES = 128
HEAD_NONE, HEAD_NGP, HEAD_SG, HEAD_SG_FEATURES = 0, 1, 2, 3
BG_WHITE, BG_BLACK, BG_CUSTOM, BG_NONE = 0, 1, 2, 3


class GridDesc(Structure):
    _fields_ = [
        ("n_levels", c_uint32), ("n_features", c_uint32), ("log2_hashmap_size", c_uint32),
        ("base_resolution", c_uint32), ("per_level_scale", c_float), ("hashed_mask", c_uint32),
        ("offset", c_uint32 * (QF_MAX_LEVELS + 1)), ("resolution", c_uint32 * QF_MAX_LEVELS),
        ("scale", c_float * QF_MAX_LEVELS),
    ]


class FieldDesc(Structure):
    _fields_ = [("grid", GridDesc), ("aabb", c_float * 6), ("head", c_int32), ("n_lobes", c_int32)]


class SGHead(Structure):
    _fields_ = [("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
                ("wout", c_void_p), ("bout", c_void_p)]


class Camera(Structure):
    _fields_ = [("c2w", c_float * 12), ("fx", c_float), ("fy", c_float), ("cx", c_float), ("cy", c_float),
                ("width", c_int32), ("height", c_int32)]


class FrameJob(Structure):
    """qf_frame_job (include/qf_hip.h): one render-only camera frame as one host call."""
    _fields_ = [("camera", c_void_p), ("rays_o", c_void_p), ("rays_d", c_void_p), ("n_rays", c_int64),
                ("max_hits", c_int32), ("cull_chunks", c_int32), ("min_separation", c_float), ("bg_mode", c_int32),
                ("delta_const", c_float), ("reserved_", c_int32),
                ("hit_tri", c_void_p), ("hit_t", c_void_p), ("hit_count", c_void_p), ("final_count", c_void_p),
                ("tile_base", c_void_p), ("total", c_void_p), ("host_block", c_void_p), ("dropped", c_void_p),
                ("xyz_c", c_void_p), ("dirs_c", c_void_p), ("depth_c", c_void_p), ("tri_c", c_void_p),
                ("field", c_void_p), ("table", c_void_p), ("base_w", c_void_p), ("head_ngp_w", c_void_p),
                ("head_sg", c_void_p), ("rgb_c", c_void_p), ("sigma_c", c_void_p),
                ("bkgd", c_void_p), ("out_rgb", c_void_p), ("out_alpha", c_void_p), ("out_depth", c_void_p),
                ("out_packed", c_void_p)]


class TextureSet(Structure):
    _fields_ = [("alpha", c_void_p), ("diffuse", c_void_p), ("colors", c_void_p * QF_MAX_LOBES),
                ("lambda_axis", c_void_p * QF_MAX_LOBES), ("texture_size", c_int32), ("n_lobes", c_int32),
                ("sigmoid_codec", c_int32), ("lambda_thres", c_float)]


_P = c_void_p
_SIGNATURES = {
    "qf_status_string": (c_char_p, [c_int]),
    "qf_frame_render": (c_int, [_P, POINTER(FrameJob), _P]),
    "qf_abi_version": (c_int, []),
    "qf_device_cu_count": (c_int, []),
    "qf_grid_desc_init": (c_int, [POINTER(GridDesc), c_uint32, c_uint32, c_uint32, c_double]),
    "qf_grid_encode": (c_int, [POINTER(GridDesc), _P, _P, c_int64, _P, _P]),
    "qf_grid_encode_backward": (c_int, [POINTER(GridDesc), _P, _P, _P, c_int64, _P, _P, _P]),
    "qf_grid_backward_workspace_bytes": (c_int64, [c_int64]),
    "qf_grid_encode_backward_ws": (c_int, [POINTER(GridDesc), _P, _P, _P, c_int64, _P, _P, _P, c_int64, _P]),
    "qf_grid_encode_double_backward": (c_int, [POINTER(GridDesc), _P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int64, _P]),
    "qf_grid_mlp_forward": (c_int, [POINTER(GridDesc), _P, _P, _P, c_int64, _P, _P]),
    "qf_field_forward": (c_int, [POINTER(FieldDesc), _P, _P, _P, POINTER(SGHead), _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qf_field_forward_bf16": (c_int, [POINTER(FieldDesc), _P, _P, _P, POINTER(SGHead), _P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "qf_ngp_mlp_backward": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "qf_sg_mlp_backward": (c_int, [_P, _P, _P, c_int64, _P, _P, POINTER(SGHead), c_int32, c_int64, _P, _P, POINTER(SGHead), _P]),
    "qf_sg_features_to_rgb": (c_int, [_P, c_int64, _P, c_int64, c_int32, _P, _P]),
    "qf_sg_features_to_rgb_backward": (c_int, [_P, c_int64, _P, _P, c_int64, c_int32, _P, c_int64, _P]),
    "qf_deform_field_forward": (c_int, [POINTER(GridDesc), _P, c_float, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "qf_deform_mlp_backward": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qf_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double, c_int32, c_int64, _P]),
    "qf_apply_deformation": (c_int, [_P, c_float, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "qf_mark_pack_boundaries": (c_int, [_P, c_int64, _P, _P]),
    "qf_exponential_integration": (c_int, [_P, c_int32, _P, _P, c_int64, c_int64, c_int32, _P, _P, _P]),
    "qf_sum_reduce": (c_int, [_P, c_int32, _P, c_int64, c_int64, _P, _P]),
    "qf_derive_properties": (c_int, [_P, _P, _P, _P, c_float, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "qf_deform_resort_tiles": (c_int, [_P, c_float, _P, _P, _P, _P, c_int32, _P, c_int64, _P, c_int32, c_int32, _P, _P, _P]),
    "qf_composite_tiles": (c_int, [_P, _P, _P, c_float, _P, c_int32, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "qf_row_sample_counts": (c_int, [_P, c_int32, c_int32, c_int32, _P, _P]),
    "qf_derive_properties_backward": (c_int, [_P, _P, _P, _P, c_float, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qf_pack_info": (c_int, [_P, c_int64, c_int64, _P, _P]),
    "qf_exclusive_scan": (c_int, [_P, _P, c_int64, c_int64, c_int32, _P, _P]),
    "qf_accumulate_along_rays": (c_int, [_P, _P, c_int32, _P, c_int64, c_int64, _P, _P]),
    "qf_render_from_density": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qf_bvh_create": (c_int, [_P, c_int64, POINTER(c_void_p)]),
    "qf_bvh_create_ex": (c_int, [_P, c_int64, c_int32, POINTER(c_void_p)]),
    "qf_bvh_refit": (c_int, [_P, _P, c_int64]),
    "qf_bvh_refit_device": (c_int, [_P, _P, c_int64, _P]),
    "qf_bvh_num_wide_nodes": (c_int64, [_P]),
    "qf_bvh_max_stack": (c_int32, [_P]),
    "qf_bvh_copy_wide_nodes": (c_int, [_P, _P, c_int64]),
    "qf_bvh_set_min_separation": (c_int, [_P, c_float]),
    "qf_bvh_min_separation": (c_float, [_P]),
    "qf_filter_hits": (c_int, [_P, c_int64, c_int32, _P, _P, _P, _P]),
    "qf_bvh_destroy": (None, [_P]),
    "qf_bvh_num_triangles": (c_int64, [_P]),
    "qf_bvh_num_nodes": (c_int64, [_P]),
    "qf_bvh_max_depth": (c_int32, [_P]),
    "qf_bvh_copy_nodes": (c_int, [_P, _P, c_int64]),
    "qf_bvh_copy_tri_ids": (c_int, [_P, _P, c_int64]),
    "qf_bvh_intersect": (c_int, [_P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "qf_bvh_repair_overflow": (c_int, [_P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "qf_raster_intersect": (c_int, [_P, POINTER(Camera), _P, _P, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_int32, _P, _P]),
    "qf_raster_intersect_wide": (c_int, [_P, POINTER(Camera), _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_int32, _P,
                                         _P]),
    "qf_raster_intersect_slabs": (c_int, [_P, POINTER(Camera), _P, _P, c_int64, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P,
                                          _P]),
    "qf_grid_march_count": (c_int, [POINTER(c_float), POINTER(c_int32), _P, _P, _P, _P, _P, c_int64, c_float, c_float,
                                    c_float, _P, _P]),
    "qf_grid_march_write": (c_int, [POINTER(c_float), POINTER(c_int32), _P, _P, _P, _P, _P, c_int64, c_float, c_float,
                                    c_float, _P, _P, _P, _P, _P]),
    "qf_generate_rays": (c_int, [POINTER(Camera), c_int32, _P, _P, _P]),
    "qf_scatter_max": (c_int, [_P, _P, c_int64, c_int64, _P, _P]),
    "qf_sample_offsets_temp_bytes": (c_int64, [c_int64]),
    "qf_sample_offsets": (c_int, [_P, c_int64, c_int32, _P, _P, c_int64, _P]),
    "qf_frame_offsets_temp_bytes": (c_int64, [c_int64]),
    "qf_frame_offsets": (c_int, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, _P, _P, _P, c_int32, _P]),
    "qf_banded_tile_count": (c_int64, [c_int32, c_int32, c_int32]),
    "qf_tile_offsets": (c_int, [_P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "qf_pack_tiles": (c_int, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, c_int32, _P]),
    "qf_pack_samples": (c_int, [_P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P]),
    "qf_tile_totals": (c_int, [_P, c_int32, c_int32, _P, _P]),
    "qf_coherent_order": (c_int, [_P, _P, _P, c_int32, c_int32, _P, _P]),
    "qf_coherent_layout": (c_int, [_P, _P, _P, c_int32, c_int32, _P, _P, c_int32, _P]),
    "qf_resort_by_depth": (c_int, [_P, _P, c_int64, _P, _P]),
    "qf_resort_samples": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qf_split_layout": (c_int, [_P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "qf_mesh_update_d": (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P, _P]),
    "qf_texel_indices": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "qf_texel_records_pack": (c_int, [_P, _P, _P, c_int64, _P, _P]),
    "qf_texel_indices_packed": (c_int, [_P, _P, _P, c_int64, c_int32, _P, _P]),
    "qf_texture_fetch": (c_int, [POINTER(TextureSet), _P, c_int64, _P, _P]),
    "qf_texture_shade": (c_int, [POINTER(TextureSet), _P, _P, c_int64, _P, _P, _P]),
    "qf_texture_pack": (c_int, [POINTER(TextureSet), _P, _P]),
    "qf_texture_shade_packed": (c_int, [_P, c_int32, c_int32, c_int32, c_float, _P, _P, c_int64, _P, _P, _P]),
    "qf_texture_shade_points": (c_int, [_P, c_int32, c_int32, c_int32, c_float, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
ABI_VERSION = 5              # QF_ABI_VERSION of include/qf_hip.h

_lib = None


def lib() -> ctypes.CDLL:
    """Load libqf_hip.so (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m quadraturefields_amd.build` "
                "(hipcc --offload-arch=gfx950).  quadraturefields_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        version = handle.qf_abi_version()
        experiment = os.environ.get("QF_HIP_LIBRARY_EXPERIMENT") == "1" and version == ABI_VERSION + 1000
        if version != ABI_VERSION and not experiment:
            raise RuntimeError(f"{LIB_PATH}: ABI version {version}, expected {ABI_VERSION} (an experiment build "
                               "reports +1000 and is only loaded with QF_HIP_LIBRARY_EXPERIMENT=1)")
        _lib = handle
    return _lib


class QFError(RuntimeError):
    pass


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().qf_status_string(status).decode()
        if status == -1:
            raise ValueError(f"{what}: {msg}")
        raise QFError(f"{what}: {msg} (status {status})")


def ptr(t, dtype=None, device_required: bool = True):
    """Raw pointer of a contiguous tensor (None -> NULL).  The HIP kernels only accept device memory."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    if device_required and not t.is_cuda:
        raise RuntimeError("quadraturefields_amd kernels need tensors on the HIP device (no CPU fallback)")
    return c_void_p(t.data_ptr())


_OWNER_PID = os.getpid()          # the process that imported the package (and will own the HIP context)
_FORKED = False


def _after_fork_in_child():
    global _FORKED
    _FORKED = True


os.register_at_fork(after_in_child=_after_fork_in_child)


def raw_stream() -> int:
    """The current torch HIP stream of the current device as an integer handle.  The public
    ``torch.cuda.current_stream().cuda_stream`` costs ~9 us of Python per call (device-index and availability look-ups,
    a Stream object) -- with ~10 launches per 0.3 ms row band of a sharded frame that alone kept the host behind the
    GPU; the raw getter is one C call."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream():
    """The current torch HIP stream as a raw handle.  Every kernel launch of the package goes through here, so this is
    also where a FORKED child is stopped: the reference runs its intersector inside DataLoader worker processes
    (train_finetune.py:313 with num_workers > 0), and a HIP context cannot be created or used in a forked child -- it
    hangs or corrupts the parent's.  Use num_workers=0 for the device path, or keep only ray generation in the workers
    (INTEGRATION.md section 1)."""
    if _FORKED:
        raise RuntimeError(
            "quadraturefields_amd was called from a forked child process (a DataLoader worker?): the HIP kernels must run "
            f"in the process that imported the package (pid {_OWNER_PID}).  Use num_workers=0 for the device path, or a "
            "'spawn' multiprocessing context.")
    return c_void_p(raw_stream())


def resolve_device(device) -> torch.device:
    """``torch.device(device)`` WITH an index: "cuda" means the current device at construction time.  The package
    compares ``device.index`` with the raw current-device ordinal (``mesh_utils._on_device``, the renderers' fast
    paths); an index-less device never compared equal -- ``FrameRenderer.render_async`` then called itself until
    RecursionError."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


def i64c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.int64).contiguous()


def make_grid_desc(n_levels: int, log2_hashmap_size: int, base_resolution: int, per_level_scale: float) -> GridDesc:
    d = GridDesc()
    check(lib().qf_grid_desc_init(ctypes.byref(d), n_levels, log2_hashmap_size, base_resolution,
                                  float(per_level_scale)), "qf_grid_desc_init")
    return d


def grid_encode_backward(desc, table, x01, dfeat, n: int, grad_table, grad_x01) -> None:
    """qf_grid_encode_backward with the workspace of the LDS-partitioned table scatter (torch memory) when the batch
    is large enough for it to pay; tensors as the C entry point takes them (None -> NULL)."""
    ws, ws_bytes = None, 0
    if grad_table is not None and n >= (1 << 15):
        ws_bytes = int(lib().qf_grid_backward_workspace_bytes(n))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x01.device)
    check(lib().qf_grid_encode_backward_ws(desc, ptr(table), ptr(x01), ptr(dfeat), n, ptr(grad_table), ptr(grad_x01),
                                           ptr(ws), ws_bytes, stream()), "qf_grid_encode_backward")
