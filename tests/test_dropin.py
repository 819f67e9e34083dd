"""CPU: the import surface of the reference's two harness scripts resolves in the package.

The lists below are DATA -- the names `examples/train_finetune.py:2-25,216,272` and
`examples/test_baking_texture_images.py:5-30,196,233,332` import from the modules that are on the accelerated path
(everything else those scripts import is third-party tooling: cv2, lpips, tqdm, tensorboard, torchmetrics ...), plus the
attributes the scripts touch on the objects they build.  Checked two ways: against ``quadraturefields_amd.*`` (the
import swap of INTEGRATION.md section 1) and, in a fresh interpreter with ``quadraturefields_amd/dropin`` first on the
path, under the reference's own module names (no edit of the scripts at all).
"""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (module as the harness names it, names imported from it, harness line)
HARNESS_IMPORTS = [
    ("mesh_utils", ["MeshIntersection", "MeshFinetune"], "train_finetune.py:2, test_baking_texture_images.py:5"),
    ("radiance_fields.ngp", ["NGPRadianceField", "NGPRadianceFieldSGNew"], "train_finetune.py:15, test_baking...:20"),
    ("field", ["Field"], "train_finetune.py:16, test_baking...:21"),
    ("utils", ["MIPNERF360_UNBOUNDED_SCENES", "NERF_SYNTHETIC_SCENES", "render_image_finetune_with_occgrid",
               "render_image_field_with_occgrid", "render_image_with_occgrid", "set_random_seed"], "train_finetune.py:17-24"),
    ("utils", ["MIPNERF360_UNBOUNDED_SCENES", "NERF_SYNTHETIC_SCENES", "render_image_bake_texture_images_with_occgrid",
               "render_image_field_with_occgrid", "render_image_with_occgrid", "set_random_seed"],
     "test_baking_texture_images.py:22-29"),
    ("nerfacc.estimators.occ_grid", ["OccGridEstimator"], "train_finetune.py:25, test_baking...:30"),
    ("build.lib.intersector", ["Intersector"], "train_finetune.py:216,716 (from build.lib import intersector)"),
    ("datasets.nerf_synthetic", ["SubjectLoader"], "train_finetune.py:272, test_baking...:233"),
    ("texture_utils", ["FeatureCompression"], "test_baking_texture_images.py:332"),
    # what the reference's own modules on the path import from the third-party libraries
    ("datasets.utils", ["Rays", "namedtuple_map"], "utils.py:16"),
    ("field_rendering", ["rendering_field"], "utils.py:23"),
    ("nerfacc.volrend", ["accumulate_along_rays_", "render_weight_from_density", "rendering"], "utils.py:22"),
    ("nerfacc.pack", ["pack_info"], "field_rendering.py:10"),
    ("nerfacc.scan", ["exclusive_prod", "exclusive_sum"], "field_rendering.py:11"),
    ("kaolin.render.spc", ["mark_pack_boundaries", "exponential_integration", "sum_reduce"], "utils.py:24,869-879"),
    ("tinycudann", ["Encoding", "Network", "NetworkWithInputEncoding"], "ngp.py:325-358, field.py:157-171"),
]

# harness module name -> where the import swap of INTEGRATION.md section 1 points it
SWAP = {
    "mesh_utils": "quadraturefields_amd.mesh_utils", "radiance_fields.ngp": "quadraturefields_amd.radiance_fields.ngp",
    "field": "quadraturefields_amd.field", "utils": "quadraturefields_amd.utils",
    "nerfacc.estimators.occ_grid": "quadraturefields_amd.estimators", "build.lib.intersector": "quadraturefields_amd.intersector",
    "datasets.nerf_synthetic": "quadraturefields_amd.datasets.nerf_synthetic", "texture_utils": "quadraturefields_amd.texture_utils",
    "datasets.utils": "quadraturefields_amd.datasets.utils", "field_rendering": "quadraturefields_amd.field_rendering",
    "nerfacc.volrend": "quadraturefields_amd.field_rendering", "nerfacc.pack": "quadraturefields_amd.field_rendering",
    "nerfacc.scan": "quadraturefields_amd.field_rendering", "kaolin.render.spc": "quadraturefields_amd.spc_render",
    "tinycudann": "quadraturefields_amd.tinycudann",
}

# (class, attributes / methods the two scripts touch on its instances) -- grep of `<object>.<attr>` over both scripts
TOUCHED = [
    ("quadraturefields_amd.mesh_utils", "MeshIntersection",
     ["sampling_raytrace_numpy", "sampling_indexing", "find_deltas"]),                 # + .mesh / .vertices / .rayintersector (instance)
    ("quadraturefields_amd.mesh_utils", "RayIntersector", ["inter", "intersects_id", "update_intersector"]),
    ("quadraturefields_amd.mesh_utils", "MeshFinetune", ["reset_d", "update_faces", "update_d"]),
    ("quadraturefields_amd.radiance_fields.ngp", "NGPRadianceField",
     ["query_density", "forward", "eval", "train", "parameters", "state_dict", "load_state_dict"]),
    ("quadraturefields_amd.radiance_fields.ngp", "NGPRadianceFieldSGNew", ["query_density", "forward", "features_to_rgb", "features"]),
    ("quadraturefields_amd.field", "Field", ["forward", "parameters", "state_dict"]),
    ("quadraturefields_amd.estimators", "OccGridEstimator",
     ["sampling", "update_every_n_steps", "eval", "train", "state_dict", "load_state_dict"]),
    ("quadraturefields_amd.texture_utils", "FeatureCompression", ["get_features_from_texture_map", "compress", "save_to_file"]),
    ("quadraturefields_amd.datasets.nerf_synthetic", "SubjectLoader",
     ["update_num_rays", "fetch_data", "preprocess", "HEIGHT", "WIDTH", "__getitem__", "__len__"]),
    ("quadraturefields_amd.intersector", "Intersector", ["find_intersections", "update_vertices"]),
]


def test_import_swap_resolves_every_harness_name():
    for mod, names, where in HARNESS_IMPORTS:
        m = importlib.import_module(SWAP[mod])
        for n in names:
            assert hasattr(m, n), f"{SWAP[mod]}.{n} missing ({where})"
    from quadraturefields_amd import utils
    assert utils.MIPNERF360_UNBOUNDED_SCENES == ["garden", "bicycle", "bonsai", "counter", "kitchen", "room", "stump"]
    assert utils.NERF_SYNTHETIC_SCENES == ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]


def test_touched_attributes_exist():
    for mod, cls, attrs in TOUCHED:
        c = getattr(importlib.import_module(mod), cls)
        for a in attrs:
            assert hasattr(c, a), f"{mod}.{cls}.{a}"


def test_render_image_field_with_occgrid_signature_is_the_references():
    """utils.py:353-371 of the reference: positional order and defaults."""
    import inspect
    from quadraturefields_amd import utils
    sig = inspect.signature(utils.render_image_field_with_occgrid)
    assert list(sig.parameters) == ["radiance_field", "estimator", "rays", "near_plane", "far_plane", "render_step_size",
                                    "render_bkgd", "cone_angle", "alpha_thre", "test_chunk_size", "timestamps"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["near_plane"], d["far_plane"], d["render_step_size"], d["cone_angle"], d["alpha_thre"], d["test_chunk_size"]) == \
        (0.0, 1e10, 1e-3, 0.0, 0.0, 8192)
    assert d["render_bkgd"] is None and d["timestamps"] is None


def test_dropin_directory_serves_the_harness_imports_verbatim():
    """A fresh interpreter with quadraturefields_amd/dropin first on the path executes the harness's own import
    statements (restated from HARNESS_IMPORTS) under the reference's module names."""
    lines = ["import sys"]
    for mod, names, _ in HARNESS_IMPORTS:
        if mod == "build.lib.intersector":
            lines.append("from build.lib import intersector; intersector.Intersector")
        elif mod == "tinycudann":
            lines.append("import tinycudann as tcnn; " + "; ".join(f"tcnn.{n}" for n in names))
        elif mod == "kaolin.render.spc":
            lines.append("import kaolin.render.spc as spc_render; " + "; ".join(f"spc_render.{n}" for n in names))
        else:
            lines.append(f"from {mod} import ({', '.join(names)})")
    lines.append("import mesh_utils, utils; assert 'quadraturefields_amd' in mesh_utils.MeshIntersection.__module__")
    lines.append("print('ok')")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "quadraturefields_amd", "dropin"), ROOT])
    proc = subprocess.run([sys.executable, "-c", "\n".join(lines)], env=env, capture_output=True, text=True, timeout=300,
                          cwd="/tmp")
    assert proc.returncode == 0 and proc.stdout.strip().endswith("ok"), proc.stderr[-2000:]


def test_bench_refuses_a_world_it_was_not_launched_with():
    """VERDICT r2 missing 2 / weak 6: ``python bench.py --gpus 8`` without torchrun (WORLD_SIZE unset) must never print
    a 1-rank line for an 8-GPU request.  On a box with fewer GPUs than asked it exits non-zero before any GPU call."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                          env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert proc.returncode != 0
    assert '"metric"' not in proc.stdout
    assert "--gpus 8" in (proc.stderr + proc.stdout)
    # launched with a WORLD_SIZE that disagrees with --gpus: refused as well
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                          env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert proc.returncode != 0 and '"metric"' not in proc.stdout


def test_non_unit_directions_on_the_host_are_called_out():
    """The multi-hit rule is stated for unit directions: host arrays are checked, once per intersector."""
    import warnings
    import numpy as np
    from quadraturefields_amd.mesh_utils import RayIntersector
    ri = RayIntersector.__new__(RayIntersector)          # no device needed: only the check itself
    ri.min_separation = 1e-3
    d = np.tile(np.array([[0.0, 0.0, 2.0]], np.float32), (4, 1))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ri._warn_if_not_unit(d)
        ri._warn_if_not_unit(d)                          # once
        assert len(w) == 1 and "unit" in str(w[0].message)
    ri2 = RayIntersector.__new__(RayIntersector)
    ri2.min_separation = 1e-3
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ri2._warn_if_not_unit(d / 2.0)
        assert len(w) == 0
