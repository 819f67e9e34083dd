"""CPU: the C-ABI library builds, loads without a GPU and exports exactly what include/qf_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "qf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from quadraturefields_amd import _C
    declared = _header_functions()
    assert declared, "no functions parsed from the header"
    assert sorted(_C.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_integration_index_lists_every_entry_point():
    """INTEGRATION.md's entry-point index names every function the header declares."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [f for f in _header_functions() if f"`{f}`" not in text]
    assert not missing, missing


def test_library_exports_every_symbol(lib):
    from quadraturefields_amd import _C
    raw = ctypes.CDLL(_C.LIB_PATH)
    for name in _header_functions():
        getattr(raw, name)            # AttributeError if missing


def test_host_only_entry_points(lib):
    """Status strings and the grid level table are host computations: callable without a GPU."""
    from quadraturefields_amd import _C
    from oracle import fields as ofields
    assert lib.qf_abi_version() == _C.ABI_VERSION == 5
    assert lib.qf_status_string(0) == b"ok"
    assert b"invalid" in lib.qf_status_string(-1)
    for log2_T, pls in [(19, ofields.ngp_per_level_scale(4096, 16, 16)), (21, ofields.ngp_per_level_scale(4096, 16, 16)),
                        (24, ofields.field_per_level_scale(512, 1.5, 16, 16)), (8, ofields.ngp_per_level_scale(4096, 16, 16))]:
        d = _C.make_grid_desc(16, log2_T, 16, pls)
        lv = ofields.grid_levels(16, log2_T, 16, pls)
        assert list(d.offset) == lv.offset and list(d.resolution) == lv.resolution and list(d.scale) == lv.scale
        assert [bool((d.hashed_mask >> l) & 1) for l in range(16)] == lv.hashed
    # table sizes of the reference's three configurations
    assert _C.make_grid_desc(16, 19, 16, ofields.ngp_per_level_scale(4096, 16, 16)).offset[16] == 6299960
    assert _C.make_grid_desc(16, 21, 16, ofields.ngp_per_level_scale(4096, 16, 16)).offset[16] == 22565520
    with pytest.raises(ValueError):
        _C.make_grid_desc(0, 19, 16, 1.5)


def test_no_cpu_fallback():
    """Host tensors are refused: the product path never computes on the CPU."""
    import torch
    from quadraturefields_amd import _C, spc_render
    with pytest.raises(RuntimeError):
        spc_render.mark_pack_boundaries(torch.zeros(4, dtype=torch.long))
    with pytest.raises(RuntimeError):
        _C.ptr(torch.zeros(3))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "quadraturefields_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f.endswith(".hip") or f.endswith(".h"), f


def test_forked_worker_is_refused():
    """SURVEY 8b "Threading": the reference calls its intersector from forked DataLoader workers; a HIP context must not
    be touched there.  The package refuses the call in a forked child with a clear error instead of hanging."""
    import multiprocessing as mp
    from quadraturefields_amd import _C

    def child(q):
        try:
            _C.stream()
            q.put("no error")
        except RuntimeError as e:
            q.put(str(e))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    msg = q.get(timeout=60)
    p.join(timeout=30)
    assert "forked child" in msg and "num_workers=0" in msg


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the header's structs (``_C.Camera``, ``GridDesc``, ``FieldDesc``, ``SGHead``, ``TextureSet``,
    ``FrameJob``) have the C compiler's sizes and field offsets: a C translation unit that includes ``include/qf_hip.h``
    (which must compile as plain C -- it is the FFI boundary) prints them."""
    import shutil
    import subprocess
    from quadraturefields_amd import _C
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pairs = {"qf_camera": _C.Camera, "qf_grid_desc": _C.GridDesc, "qf_field_desc": _C.FieldDesc, "qf_sg_head": _C.SGHead,
             "qf_texture_set": _C.TextureSet, "qf_frame_job": _C.FrameJob}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "qf_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(ROOT, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
