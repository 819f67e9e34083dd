"""In-tree build of libqf_hip.so for gfx950 (hipcc, no cmake): ``python -m quadraturefields_amd.build``.

The shared library is the C-ABI drop-in boundary declared in ``include/qf_hip.h``; it links only
against the HIP runtime.  The built ``.so`` stays inside the package directory so it travels with the
source snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(PKG_DIR, "libqf_hip.so")
ARCH = "gfx950"

# (source, extra flags).  exact.hip carries every integer-deciding comparison: no FMA contraction.
SOURCES = [
    ("field_eval.hip", []),
    ("field_eval_bf16.hip", []),
    ("grid_backward.hip", []),
    ("mlp_train.hip", []),
    ("scan.hip", []),
    ("composite.hip", []),
    ("optim.hip", []),
    ("exact.hip", ["-ffp-contract=off"]),
    ("bvh_build.cpp", ["-x", "hip"]),
    ("misc.cpp", ["-x", "hip"]),
    ("frame.cpp", ["-x", "hip"]),
]
COMMON = os.environ.get("QF_EXTRA_HIPCC_FLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps(src: str):
    deps = [os.path.join(CSRC, src), os.path.join(ROOT, "include", "qf_hip.h"), os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return deps


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(item):
    src, extra = item
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    if not _stale(obj, _deps(src)):
        return obj, False
    cmd = [_hipcc()] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), proc.stderr))
    if proc.stderr.strip():
        sys.stderr.write(proc.stderr)
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libqf_hip.so; returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), proc.stderr))
        if verbose:
            print("linked", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
