"""Build the in-tree HIP extension (libvv_hip.so) for gfx950 with hipcc.  `python -m vibevoice_rocm_amd.build`."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("vv_kernels.hip", "vv_gemv_stream.hip", "vv_gemv_rows.hip", "vv_mfma_gemm.hip", "vv_block1d.hip", "vv_convffn.hip",
                                               "vv_fused.hip", "vv_attn_decode.hip", "vv_attn_prefill.hip", "vv_model.hip")]
HDR = [os.path.join(ROOT, "include", "vv_hip.h"), os.path.join(HERE, "csrc", "vv_common.h")]
OUT = os.path.join(HERE, "libvv_hip.so")
OBJDIR = os.path.join(HERE, "csrc", "build")
STAMP = os.path.join(HERE, "libvv_hip.stamp")      # next to the library: travels with it to the GPU box (git-ignored)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def find_hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the MI355X kernels)")


def _flavour() -> str:
    """What the library was built from: the translation-unit list and the compiler flags (a changed list or flag set rebuilds)."""
    return hashlib.sha256("\n".join([os.path.basename(s) for s in SRC] + FLAGS).encode()).hexdigest()[:16]


def _check_sources():
    missing = [p for p in SRC + HDR if not os.path.exists(p)]
    if missing:
        raise RuntimeError("missing kernel sources: " + ", ".join(os.path.relpath(p, ROOT) for p in missing))


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    if not all(os.path.exists(p) for p in SRC + HDR):
        return True          # a box that received the prebuilt library without the sources (nothing to rebuild from)
    t = os.path.getmtime(OUT)
    if not all(os.path.getmtime(p) <= t for p in SRC + HDR):
        return False
    try:
        return open(STAMP).read().strip() == _flavour()
    except OSError:
        return False


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.basename(src) + ".o")


def build(force: bool = False, verbose: bool = True, jobs: int = 6) -> str:
    """One object per translation unit (compiled in parallel, rebuilt only when the unit or a header changed), then one link."""
    if not force and up_to_date():
        return OUT
    _check_sources()
    os.makedirs(OBJDIR, exist_ok=True)
    flags = [*FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]
    hipcc = find_hipcc()
    hdr_t = max(os.path.getmtime(p) for p in HDR)
    todo = [s for s in SRC if force or not os.path.exists(_obj(s)) or os.path.getmtime(_obj(s)) < max(os.path.getmtime(s), hdr_t)]
    procs = []
    for s in todo:
        cmd = [hipcc, *flags, "-c", s, "-o", _obj(s)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        while sum(p.poll() is None for _, p in procs) >= jobs:
            next(p for _, p in procs if p.poll() is None).wait()
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(os.path.basename(f) for f in failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in SRC], "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(_flavour() + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
