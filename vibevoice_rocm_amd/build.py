"""Build the in-tree HIP extension (libvv_hip.so) for gfx950 with hipcc.  `python -m vibevoice_rocm_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("vv_kernels.hip", "vv_gemv_stream.hip", "vv_mfma_gemm.hip", "vv_chain.hip", "vv_block1d.hip", "vv_model.hip")]
HDR = [os.path.join(ROOT, "include", "vv_hip.h"), os.path.join(HERE, "csrc", "vv_common.h")]
OUT = os.path.join(HERE, "libvv_hip.so")


def find_hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the MI355X kernels)")


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(p) <= t for p in SRC + HDR)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    cmd = [find_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"), *SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
