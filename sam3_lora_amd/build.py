"""Build the in-tree HIP shared library (gfx950 only).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
SRC = os.path.join(PKG_DIR, "csrc", "lora_kernels.hip")
SRCS = [SRC, os.path.join(PKG_DIR, "csrc", "vit_kernels.hip"), os.path.join(PKG_DIR, "csrc", "loss_kernels.hip"),
        os.path.join(PKG_DIR, "csrc", "fp8_kernels.hip"), os.path.join(PKG_DIR, "csrc", "seg_kernels.hip")]
INCLUDE = os.path.join(REPO_DIR, "include")
LIB_NAME = "libsam3_lora_amd.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # every source the library is compiled from: the .hip units, the .inc files they include, the public headers
    deps = SRCS + sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.inc"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h")))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> sam3_lora_amd/libsam3_lora_amd.so for gfx950."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", INCLUDE] + SRCS + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
