"""Build libvsel.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m visionselector_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvsel.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(PKG_DIR), "include", "vsel.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build_native(force: bool = False, verbose: bool = True) -> str:
    """Compile every csrc/*.hip into libvsel.so.  hipcc cross-compiles gfx950 without a GPU."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    build_dir = os.path.join(PKG_DIR, "build")
    os.makedirs(build_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(build_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(p) for p in [src] + glob.glob(os.path.join(CSRC, "*.h"))
                + [os.path.join(os.path.dirname(PKG_DIR), "include", "vsel.h")]):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("VSEL_HIPCC_FLAGS", "").split() + \
            ["-c", src, "-o", obj]
        if verbose:
            print("[vsel build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed ({p.returncode}): {' '.join(cmd)}\n{out}")
        if verbose and out.strip():
            print(out)
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[vsel build]", " ".join(link), flush=True)
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {r.stdout}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
