"""Build libgotennet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libgotennet_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-variable"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "gotennet_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB) -> str:
    """``extra_flags`` / ``out``: tuning variants (-DGN_...=n) built next to the product library (tools/variants.py)."""
    if not force and not needs_build() and out == LIB:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    compile_flags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    srcs = sources()
    with tempfile.TemporaryDirectory(prefix="gn_build_") as td:
        objs = [os.path.join(td, os.path.basename(s_)[:-4] + ".o") for s_ in srcs]

        def one(job):
            cmd = [hipcc] + compile_flags + ["-c", job[0], "-o", job[1]]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)

        # one translation unit per worker: the files are independent and the largest takes about a minute
        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
            list(pool.map(one, zip(srcs, objs)))
        link = [hipcc] + [f for f in FLAGS if f.startswith("--offload-arch")] + ["-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
