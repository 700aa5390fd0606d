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


OBJ_CACHE = os.path.join(PKG, "build", "obj")     # git-ignored; objects of the DEFAULT flags, keyed by their inputs' mtimes


def _stamp(src: str) -> str:
    deps = [src] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(PKG, "..", "include", "gotennet_hip.h")]
    return ";".join(f"{os.path.basename(d)}:{os.path.getmtime(d):.3f}" for d in deps) + ";" + " ".join(FLAGS)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, only=None) -> str:
    """``extra_flags`` / ``out``: tuning variants (-DGN_...=n) built next to the product library (tools/variants.py).
    ``only``: basenames of the translation units the extra flags apply to -- the others are linked from the object cache of
    the default build (a variant of one kernel then costs one compile)."""
    if not force and not needs_build() and out == LIB:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    base_flags = [f for f in FLAGS if f != "-shared"]
    srcs = sources()
    os.makedirs(OBJ_CACHE, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="gn_build_") as td:
        jobs, objs = [], []
        for s_ in srcs:
            name = os.path.basename(s_)[:-4]
            variant = bool(extra_flags) and (only is None or os.path.basename(s_) in only)
            if variant:
                obj = os.path.join(td, name + ".o")
                jobs.append((s_, obj, base_flags + list(extra_flags), None))
            else:
                obj, st = os.path.join(OBJ_CACHE, name + ".o"), os.path.join(OBJ_CACHE, name + ".stamp")
                stamp = _stamp(s_)
                fresh = os.path.exists(obj) and os.path.exists(st) and open(st).read() == stamp
                if not fresh or (force and not extra_flags and not os.environ.get("GN_BUILD_CACHE")):   # force = a real rebuild (the driver's build check); GN_BUILD_CACHE=1 keeps fresh objects
                    jobs.append((s_, obj, base_flags, (st, stamp)))
            objs.append(obj)

        def one(job):
            cmd = [hipcc] + job[2] + ["-c", job[0], "-o", job[1]]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            if job[3] is not None:
                with open(job[3][0], "w") as fh:
                    fh.write(job[3][1])

        # one translation unit per worker: the files are independent and the largest takes about a minute
        if jobs:
            with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
                list(pool.map(one, jobs))
        link = [hipcc] + [f for f in FLAGS if f.startswith("--offload-arch")] + ["-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
