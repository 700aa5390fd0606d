#!/usr/bin/env python
"""Build tuning variants of libgotennet_hip.so (extra -D flags) next to the product library and print, for each, the
command prefix that selects it:   python tools/variants.py name1="-DGN_X=1 -DGN_Y=2" name2="..."
-> gotennet_amd/variants/lib_<name>.so  (travels to the GPU box with the snapshot; select with GN_LIB_PATH)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd.build import PKG, build_library  # noqa: E402

out_dir = os.path.join(PKG, "variants")
os.makedirs(out_dir, exist_ok=True)
jobs = []
only = None
for arg in sys.argv[1:]:
    if arg.startswith("--only="):        # translation units the flags apply to (the rest comes from the default build's object cache)
        only = arg[7:].split(",")
        continue
    name, flags = arg.split("=", 1)
    jobs.append((name, flags.split()))


def one(job):
    name, flags = job
    path = os.path.join(out_dir, f"lib_{name}.so")
    build_library(force=True, extra_flags=flags, out=path, only=only)
    return name, path


with ThreadPoolExecutor(max_workers=4) as ex:
    for name, path in ex.map(one, jobs):
        print(f"{name}: GN_LIB_PATH={os.path.relpath(path)}")
