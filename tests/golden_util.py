"""Loader for the golden fixtures written by tools/make_golden.py (data only)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith("kat_"))


def load_case(name, dtype=torch.float32):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(bytes(z["cfg"]).decode())
    sd = {k[3:]: torch.from_numpy(z[k]).to(dtype) if z[k].dtype.kind == "f" else torch.from_numpy(z[k])
          for k in z.files if k.startswith("sd/")}
    head = {k[5:]: torch.from_numpy(z[k]).to(dtype) for k in z.files if k.startswith("head/")}
    t = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith(("sd/", "head/")) and k != "cfg"}
    return cfg, sd, head, t


def rel_err(a, b):
    """max|a-b| / max|b|  (per-tensor max-norm relative error)."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
