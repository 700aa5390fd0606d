"""CPU estimate of the error of a candidate projection arithmetic before any kernel is written:
    y = x_hi16 . W_hi16  +  x_8 . W_lo8  +  x_lo8 . W_8        (fp32 / fp64 accumulation)
i.e. the main term of the shipping 2 x fp16 split kept, its two correction terms carried by fp8 (e4m3, OCP) operands, which
the matrix pipe of gfx950 runs at 2.4 x the fp16 rate on random data (profiles/r05_mfma_power_probe.txt).
Every Linear of the oracle (forward AND the input gradient of the force backward) is routed through the emulated product;
energies / forces / (h, X) are compared with the fixture's fp64 truth.  Uses the oracle: a tools/ script, never the product.

    python tools/f8_accuracy_sim.py [case ...] [--modes f16x2,f16f8,...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gotennet_oracle as orc                      # noqa: E402
from tests.golden_util import load_case, rel_err                # noqa: E402

F8 = torch.float8_e4m3fn


def _row_exp(x):
    m = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    return torch.ceil(torch.log2(m))


def _planes(x, per_row, f8_hi_shift=8, f8_lo_shift=20):
    """x (fp64 holding fp32 values) -> scale, hi16, lo (exact remainder), x8, lo8 (all as fp64 in the scaled domain)."""
    e = _row_exp(x) if per_row else torch.ceil(torch.log2(x.abs().max().clamp_min(1e-300)))
    s = torch.pow(torch.tensor(2.0, dtype=torch.float64), e)
    xs = (x / s).float()
    hi = xs.half()
    lo = (xs - hi.float())
    lo16 = lo.half()
    x8 = (xs * 2.0 ** f8_hi_shift).to(F8).double() * 2.0 ** -f8_hi_shift
    lo8 = (lo * 2.0 ** f8_lo_shift).to(F8).double() * 2.0 ** -f8_lo_shift
    xi = torch.round(xs.double() * 127.0).clamp(-127, 127) / 127.0
    li = torch.round(lo.double() * 4096.0 * 127.0).clamp(-127, 127) / (127.0 * 4096.0)
    return s, hi.double(), lo16.double(), x8, lo8, xi, li


def emu_matmul(x, w, mode):
    """x [M,K] . w[N,K]^T in the emulated arithmetic (fp64 accumulation)."""
    if mode == "exact":
        return x @ w.t()
    sx, xh, xl, x8, xl8, xi, xli = _planes(x, per_row=True)
    sw, wh, wl, w8, wl8, wi, wli = _planes(w, per_row=False)
    if mode == "f16x2":
        y = xh @ wh.t() + xh @ wl.t() + xl @ wh.t()
    elif mode == "f16f8":
        y = xh @ wh.t() + x8 @ wl8.t() + xl8 @ w8.t()
    elif mode == "f16f8_wl16":       # weight correction in fp16, activation correction in fp8
        y = xh @ wh.t() + xh @ wl.t() + xl8 @ w8.t()
    elif mode == "f16i8":            # corrections as ONE int8 product over the concatenated depth [x8 | xl8] . [wl8 ; w8]
        y = xh @ wh.t() + xi @ wli.t() + xli @ wi.t()
    elif mode == "f16x1":
        y = xh @ wh.t()
    else:
        raise ValueError(mode)
    return y * sx * sw


class _Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, mode):
        ctx.save_for_backward(w)
        ctx.mode = mode
        shp = x.shape
        y = emu_matmul(x.reshape(-1, shp[-1]).double(), w.double(), mode)
        return y.reshape(*shp[:-1], w.shape[0]).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        shp = g.shape
        gx = emu_matmul(g.reshape(-1, shp[-1]).double(), w.double().t().contiguous(), ctx.mode)
        return gx.reshape(*shp[:-1], w.shape[1]).to(g.dtype), None, None


def run(case, mode):
    cfg, sd, head, t = load_case(case, torch.float64)
    real = torch.nn.functional.linear

    def lin(x, w, b=None):
        y = _Lin.apply(x, w, mode)
        return y if b is None else y + b
    orc.F.linear = lin
    try:
        e, f, (h, X, _) = orc.energy_and_forces(sd, cfg, head, t["z"], t["pos"].double(), t["batch"], cfg["n_mol"])
    finally:
        orc.F.linear = real
    return (rel_err(h, t["h_f64"]), rel_err(X, t["X_f64"]), rel_err(e, t["energy_f64"]), rel_err(f, t["forces_f64"]))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    modes = "exact,f16x2,f16f8,f16i8,f16x1"
    for a in sys.argv[1:]:
        if a.startswith("--modes="):
            modes = a.split("=", 1)[1]
    cases = args or ["c2_model_3mol_seeded"]
    print(f"{'case':28s} {'mode':12s} {'h':>9s} {'X':>9s} {'energy':>9s} {'forces':>9s}   (max-norm relative error vs fp64 truth)")
    for c in cases:
        for m in modes.split(","):
            r = run(c, m)
            print(f"{c:28s} {m:12s} " + " ".join(f"{v:9.2e}" for v in r), flush=True)
