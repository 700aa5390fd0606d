"""Energy / force read-out on the MI355X path.

``Atomwise`` mirrors the reference head (gotennet/models/components/outputs.py:
323-376, with SchnetMLP layers.py:225-273) for its default shape on this path:
two Dense layers with SiLU, sum aggregation, optional mean/stddev/atomref and
``derivative`` (forces = -dE/dpos).  Parameters keep the reference's state_dict
keys (``out_net.1.out_net.{0,1}.{weight,bias}``, ``standardize.{mean,stddev}``,
``atomref.weight``).  The arithmetic runs in libgotennet_hip.so
(gn_gemm + gn_head_energy; gn_head_grad for the derivative).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine
from ._lib import GotenNetHipError, call, ptr
from .layers import Dense, activation_kind, resolve_activation, shifted_softplus


class _Identity(nn.Module):
    """Placeholder for the reference's ``GetItem('representation')`` at out_net.0 (no parameters)."""


class SchnetMLP(nn.Module):
    def __init__(self, n_in, n_out, n_hidden=None, n_layers=2, activation=F.silu):
        super().__init__()
        if n_layers != 2:
            raise NotImplementedError("the accelerated head implements n_layers=2 (the reference default)")
        n_hidden = n_in // 2 if n_hidden is None else n_hidden
        if not isinstance(n_hidden, int):
            (n_hidden,) = n_hidden
        self.out_net = nn.Sequential(Dense(n_in, n_hidden, activation=activation), Dense(n_hidden, n_out, activation=None))


class ScaleShift(nn.Module):
    def __init__(self, mean, stddev):
        super().__init__()
        self.register_buffer("mean", torch.as_tensor(mean, dtype=torch.float32).reshape(-1))
        self.register_buffer("stddev", torch.as_tensor(stddev, dtype=torch.float32).reshape(-1))


def molecule_ptr(batch: torch.Tensor, n_mol: int) -> torch.Tensor:
    """int32 [n_mol+1] offsets of each molecule in the (sorted) batch vector -- index plumbing."""
    cnt = torch.bincount(batch, minlength=n_mol)
    out = torch.zeros(n_mol + 1, dtype=torch.int32, device=batch.device)
    out[1:] = torch.cumsum(cnt, 0)
    return out


class Atomwise(nn.Module):
    def __init__(self, n_in: int, n_out: int = 1, aggregation_mode: Optional[str] = "sum", n_layers: int = 2,
                 n_hidden: Optional[int] = None, activation=shifted_softplus, property: str = "y",
                 contributions: Optional[str] = None, derivative: Optional[str] = None, negative_dr: bool = True,
                 create_graph: bool = True, mean=None, stddev=None, atomref=None, outnet=None,
                 return_vector: Optional[str] = None, standardize: bool = True):
        super().__init__()
        if n_out != 1 or aggregation_mode != "sum" or outnet is not None or return_vector:
            raise NotImplementedError("accelerated Atomwise: n_out=1, aggregation_mode='sum', default out_net")
        self.act_kind = activation_kind(activation)      # (reference default: shifted_softplus, outputs.py:246)
        activation = resolve_activation(activation)
        self.property, self.contributions, self.derivative = property, contributions, derivative
        self.negative_dr = negative_dr
        self.out_net = nn.Sequential(_Identity(), SchnetMLP(n_in, n_out, n_hidden, n_layers, activation))
        mean = torch.zeros(1) if mean is None else mean
        stddev = torch.ones(1) if stddev is None else stddev
        self.standardize = ScaleShift(mean, stddev) if standardize else nn.Identity()
        self.atomref = nn.Embedding.from_pretrained(atomref.type(torch.float32)) if atomref is not None else None

    # ---- raw (non-autograd) pieces used by the fused pipeline -----------------------
    def invalidate_packed(self):
        """Drop the cached host scalars / transposed weight.  The cache notices updates through autograd's version
        counter and data_ptr; a write through ``param.data`` bumps neither -- call this after such a write."""
        self._cache = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._cache = None
        return out

    def _packed(self):
        """Host copies of the kernel's scalar arguments (a read-back per step would synchronise the stream, and is not
        allowed inside a hipGraph capture) and the transposed first-layer weight, rebuilt when any parameter changes."""
        d0, d1 = self.out_net[1].out_net[0], self.out_net[1].out_net[1]
        ts = [d0.weight, d0.bias, d1.weight, d1.bias]
        if isinstance(self.standardize, ScaleShift):
            ts += [self.standardize.stddev, self.standardize.mean]
        key = tuple((t._version, t.data_ptr()) for t in ts)
        c = getattr(self, "_cache", None)
        if c is None or c["key"] != key:
            scale, shift = 1.0, 0.0
            if isinstance(self.standardize, ScaleShift):
                scale, shift = float(self.standardize.stddev[0]), float(self.standardize.mean[0])
            c = dict(key=key, scale=scale, shift=shift, b2=float(d1.bias.detach().cpu()[0]),
                     w1=d0.weight.detach(), w1t=d0.weight.detach().t().contiguous())
            self._cache = c
        return d0, d1, c

    def _weights(self):
        d0, d1, c = self._packed()
        return d0, d1, c["scale"], c["shift"]

    def energy_raw(self, h: torch.Tensor, z32: torch.Tensor, mol_ptr: torch.Tensor, n_mol: int):
        """-> (energy [n_mol,1], y [N], pre1 [N,Hd])."""
        d0, d1, c = self._packed()
        scale, shift, b2 = c["scale"], c["shift"], c["b2"]
        N, Fd = h.shape
        Hd = d0.out_features
        pre1 = torch.empty((N, Hd), dtype=torch.float32, device=h.device)
        engine.gemm(h, Fd, c["w1"], d0.bias.detach(), pre1, Hd, N, Hd, Fd)
        y = torch.empty(N, dtype=torch.float32, device=h.device)
        e = torch.empty((n_mol, 1), dtype=torch.float32, device=h.device)
        call("gn_head_energy", ptr(pre1), ptr(d1.weight.detach()), b2, scale, shift,
             ptr(self.atomref.weight.detach()) if self.atomref is not None else None, ptr(z32), ptr(mol_ptr),
             n_mol, Hd, ptr(y), ptr(e), self.act_kind, engine._stream())
        return e, y, pre1

    def grad_h_raw(self, pre1: torch.Tensor, Fd: int) -> torch.Tensor:
        """d(sum of energies)/dh [N,F]."""
        d0, d1, c = self._packed()
        N, Hd = pre1.shape
        g1 = torch.empty_like(pre1)
        call("gn_head_grad", ptr(pre1), ptr(d1.weight.detach()), c["scale"], N, Hd, ptr(g1), self.act_kind, engine._stream())
        gh = torch.empty((N, Fd), dtype=torch.float32, device=pre1.device)
        engine.gemm(g1, Hd, c["w1t"], None, gh, Fd, N, Fd, Hd)
        return gh

    # ---- reference-style call --------------------------------------------------------
    def forward(self, inputs):
        """``inputs`` as in the reference: ``.z, .batch, .pos, .representation``.  With
        ``derivative`` set, ``inputs.pos`` must be the leaf the representation was computed
        from with requires_grad (goten_model.py:580-588)."""
        h = inputs["representation"] if isinstance(inputs, dict) else inputs.representation
        z = inputs["z"] if isinstance(inputs, dict) else inputs.z
        batch = inputs["batch"] if isinstance(inputs, dict) else inputs.batch
        pos = inputs["pos"] if isinstance(inputs, dict) else inputs.pos
        if not h.is_cuda:
            raise GotenNetHipError("gotennet_amd.outputs.Atomwise runs on a ROCm device only")
        n_mol = int(batch[-1].item()) + 1 if batch.numel() else 0
        y = _AtomwiseFn.apply(h, self, z.to(torch.int32), molecule_ptr(batch, n_mol), n_mol)
        result = {self.property: y}
        if self.contributions:
            result[self.contributions] = self._last_y.reshape(-1, 1)
        if self.derivative:
            sign = -1.0 if self.negative_dr else 1.0
            (dy,) = torch.autograd.grad(outputs=y, inputs=[pos], grad_outputs=torch.ones_like(y), retain_graph=True)
            result[self.derivative] = sign * dy
        return result


class _AtomwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, head, z32, mol_ptr, n_mol):
        e, y, pre1 = head.energy_raw(h.detach().contiguous(), z32, mol_ptr, n_mol)
        head._last_y = y
        ctx.head, ctx.mol_ptr, ctx.n_mol, ctx.F = head, mol_ptr, n_mol, h.shape[1]
        ctx.save_for_backward(pre1)
        return e

    @staticmethod
    def backward(ctx, ge):
        (pre1,) = ctx.saved_tensors
        gh = ctx.head.grad_h_raw(pre1, ctx.F)
        # per-molecule upstream gradient (ones for energies -> forces): row scale by ge[molecule]
        mp = ctx.mol_ptr.long()
        per_atom = torch.repeat_interleave(ge.reshape(-1), mp[1:] - mp[:-1])
        return gh * per_atom.unsqueeze(1), None, None, None, None
