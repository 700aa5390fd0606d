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
    """Reference layers.py:225-273: ``n_layers - 1`` activated Dense layers and a linear output layer; ``n_hidden`` an
    int (every hidden layer), a list, or None (pyramid: each layer half the width of the one before)."""

    def __init__(self, n_in, n_out, n_hidden=None, n_layers=2, activation=shifted_softplus):
        super().__init__()
        if n_layers < 1:
            raise ValueError("n_layers >= 1")
        if n_hidden is None:
            neurons, c = [], n_in
            for _ in range(n_layers):
                neurons.append(c)
                c = c // 2
            neurons.append(n_out)
        else:
            hidden = [n_hidden] * (n_layers - 1) if isinstance(n_hidden, int) else list(n_hidden)
            neurons = [n_in] + hidden + [n_out]
        self.n_neurons = neurons
        layers = [Dense(neurons[i], neurons[i + 1], activation=activation) for i in range(n_layers - 1)]
        layers.append(Dense(neurons[-2], neurons[-1], activation=None))
        self.out_net = nn.Sequential(*layers)


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
        if n_out != 1 or outnet is not None or return_vector:
            raise NotImplementedError("accelerated Atomwise: n_out=1, default out_net, no return_vector")
        if aggregation_mode not in ("sum", "add", "mean", None):
            raise NotImplementedError(f"aggregation_mode={aggregation_mode!r}: 'sum', 'mean' or None on the accelerated path")
        self.aggregation_mode = aggregation_mode
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
        allowed inside a hipGraph capture) and the transposed hidden-layer weights, rebuilt when any parameter changes."""
        layers = list(self.out_net[1].out_net)
        ts = [t for d in layers for t in (d.weight, d.bias)]
        if isinstance(self.standardize, ScaleShift):
            ts += [self.standardize.stddev, self.standardize.mean]
        key = tuple((t._version, t.data_ptr()) for t in ts)
        c = getattr(self, "_cache", None)
        if c is None or c["key"] != key:
            scale, shift = 1.0, 0.0
            if isinstance(self.standardize, ScaleShift):
                scale, shift = float(self.standardize.stddev[0]), float(self.standardize.mean[0])
            c = dict(key=key, scale=scale, shift=shift, b2=float(layers[-1].bias.detach().cpu()[0]),
                     w=[d.weight.detach() for d in layers], b=[d.bias.detach() for d in layers],
                     wt=[d.weight.detach().t().contiguous() for d in layers[:-1]])
            self._cache = c
        return layers, c

    def _weights(self):
        _, c = self._packed()
        return c["scale"], c["shift"]

    def energy_raw(self, h: torch.Tensor, z32: torch.Tensor, mol_ptr: torch.Tensor, n_mol: int):
        """-> (energy [n_mol,1] (sum or mean over the molecule's atoms), y [N] per-atom contributions, tape).
        ``tape`` (pre-activations of the hidden layers + the aggregation's per-atom weights) goes to ``grad_h_raw``."""
        layers, c = self._packed()
        N = h.shape[0]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=h.device)
        x, pres = h, []
        hidden = layers[:-1]
        for k, d in enumerate(hidden):
            pre = new(N, d.out_features)
            if k + 1 < len(hidden):                  # activated output feeds the next layer; the pre-activation is kept
                xa = new(N, d.out_features)
                engine.gemm(x, d.in_features, c["w"][k], c["b"][k], xa, d.out_features, N, d.out_features, d.in_features,
                            act=(0, d.out_features), pre_out=pre, kind=self.act_kind)
                x = xa
            else:                                    # last hidden layer: gn_head_energy applies the activation itself
                engine.gemm(x, d.in_features, c["w"][k], c["b"][k], pre, d.out_features, N, d.out_features, d.in_features,
                            kind=self.act_kind)
            pres.append(pre)
        last_in = pres[-1] if pres else h.contiguous()
        act = self.act_kind if pres else 11          # GN_ACT_NONE: n_layers = 1, y = W h + b
        y, e = new(N), new(n_mol, 1)
        mean = self.aggregation_mode == "mean"
        atom_scale = new(N) if mean else None
        call("gn_head_energy", ptr(last_in), ptr(c["w"][-1]), c["b2"], c["scale"], c["shift"],
             ptr(self.atomref.weight.detach()) if self.atomref is not None else None, ptr(z32), ptr(mol_ptr),
             n_mol, last_in.shape[1], ptr(y), ptr(e), int(mean), ptr(atom_scale), act, engine._stream())
        return e, y, (pres, last_in, atom_scale)

    def grad_h_raw(self, tape, Fd: int) -> torch.Tensor:
        """d(sum over molecules of the aggregated property)/dh [N,F]."""
        layers, c = self._packed()
        pres, last_in, atom_scale = tape
        N, Hd = last_in.shape
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=last_in.device)
        g = new(N, Hd)
        call("gn_head_grad", ptr(last_in), ptr(c["w"][-1]), c["scale"], ptr(atom_scale), N, Hd, ptr(g),
             self.act_kind if pres else 11, engine._stream())
        for k in range(len(pres) - 1, -1, -1):       # g is d/d(pre_k); through layer k's weight, then act'(pre_{k-1})
            d = layers[k]
            gi = new(N, d.in_features)
            engine.gemm(g, d.out_features, c["wt"][k], None, gi, d.in_features, N, d.in_features, d.out_features,
                        dgate=pres[k - 1] if k > 0 else None, kind=self.act_kind)
            g = gi
        return g

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
        result = {self.property: y}                  # [n_mol,1], or the per-atom values [N,1] for aggregation_mode=None
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
        e, y, tape = head.energy_raw(h.detach().contiguous(), z32, mol_ptr, n_mol)
        head._last_y = y
        ctx.head, ctx.mol_ptr, ctx.n_mol, ctx.F, ctx.tape = head, mol_ptr, n_mol, h.shape[1], tape
        return y.reshape(-1, 1).clone() if head.aggregation_mode is None else e

    @staticmethod
    def backward(ctx, ge):
        gh = ctx.head.grad_h_raw(ctx.tape, ctx.F)
        if ctx.head.aggregation_mode is None:        # per-atom outputs: upstream gradient per atom
            return gh * ge.reshape(-1, 1), None, None, None, None
        # per-molecule upstream gradient (ones for energies -> forces): row scale by ge[molecule]
        mp = ctx.mol_ptr.long()
        per_atom = torch.repeat_interleave(ge.reshape(-1), mp[1:] - mp[:-1])
        return gh * per_atom.unsqueeze(1), None, None, None, None
