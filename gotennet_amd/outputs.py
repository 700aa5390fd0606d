"""Read-out heads on the MI355X path: Atomwise (energy / forces), Dipole, ElectronicSpatialExtentV2.

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
    """int32 [n_mol+1] offsets of each molecule in the batch vector -- index plumbing.  Precondition: ``batch`` is
    non-decreasing (each molecule's atoms contiguous: what a PyG ``Batch`` carries and ``radius_graph(batch=batch)`` assumes).
    An unsorted vector gives wrong per-molecule sums (in-bounds by construction, gn_graph.hip); ``EnergyForces(check_edges=True)``
    validates it."""
    out = torch.empty(n_mol + 1, dtype=torch.int32, device=batch.device)
    if batch.is_cuda and batch.dtype == torch.int64:           # one launch, no host read (torch.bincount synchronises)
        call("gn_molecule_ptr", ptr(batch.contiguous()), batch.shape[0], n_mol, ptr(out), engine._stream())
        return out
    cnt = torch.zeros(n_mol, dtype=torch.int32, device=batch.device)
    cnt.index_add_(0, batch, torch.ones_like(batch, dtype=torch.int32))
    out[0] = 0
    out[1:] = torch.cumsum(cnt, 0)
    return out


class Atomwise(nn.Module):
    def __init__(self, n_in: int, n_out: int = 1, aggregation_mode: Optional[str] = "sum", n_layers: int = 2,
                 n_hidden: Optional[int] = None, activation=shifted_softplus, property: str = "y",
                 contributions: Optional[str] = None, derivative: Optional[str] = None, negative_dr: bool = True,
                 create_graph: bool = True, mean=None, stddev=None, atomref=None, outnet=None,
                 return_vector: Optional[str] = None, standardize: bool = True):
        super().__init__()
        if n_out < 1 or outnet is not None or return_vector:
            raise NotImplementedError("accelerated Atomwise: the default out_net (SchnetMLP), no return_vector")
        #: ``n_out`` properties per atom (reference outputs.py:241, SchnetMLP(n_in, n_out, ...)): the last layer has n_out rows;
        #: each is one gn_head_energy launch on the shared hidden activations, the derivative is that of the SUM of the outputs
        #: (``grad_outputs=torch.ones_like(y)``, outputs.py:365-375)
        self.n_out = int(n_out)
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
            if getattr(self, "n_out", 1) > 1:        # per-output host scalars and weight rows (n_out launches share the hidden layers)
                n_out = self.n_out
                per = lambda t: [float(v) for v in t.detach().reshape(-1).cpu().expand(n_out).tolist()]
                sc_l, sh_l = [1.0] * n_out, [0.0] * n_out
                if isinstance(self.standardize, ScaleShift):
                    sc_l, sh_l = per(self.standardize.stddev), per(self.standardize.mean)
                w_last = layers[-1].weight.detach()
                c.update(scales=sc_l, shifts=sh_l, b2s=[float(v) for v in layers[-1].bias.detach().cpu().tolist()],
                         w_rows=[w_last[o].contiguous() for o in range(n_out)],
                         # d(sum of the outputs)/d(last hidden activation): sum_o stddev_o W_o
                         w_sum=(w_last * torch.tensor(sc_l, dtype=w_last.dtype, device=w_last.device).unsqueeze(1)).sum(0).contiguous(),
                         atomref_cols=([self.atomref.weight.detach()[:, o].contiguous() for o in range(n_out)]
                                       if self.atomref is not None else None))
            self._cache = c
        return layers, c

    def _weights(self):
        _, c = self._packed()
        return c["scale"], c["shift"]

    def energy_raw(self, h: torch.Tensor, z32: torch.Tensor, mol_ptr: torch.Tensor, n_mol: int, raw: bool = False,
                   mode: Optional[str] = None):
        """-> (energy [n_mol,1] (sum or mean over the molecule's atoms), y [N] per-atom contributions, tape).
        ``tape`` (pre-activations of the hidden layers + the aggregation's per-atom weights) goes to ``grad_h_raw``.
        ``raw``: y = the MLP output itself (no standardisation, no atomref) -- what ElectronicSpatialExtentV2 reads."""
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
                            act=(0, d.out_features), pre_out=pre, kind=self.act_kind, mode=mode)
                x = xa
            else:                                    # last hidden layer: gn_head_energy applies the activation itself
                engine.gemm(x, d.in_features, c["w"][k], c["b"][k], pre, d.out_features, N, d.out_features, d.in_features,
                            kind=self.act_kind, mode=mode)
            pres.append(pre)
        last_in = pres[-1] if pres else h.contiguous()
        act = self.act_kind if pres else 11          # GN_ACT_NONE: n_layers = 1, y = W h + b
        mean = self.aggregation_mode == "mean"
        atom_scale = new(N) if mean else None
        if getattr(self, "n_out", 1) > 1:
            # n_out properties: one launch per output row of the last layer on the shared hidden activations
            n_out = self.n_out
            y_t, e_t = new(n_out, N), new(n_out, n_mol)
            for o in range(n_out):
                call("gn_head_energy", ptr(last_in), ptr(c["w_rows"][o]), c["b2s"][o], 1.0 if raw else c["scales"][o],
                     0.0 if raw else c["shifts"][o], 0.0,
                     ptr(c["atomref_cols"][o]) if (c["atomref_cols"] is not None and not raw) else None, ptr(z32), ptr(mol_ptr),
                     n_mol, last_in.shape[1], ptr(y_t[o]), ptr(e_t[o]), int(mean), ptr(atom_scale), act, engine._stream())
            return e_t.t().contiguous(), y_t.t().contiguous(), (pres, last_in, atom_scale)
        y, e = new(N), new(n_mol, 1)
        call("gn_head_energy", ptr(last_in), ptr(c["w"][-1]), c["b2"], 1.0 if raw else c["scale"], 0.0 if raw else c["shift"],
             0.0 if raw else c.get("mol_shift", 0.0),
             ptr(self.atomref.weight.detach()) if (self.atomref is not None and not raw) else None, ptr(z32), ptr(mol_ptr),
             n_mol, last_in.shape[1], ptr(y), ptr(e), int(mean), ptr(atom_scale), act, engine._stream())
        return e, y, (pres, last_in, atom_scale)

    def grad_h_raw(self, tape, Fd: int, mode: Optional[str] = None, upstream: Optional[torch.Tensor] = None) -> torch.Tensor:
        """d(sum over molecules AND outputs of the aggregated property)/dh [N,F].  ``upstream`` [N, n_out] (n_out > 1 only): a
        per-atom, per-output weight of that sum (an autograd caller's ``grad_output`` spread to the atoms)."""
        layers, c = self._packed()
        pres, last_in, atom_scale = tape
        N, Hd = last_in.shape
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=last_in.device)
        g = new(N, Hd)
        act_last = self.act_kind if pres else 11
        if getattr(self, "n_out", 1) > 1 and upstream is not None:
            # sum_o upstream[n, o] stddev_o W_o act'(.): one launch per output with a per-atom scale, summed (linear from here on)
            part = new(N, Hd)
            for o in range(self.n_out):
                a_o = upstream[:, o].contiguous() if atom_scale is None else (upstream[:, o] * atom_scale).contiguous()
                call("gn_head_grad", ptr(last_in), ptr(c["w_rows"][o]), c["scales"][o], ptr(a_o), N, Hd,
                     ptr(g if o == 0 else part), act_last, engine._stream())
                if o:
                    g.add_(part)
        elif getattr(self, "n_out", 1) > 1:
            call("gn_head_grad", ptr(last_in), ptr(c["w_sum"]), 1.0, ptr(atom_scale), N, Hd, ptr(g), act_last, engine._stream())
        else:
            call("gn_head_grad", ptr(last_in), ptr(c["w"][-1]), c["scale"], ptr(atom_scale), N, Hd, ptr(g),
                 act_last, engine._stream())
        for k in range(len(pres) - 1, -1, -1):       # g is d/d(pre_k); through layer k's weight, then act'(pre_{k-1})
            d = layers[k]
            gi = new(N, d.in_features)
            engine.gemm(g, d.out_features, c["wt"][k], None, gi, d.in_features, N, d.in_features, d.out_features,
                        dgate=pres[k - 1] if k > 0 else None, kind=self.act_kind, mode=mode)
            g = gi
        return g

    def _per_atom_property(self, e: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """aggregation_mode=None: the property IS the per-atom contributions."""
        return y.reshape(-1, getattr(self, "n_out", 1)).clone()

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
            result[self.contributions] = self._last_y.reshape(-1, getattr(self, "n_out", 1))
        if self.derivative:
            sign = -1.0 if self.negative_dr else 1.0
            (dy,) = torch.autograd.grad(outputs=y, inputs=[pos], grad_outputs=torch.ones_like(y), retain_graph=True)
            result[self.derivative] = sign * dy
        return result


class AtomwiseV3(Atomwise):
    """Reference ``AtomwiseV3`` (outputs.py:96-229) for ``n_out = 1`` with the default ``out_net``: like ``Atomwise`` but the
    per-atom values are only SCALED (``y_i = MLP(h_i) * stddev [+ atomref]``) and ``mean`` is added AFTER the aggregation
    (``y = agg(y_i) + mean``; with ``aggregation_mode=None`` per atom), and -- as in the reference -- the arithmetic reads the
    constructor's ``mean`` / ``stddev`` ATTRIBUTES, not the ``standardize`` buffers it also registers (those only carry
    the ``state_dict`` keys).  Same kernels: ``gn_head_energy`` with ``shift = 0`` and ``mol_shift = mean``."""

    def __init__(self, n_in: int, n_out: int = 1, aggregation_mode: Optional[str] = "sum", n_layers: int = 2,
                 n_hidden: Optional[int] = None, activation=shifted_softplus, property: str = "y",
                 contributions: Optional[str] = None, derivative: Optional[str] = None, negative_dr: bool = True,
                 create_graph: bool = True, mean=None, stddev=None, atomref=None, outnet=None,
                 return_vector: Optional[str] = None, standardize: bool = True):
        if n_out != 1:
            raise NotImplementedError("accelerated AtomwiseV3: n_out=1")
        mean = 0.0 if mean is None else mean
        stddev = 1.0 if stddev is None else stddev
        as_t = lambda v: v if isinstance(v, torch.Tensor) else torch.tensor([float(v)])
        super().__init__(n_in, n_out, aggregation_mode, n_layers, n_hidden, activation, property, contributions,
                         derivative, negative_dr, create_graph, as_t(mean), as_t(stddev), atomref, outnet, return_vector,
                         standardize)
        self.mean, self.stddev = mean, stddev

    def _packed(self):
        layers, c = super()._packed()
        f = lambda v: float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v)
        c = dict(c, scale=f(self.stddev), shift=0.0, mol_shift=f(self.mean))
        return layers, c

    def energy_raw(self, h, z32, mol_ptr, n_mol, raw: bool = False, mode: Optional[str] = None):
        if self.aggregation_mode is None and not raw:        # y_n + mean per atom: every atom is its own segment
            N = h.shape[0]
            mol_ptr, n_mol = torch.arange(N + 1, dtype=torch.int32, device=h.device), N
        return super().energy_raw(h, z32, mol_ptr, n_mol, raw=raw, mode=mode)

    def _per_atom_property(self, e, y):
        return e.reshape(-1, 1)


class _AtomwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, head, z32, mol_ptr, n_mol):
        e, y, tape = head.energy_raw(h.detach().contiguous(), z32, mol_ptr, n_mol)
        head._last_y = y
        ctx.head, ctx.mol_ptr, ctx.n_mol, ctx.F, ctx.tape = head, mol_ptr, n_mol, h.shape[1], tape
        return head._per_atom_property(e, y) if head.aggregation_mode is None else e

    @staticmethod
    def backward(ctx, ge):
        if getattr(ctx.head, "n_out", 1) > 1:        # [n_mol or N, n_out] upstream: per-atom, per-output weights of the sum
            ge = ge.reshape(-1, ctx.head.n_out).to(torch.float32)
            if ctx.head.aggregation_mode is not None:
                mp = ctx.mol_ptr.long()
                ge = torch.repeat_interleave(ge, mp[1:] - mp[:-1], dim=0)
            return ctx.head.grad_h_raw(ctx.tape, ctx.F, upstream=ge.contiguous()), None, None, None, None
        gh = ctx.head.grad_h_raw(ctx.tape, ctx.F)
        if ctx.head.aggregation_mode is None:        # per-atom outputs: upstream gradient per atom
            return gh * ge.reshape(-1, 1), None, None, None, None
        # per-molecule upstream gradient (ones for energies -> forces): row scale by ge[molecule]
        mp = ctx.mol_ptr.long()
        per_atom = torch.repeat_interleave(ge.reshape(-1), mp[1:] - mp[:-1])
        return gh * per_atom.unsqueeze(1), None, None, None, None


# ---------------------------------------------------------------------------------------------------------------------
# vector-representation read-outs of the QM9 task (reference models/tasks/QM9Task.py:168-187)
def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class GatedEquivariantBlock(nn.Module):
    """Reference outputs.py:24-93 (state_dict keys ``mix_vectors.weight``, ``scalar_net.{0,1}.{weight,bias}``).
    ``forward(scalars [N, n_sin], vectors [N, 3, n_vin]) -> (s_out [N, n_sout], v_out [N, 3, n_vout])``; ``vectors`` may be
    the ``X[:, :3, :]`` view of the [N, D, F] vector representation.  Inference only."""

    def __init__(self, n_sin: int, n_vin: int, n_sout: int, n_vout: int, n_hidden: int, activation=F.silu,
                 sactivation=None):
        super().__init__()
        if n_vin % 4 or n_hidden % 4:
            raise NotImplementedError("GatedEquivariantBlock: n_vin and n_hidden must be multiples of 4 on the HIP path")
        self.n_sin, self.n_vin, self.n_sout, self.n_vout, self.n_hidden = n_sin, n_vin, n_sout, n_vout, n_hidden
        self.act_kind = activation_kind(activation)
        self.sact_kind = activation_kind(sactivation) if sactivation is not None else -1
        self.mix_vectors = Dense(n_vin, 2 * n_vout, activation=None, bias=False)
        self.scalar_net = nn.Sequential(Dense(n_sin + n_vout, n_hidden, activation=resolve_activation(activation)),
                                        Dense(n_hidden, n_sout + n_vout, activation=None))
        self.sactivation = resolve_activation(sactivation) if sactivation is not None else None

    def invalidate_packed(self):
        self._cache = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._cache = None
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._cache = None
        return out

    def _packed(self):
        """Weights padded to the GEMM's multiples of four (zero rows / columns): V block at column 0 and W block at column
        pad4(n_vout) of the mixing product; ctx columns and output rows padded likewise."""
        ts = [self.mix_vectors.weight, self.scalar_net[0].weight, self.scalar_net[0].bias,
              self.scalar_net[1].weight, self.scalar_net[1].bias]
        key = tuple((t._version, t.data_ptr()) for t in ts)
        c = getattr(self, "_cache", None)
        if c is None or c["key"] != key:
            nv, ns, no, nh = self.n_vout, self.n_sin, self.n_sout, self.n_hidden
            pv, kc, po = _pad4(nv), _pad4(ns + nv), _pad4(no + nv)
            dev = ts[0].device
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
            wmix = z(2 * pv, self.n_vin)
            wmix[:nv], wmix[pv:pv + nv] = ts[0].detach()[:nv], ts[0].detach()[nv:]
            w0 = z(nh, kc)
            w0[:, :ns + nv] = ts[1].detach()
            w1, b1 = z(po, nh), z(po)
            w1[:no + nv], b1[:no + nv] = ts[3].detach(), ts[4].detach()
            c = dict(key=key, wmix=wmix, w0=w0, b0=ts[2].detach().contiguous(), w1=w1, b1=b1, pv=pv, kc=kc, po=po)
            self._cache = c
        return c

    def forward(self, scalars: torch.Tensor, vectors: torch.Tensor):
        if not scalars.is_cuda:
            raise GotenNetHipError("gotennet_amd.outputs.GatedEquivariantBlock runs on a ROCm device only")
        c = self._packed()
        N = scalars.shape[0]
        if vectors.shape[1] != 3 or vectors.shape[2] != self.n_vin:
            raise ValueError(f"vectors must be [N, 3, {self.n_vin}]")
        # the X[:, :3, :] view of [N, D, F] is gathered into [N, 3, F] by a device copy (index plumbing: the GEMM's row map
        # addresses input AND output rows, and the mixing product's output is dense)
        vectors = vectors.detach().to(torch.float32).contiguous()
        scalars = scalars.detach().to(torch.float32)
        if scalars.stride(-1) != 1:
            scalars = scalars.contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=scalars.device)
        pv, kc, po, nh = c["pv"], c["kc"], c["po"], self.n_hidden
        vmix, ctx, hid, x = new(N * 3, 2 * pv), new(N, kc), new(N, nh), new(N, po)
        engine.gemm(vectors, self.n_vin, c["wmix"], None, vmix, 2 * pv, N * 3, 2 * pv, self.n_vin, kind=self.act_kind)
        call("gn_geb_context", ptr(scalars), scalars.stride(0), self.n_sin, ptr(vmix), 2 * pv, self.n_vout, N, ptr(ctx), kc,
             engine._stream())
        engine.gemm(ctx, kc, c["w0"], c["b0"], hid, nh, N, nh, kc, act=(0, nh), kind=self.act_kind)
        engine.gemm(hid, nh, c["w1"], c["b1"], x, po, N, po, nh, kind=self.act_kind)
        s_out, v_out = new(N, self.n_sout), new(N, 3, self.n_vout)
        call("gn_geb_gate", ptr(x), po, self.n_sout, self.n_vout, ptr(vmix), 2 * pv, pv, N, self.sact_kind,
             ptr(s_out), self.n_sout, ptr(v_out), self.n_vout, engine._stream())
        return s_out, v_out


def _field(inputs, name):
    return inputs[name] if isinstance(inputs, dict) else getattr(inputs, name)


class Dipole(nn.Module):
    """Reference outputs.py:379-468: two GatedEquivariantBlocks on (h, X[:, :3]) -> atomic dipoles + charges,
    ``y = sum_atoms (mu_n + pos_n q_n)`` per molecule (its norm with ``predict_magnitude``).  ``mean`` / ``stddev`` are
    plain attributes like in the reference (not in the state_dict).  Inference only (the QM9 task takes no derivative)."""

    def __init__(self, n_in: int, n_hidden: Optional[int] = None, activation=F.silu, property: str = "dipole",
                 predict_magnitude: bool = False, output_v: bool = True, mean=None, stddev=None):
        super().__init__()
        self.stddev, self.mean, self.output_v = stddev, mean, output_v
        n_hidden = n_in if n_hidden is None else n_hidden
        self.property, self.derivative, self.predict_magnitude = property, None, predict_magnitude
        self.equivariant_layers = nn.ModuleList([
            GatedEquivariantBlock(n_sin=n_in, n_vin=n_in, n_sout=n_hidden, n_vout=n_hidden, n_hidden=n_hidden,
                                  activation=activation, sactivation=activation),
            GatedEquivariantBlock(n_sin=n_hidden, n_vin=n_hidden, n_sout=1, n_vout=1, n_hidden=n_hidden,
                                  activation=activation)])
        self.requires_dr = self.requires_stress = False
        self.aggregation_mode = "sum"

    def forward(self, inputs):
        pos, batch = _field(inputs, "pos"), _field(inputs, "batch")
        l0 = _field(inputs, "representation")
        l1 = _field(inputs, "vector_representation")[:, :3, :]
        for layer in self.equivariant_layers:
            l0, l1 = layer(l0, l1)
        n_mol = int(batch[-1].item()) + 1 if batch.numel() else 0
        mp = molecule_ptr(batch, n_mol)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=l0.device)
        y = new(n_mol, 1) if self.predict_magnitude else new(n_mol, 3)
        y_vec = new(n_mol, 3, 1) if self.output_v else None
        std = self.stddev is not None
        call("gn_dipole_reduce", ptr(l1), 1, ptr(l0), 1, ptr(pos.detach().to(torch.float32).contiguous()), ptr(mp), n_mol,
             float(self.stddev) if std else 1.0, float(self.mean) if std else 0.0, int(std), int(self.predict_magnitude),
             ptr(y), ptr(y_vec), engine._stream())
        result = {self.property: y}
        if self.output_v:
            result[self.property + "_vector"] = y_vec
        return result


#: standard atomic weights by atomic number (index 0: the dummy element, weight 1, as in ase.data.atomic_masses, which
#: the reference reads at outputs.py:513).  Elements up to Kr are built in; heavier ones come from the ``atomic_mass``
#: argument or from a reference checkpoint (the buffer is part of the state_dict).
_ATOMIC_MASS = [1.0, 1.008, 4.002602, 6.94, 9.0121831, 10.81, 12.011, 14.007, 15.999, 18.998403163, 20.1797,
                22.98976928, 24.305, 26.9815385, 28.085, 30.973761998, 32.06, 35.45, 39.948, 39.0983, 40.078,
                44.955908, 47.867, 50.9415, 51.9961, 54.938044, 55.845, 58.933194, 58.6934, 63.546, 65.38, 69.723,
                72.630, 74.921595, 78.971, 79.904, 83.798]


class ElectronicSpatialExtentV2(Atomwise):
    """Reference outputs.py:471-545: ``y = sum_atoms |pos_n - c|^2 x_n`` with ``x = out_net(h)`` (the raw MLP output:
    the reference does not standardise here) and ``c`` the mass-weighted centroid of the molecule."""

    def __init__(self, n_in: int, n_layers: int = 2, n_hidden: Optional[int] = None, activation=shifted_softplus,
                 property: str = "y", contributions: Optional[str] = None, mean=None, stddev=None, outnet=None,
                 atomic_mass: Optional[torch.Tensor] = None):
        super().__init__(n_in, 1, "sum", n_layers, n_hidden, activation=activation, mean=mean, stddev=stddev,
                         outnet=outnet, property=property, contributions=contributions)
        if atomic_mass is None:
            atomic_mass = torch.zeros(119)
            atomic_mass[:len(_ATOMIC_MASS)] = torch.tensor(_ATOMIC_MASS)
        self.register_buffer("atomic_mass", torch.as_tensor(atomic_mass, dtype=torch.float32))

    def forward(self, inputs):
        h, z, batch, pos = (_field(inputs, k) for k in ("representation", "z", "batch", "pos"))
        if not h.is_cuda:
            raise GotenNetHipError("gotennet_amd.outputs.ElectronicSpatialExtentV2 runs on a ROCm device only")
        n_mol = int(batch[-1].item()) + 1 if batch.numel() else 0
        if z.numel():
            # the built-in table stops at Kr (the reference reads ase's full table): an element without a mass would
            # silently shift the mass-weighted centroid (and an all-massless molecule divides 0 by 0)
            zc = z.long().clamp(0, self.atomic_mass.numel() - 1)
            if bool(((self.atomic_mass[zc] <= 0) | (z.long() != zc)).any()):
                raise ValueError("ElectronicSpatialExtentV2: an atomic number has no entry in `atomic_mass` (built-in "
                                 "table: Z <= 36); pass atomic_mass= or load a reference checkpoint that carries it")
        mp, z32 = molecule_ptr(batch, n_mol), z.to(torch.int32)
        _, x, _ = self.energy_raw(h.detach().contiguous(), z32, mp, n_mol, raw=True)
        y = torch.empty((n_mol, 1), dtype=torch.float32, device=h.device)
        call("gn_ese_reduce", ptr(x), ptr(pos.detach().to(torch.float32).contiguous()), ptr(z32), ptr(self.atomic_mass),
             self.atomic_mass.numel(), ptr(mp), n_mol, ptr(y), engine._stream())
        result = {self.property: y}
        if self.contributions:
            result[self.contributions] = x.reshape(-1, 1)
        return result
