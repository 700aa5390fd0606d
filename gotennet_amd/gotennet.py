"""MI355X-native GotenNet representation: same constructor, forward signature and
state_dict as the reference (gotennet/models/representation/gotennet.py: GATA
77-657, EQFF 660-748, GotenNet 751-1010, GotenNetWrapper 1013-1045), with the
whole forward executed by the hand-written HIP kernels in libgotennet_hip.so.

The nn.Modules below only *hold parameters under the reference's names*; there
is no eager-PyTorch or CPU implementation of the arithmetic in this package.
Calling ``forward`` on CPU tensors, or without the built library, raises.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Mapping, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import embed, engine
from ._lib import GotenNetHipError
from .layers import (BASIS_CODE, MLP, CosineCutoff, Dense, EdgeInit, NodeInit, activation_kind,
                     get_weight_init_by_string, resolve_activation, str2basis)


class _RepresentationFn(torch.autograd.Function):
    """(edge_diff, edge_vec) -> (h, X) with the hand-written backward (engine.backward), so a
    reference-style caller can do torch.autograd.grad(energy, pos) through this module
    (reference outputs.py:365-375 / goten_model.py:580-588)."""

    @staticmethod
    def forward(ctx, edge_diff, edge_vec, net, z32, edge_index):
        cfg, pw = net.config(), net.packed_weights()
        g = engine.Graph(cfg, pw, z32.shape[0], edge_index, edge_diff.detach(), edge_vec.detach())
        h, X, tape = engine.forward(cfg, pw, z32, g, save=True)
        ctx.state = (cfg, pw, z32, g, tape)
        return h, X

    @staticmethod
    def backward(ctx, gh, gX):
        cfg, pw, z32, g, tape = ctx.state
        g_vec, g_diff = engine.backward(cfg, pw, z32, g, tape, gh, gX)
        # (an unsorted edge list was sorted with differentiable indexing in GotenNet.forward, so
        # autograd un-permutes these gradients itself)
        return g_diff, g_vec, None, None, None


class _RepresentationPosFn(torch.autograd.Function):
    """pos -> (h, X) including the radius graph (GotenNetWrapper.forward, gotennet.py:1043-1045)."""

    @staticmethod
    def forward(ctx, pos, net, z32, batch):
        from .graph import distance
        cfg, pw = net.config(), net.packed_weights()
        edge_index, edge_diff, edge_vec = distance(pos.detach(), batch, net.cutoff, net.max_num_neighbors)
        g = engine.Graph(cfg, pw, z32.shape[0], edge_index, edge_diff, edge_vec)
        h, X, tape = engine.forward(cfg, pw, z32, g, save=True)
        ctx.state = (cfg, pw, z32, g, tape)
        return h, X

    @staticmethod
    def backward(ctx, gh, gX):
        cfg, pw, z32, g, tape = ctx.state
        g_vec, g_diff = engine.backward(cfg, pw, z32, g, tape, gh, gX)
        return engine.pos_gradient(g, g_vec, g_diff, sign=1.0), None, None, None


def _pack_gata(gata) -> engine.LayerWeights:
    """GEMM operands of one GATA layer: projections that share an input are concatenated into one weight."""
    c = lambda *ts: torch.cat([t.detach() for t in ts], dim=0).contiguous()
    d = lambda t: t.detach().contiguous()
    lw = engine.LayerWeights(
        Wn1=c(gata.W_q.weight, gata.W_k.weight, gata.gamma_s[0].weight, gata.gamma_v[0].weight),
        bn1=c(gata.W_q.bias, gata.W_k.bias, gata.gamma_s[0].bias, gata.gamma_v[0].bias),
        Ws2=d(gata.gamma_s[1].weight), bs2=d(gata.gamma_s[1].bias),
        Wv2=d(gata.gamma_v[1].weight), bv2=d(gata.gamma_v[1].bias),
        We=c(gata.W_re.weight, gata.W_rs.weight), be=c(gata.W_re.bias, gata.W_rs.bias))
    if not gata.last_layer and gata.edge_updates:
        dl = gata.gamma_t.dense_layers
        lw.Wt, lw.bt = d(dl[-1].weight), d(dl[-1].bias)
        if len(dl) == 2:                   # "mlp"/"mlpa": hidden layer (+ optional LayerNorm "edge_ln")
            lw.Wt0, lw.bt0 = d(dl[0].weight), d(dl[0].bias)
            if dl[0].norm is not None:
                lw.t_ln_w, lw.t_ln_b = d(dl[0].norm.weight), d(dl[0].norm.bias)
        if gata.update_info["lin_w"]:
            lw.Wedp, lw.bedp = d(gata.W_edp.weight), d(gata.W_edp.bias)
            if gata.update_info["lin_ln"] == 1:
                lw.w_ln_w, lw.w_ln_b = d(gata.gamma_w[0].weight), d(gata.gamma_w[0].bias)
            elif gata.update_info["lin_ln"] == 2:
                lw.w_ln_w, lw.w_ln_b = d(gata.W_edp.norm.weight), d(gata.W_edp.norm.bias)
        lw.Wvq = d(gata.W_vq.weight)
        lw.Wvk = [d(wk.weight) for wk in gata.W_vk] if gata.sep_htr else [d(gata.W_vk.weight)]
    if gata.layernorm_:
        lw.ln_w, lw.ln_b = d(gata.layernorm.weight), d(gata.layernorm.bias)
    if gata.steerable_norm_:
        lw.tln_w = d(gata.tensor_layernorm.weight)
    return lw


def _pack_eqff(eq, lw: Optional[engine.LayerWeights] = None) -> engine.LayerWeights:
    d = lambda t: t.detach().contiguous()
    if lw is None:
        lw = engine.LayerWeights(*([None] * 8))
    lw.Wvu = d(eq.W_vu.weight)
    lw.Wm0, lw.bm0 = d(eq.gamma_m[0].weight), d(eq.gamma_m[0].bias)
    lw.Wm1, lw.bm1 = d(eq.gamma_m[1].weight), d(eq.gamma_m[1].bias)
    return lw


class _LayerPackCache:
    """Packed weights of a stand-alone layer module, rebuilt when a parameter changes (version counter / data_ptr;
    ``invalidate_packed()`` after writes through ``.data``)."""

    def _layer_pack(self, build):
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if getattr(self, "_lp", None) is None or self._lp[0] != key:
            self._lp = (key, build(self))
        return self._lp[1]

    def invalidate_packed(self):
        self._lp = None


def _require_cuda(t: Tensor, what: str):
    if not t.is_cuda:
        raise GotenNetHipError(f"gotennet_amd.{what} runs on a ROCm device only (no CPU fallback; the CPU oracle lives in "
                               "oracle/ for tests)")


class GATA(_LayerPackCache, nn.Module):
    """One GATA layer (reference gotennet.py:78-657): parameters under the reference's names, and ``forward`` with the
    reference's signature driving the HIP kernels (inference only; inside ``GotenNet`` the stack driver sequences the
    same kernels with the neighbouring EQFF launches fused in)."""

    def __init__(self, n_atom_basis: int, activation: Callable, weight_init=nn.init.xavier_uniform_,
                 bias_init=nn.init.zeros_, aggr: str = "add", epsilon: float = 1e-7, layer_norm: str = "",
                 steerable_norm: str = "", cutoff: float = 5.0, num_heads: int = 8, dropout: float = 0.0,
                 edge_updates: Union[bool, str] = True, last_layer: bool = False, scale_edge: bool = True,
                 evec_dim: Optional[int] = None, emlp_dim: Optional[int] = None, sep_htr: bool = True,
                 sep_dir: bool = True, sep_tensor: bool = True, lmax: int = 2, edge_ln: str = ""):
        super().__init__()
        # gotennet.py:139-190: '_'-separated variants of the edge update
        self.update_info = info = {"gated": False, "rej": True, "mlp": False, "mlpa": False, "lin_w": 0, "lin_ln": 0}
        parts = edge_updates.split("_") if isinstance(edge_updates, str) else []
        allowed = ["gated", "gatedt", "norej", "norm", "mlp", "mlpa", "act", "linw", "linwa", "ln", "postln"]
        if not all(p in allowed for p in parts):
            raise ValueError(f"Invalid edge update parts. Allowed parts are {allowed}")
        for p in ("gated", "gatedt", "act"):
            if p in parts:
                info["gated"] = p
        if "norej" in parts:
            info["rej"] = False
        info["mlp"], info["mlpa"] = "mlp" in parts, "mlpa" in parts
        info["lin_w"] = 2 if "linwa" in parts else 1 if "linw" in parts else 0
        info["lin_ln"] = 2 if "postln" in parts else 1 if "ln" in parts else 0
        if aggr not in ("add", "sum", "mean", "max"):           # PyG aggregations of MessagePassing(aggr=...) GATA.aggregate uses
            raise NotImplementedError(f"aggr={aggr!r}: 'add', 'mean' or 'max' (gotennet.py:84,638)")
        self.aggr_kind = {"add": 0, "sum": 0, "mean": 1, "max": 2}[aggr]
        if edge_ln not in ("", None, "layer"):
            raise NotImplementedError(f"edge_ln={edge_ln!r}: only '' and 'layer' are on the accelerated path")
        self.edge_vec_dim = n_atom_basis if evec_dim is None else evec_dim
        self.edge_mlp_dim = n_atom_basis if emlp_dim is None else emlp_dim
        if self.edge_vec_dim != n_atom_basis and not info["lin_w"] and edge_updates:
            raise ValueError("evec_dim != n_atom_basis needs a 'linw'/'linwa' edge update (W_edp maps w_ij back to "
                             "n_atom_basis; the reference fails with a shape error otherwise)")
        if evec_dim is not None and evec_dim != n_atom_basis and (
                self.edge_vec_dim < 16 or self.edge_vec_dim > 1024 or self.edge_vec_dim & (self.edge_vec_dim - 1)):
            raise NotImplementedError("evec_dim (when it differs from n_atom_basis) must be a power of two in [16, 1024] on the "
                                      "HIP path, and at most 256 for forces")
        if self.edge_mlp_dim % 4:
            raise NotImplementedError("emlp_dim must be a multiple of 4 on the HIP path")
        self.layernorm_, self.steerable_norm_ = layer_norm, steerable_norm
        self.n_atom_basis, self.lmax, self.num_heads = n_atom_basis, lmax, num_heads
        self.last_layer, self.edge_updates, self.scale_edge = last_layer, edge_updates, scale_edge
        self.sep_htr, self.sep_dir, self.sep_tensor = sep_htr, sep_dir, sep_tensor
        self.dropout, self.epsilon, self.cutoff = dropout, epsilon, cutoff
        self.act_kind = activation_kind(activation)
        multiplier = 3 + (lmax - 1 if sep_dir else 0) + (lmax - 1 if sep_tensor else 0)
        self.multiplier = multiplier
        D_ = partial(Dense, weight_init=weight_init, bias_init=bias_init)
        self.gamma_s = nn.Sequential(D_(n_atom_basis, n_atom_basis, activation=activation),
                                     D_(n_atom_basis, multiplier * n_atom_basis, activation=None))
        self.W_q = D_(n_atom_basis, n_atom_basis, activation=None)
        self.W_k = D_(n_atom_basis, n_atom_basis, activation=None)
        self.gamma_v = nn.Sequential(D_(n_atom_basis, n_atom_basis, activation=activation),
                                     D_(n_atom_basis, multiplier * n_atom_basis, activation=None))
        self.W_re = D_(n_atom_basis, n_atom_basis, activation=activation)
        if not last_layer and edge_updates:
            two = info["mlp"] or info["mlpa"]       # gotennet.py:238-250
            self.gamma_t = MLP([n_atom_basis, self.edge_mlp_dim, n_atom_basis] if two else [n_atom_basis] * 2,
                               activation=activation,
                               last_activation=None if info["mlp"] else activation,
                               weight_init=weight_init, bias_init=bias_init, norm=edge_ln)
            ev = self.edge_vec_dim
            self.W_vq = D_(n_atom_basis, ev, activation=None, bias=False)
            if sep_htr:
                self.W_vk = nn.ModuleList([D_(n_atom_basis, ev, activation=None, bias=False) for _ in range(lmax)])
            else:
                self.W_vk = D_(n_atom_basis, ev, activation=None, bias=False)
            modules = []                               # gotennet.py:270-291 (same module order => same state_dict keys)
            if info["lin_w"] > 0:
                if info["lin_ln"] == 1:
                    modules.append(nn.LayerNorm(ev))
                if info["lin_w"] == 2:
                    modules.append(nn.SiLU())
                self.W_edp = D_(ev, n_atom_basis, activation=None,
                                norm="layer" if info["lin_ln"] == 2 else "")
                modules.append(self.W_edp)
            gate = {"gatedt": nn.Tanh, "gated": nn.Sigmoid, "act": nn.SiLU}.get(info["gated"])
            if gate is not None:
                modules.append(gate())
            self.gamma_w = nn.Sequential(*modules)
        self.W_rs = D_(n_atom_basis, n_atom_basis * multiplier, activation=None)
        # gotennet.py:305-315
        self.layernorm = nn.LayerNorm(n_atom_basis) if layer_norm != "" else nn.Identity()
        self.tensor_layernorm = TensorLayerNorm(n_atom_basis, trainable=False, lmax=lmax) if steerable_norm != "" \
            else nn.Identity()

    @property
    def htr_mode(self) -> int:
        """``mode`` argument of gn_htr_edge / gn_htr_backward (include/gotennet_hip.h)."""
        gate = 0 if self.composed_update else self.gate_kind
        return (0 if self.sep_htr else 1) | (0 if self.update_info["rej"] else 2) | (gate << 2)

    @property
    def gate_kind(self) -> int:
        return {False: 0, "gated": 1, "gatedt": 2, "act": 3}[self.update_info["gated"]]

    @property
    def composed_update(self) -> bool:
        """gamma_t is a 2-layer MLP and/or gamma_w has the W_edp projection: the edge update is sequenced from
        GEMM / LayerNorm / gate launches by engine._edge_update_composed instead of the fused default."""
        i = self.update_info
        return bool(i["mlp"] or i["mlpa"] or i["lin_w"])

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, (Dense, nn.LayerNorm, TensorLayerNorm)):
                m.reset_parameters()
        self.invalidate_packed()

    def layer_config(self) -> engine.Config:
        return engine.Config(F=self.n_atom_basis, L=1, R=0, H=self.num_heads, lmax=self.lmax, M=self.multiplier,
                             cutoff=float(self.cutoff), eps=float(self.epsilon), scale_edge=bool(self.scale_edge),
                             sep_dir=bool(self.sep_dir), sep_tensor=bool(self.sep_tensor), htr_mode=self.htr_mode,
                             layernorm=bool(self.layernorm_), steerable_norm=bool(self.steerable_norm_),
                             composed_update=self.composed_update, gate_kind=self.gate_kind,
                             t_last_act=0 if self.update_info["mlp"] else 3, lin_w=self.update_info["lin_w"],
                             lin_ln=self.update_info["lin_ln"], evec=self.edge_vec_dim, emlp=self.edge_mlp_dim,
                             act=self.act_kind, gemm_mode=engine.resolve_mode(getattr(self, "gemm_mode", None)),
                             sliced=bool(getattr(self, "sliced_kernels", False)), aggr=self.aggr_kind)

    @torch.no_grad()
    def forward(self, edge_index: Tensor, h: Tensor, X: Tensor, rl_ij: Tensor, t_ij: Tensor, r_ij: Tensor,
                n_edges: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
        """Reference GATA.forward (gotennet.py:366-450): ``h`` [N,1,F] (or [N,F]), ``X`` [N,D,F], ``rl_ij`` [E,D],
        ``t_ij`` [E,F] (or [E,1,F]), ``r_ij`` [E] distances (the cosine cutoff is applied inside, as in ``message``).
        With ``scale_edge`` the kernels normalise by the out-degree of every edge's source, recomputed from
        ``edge_index`` exactly as GotenNet.forward does (gotennet.py:986-989); a caller-supplied ``n_edges`` [E] / [E,1]
        must BE that quantity (checked on the device, ``ValueError`` otherwise: another normalisation has no kernel).
        Any edge order.  No edges: the (optionally normalised) inputs come back unchanged."""
        _require_cuda(h, "GATA")
        if self.training and self.dropout > 0:
            raise NotImplementedError("attention dropout (training mode) is not on the accelerated path; call .eval()")
        if embed.needs_embedding(self.n_atom_basis):
            raise NotImplementedError(f"n_atom_basis={self.n_atom_basis}: a stand-alone GATA layer needs a power-of-two width (the slot kernels "
                                      "tile an edge row over F/4 lanes); inside GotenNet such a model runs embedded in the next one")
        cfg, lw = self.layer_config(), self._layer_pack(_pack_gata)
        hs, ts = h.shape, t_ij.shape
        N, E = h.shape[0], edge_index.shape[1]
        f32c = lambda t: t.to(torch.float32).contiguous()
        F_, D_ = self.n_atom_basis, (self.lmax + 1) ** 2 - 1          # (explicit widths: E may be 0)
        h2, X2, t2 = f32c(h.reshape(N, F_)), f32c(X), f32c(t_ij.reshape(E, F_))
        rl, r = f32c(rl_ij.reshape(E, D_)), f32c(r_ij.reshape(-1))
        edge_index = edge_index.contiguous()
        if E == 0:                                   # no messages, no edge update: only the input norms act (gotennet.py:397-398)
            ho, Xo = engine.gata_input_norms(cfg, lw, h2, X2)
            return ho.clone().reshape(hs) if ho is h2 else ho.reshape(hs), Xo.clone() if Xo is X2 else Xo, t_ij.clone()
        order = None
        if E:
            bits = engine.validate_edges(edge_index, N)
            if bits & 2:
                raise ValueError(f"edge_index holds indices outside [0, {N})")
            if n_edges is not None and self.scale_edge:      # (after the range check: the gather below indexes with edge_index)
                deg = torch.zeros(N, dtype=torch.float32, device=h.device).index_add_(
                    0, edge_index[0], torch.ones(E, dtype=torch.float32, device=h.device))
                if not bool(torch.equal(n_edges.reshape(-1).to(torch.float32), deg[edge_index[0]])):
                    raise ValueError("GATA.forward: n_edges differs from the out-degree of each edge's source "
                                     "(gotennet.py:986-989); the accelerated path implements that normalisation only")
            if bits & 1:
                order = torch.sort(edge_index[1], stable=True).indices
                edge_index, rl, r, t2 = edge_index[:, order].contiguous(), rl[order].contiguous(), r[order].contiguous(), \
                    t2[order].contiguous()
        g = engine.Graph(cfg, None, N, edge_index)
        g.rl = rl
        engine.call("gn_cosine_cutoff", engine.ptr(r), E, float(self.cutoff), engine.ptr(g.cut), engine._stream())
        ho, Xo, to = engine.gata_layer(cfg, lw, g, h2, X2, t2)
        if order is not None and to is not t2:
            inv = torch.empty_like(order)
            inv[order] = torch.arange(E, device=order.device)
            to = to[inv]
        elif order is not None:
            to = f32c(t_ij.reshape(E, -1))
        return ho.reshape(hs), Xo, to.reshape(ts)


class TensorLayerNorm(nn.Module):
    """Buffer container for the reference TensorLayerNorm (layers.py:1497-1527); gn_tensor_norm computes it."""

    def __init__(self, hidden_channels, trainable, lmax=1, **kwargs):
        super().__init__()
        if trainable:
            raise NotImplementedError("trainable TensorLayerNorm")
        self.hidden_channels, self.eps, self.lmax = hidden_channels, 1e-12, lmax
        self.register_buffer("weight", torch.ones(hidden_channels))

    def reset_parameters(self):
        with torch.no_grad():
            self.weight.fill_(1.0)


class EQFF(_LayerPackCache, nn.Module):
    """EQFF block (reference gotennet.py:660-748): parameters under the reference's names; ``forward(h, X)`` drives the
    HIP kernels (inference only)."""

    def __init__(self, n_atom_basis: int, activation: Callable, lmax: int, epsilon: float = 1e-8,
                 weight_init=nn.init.xavier_uniform_, bias_init=nn.init.zeros_):
        super().__init__()
        self.lmax, self.n_atom_basis, self.epsilon = lmax, n_atom_basis, epsilon
        self.act_kind = activation_kind(activation)
        D_ = partial(Dense, weight_init=weight_init, bias_init=bias_init)
        self.gamma_m = nn.Sequential(D_(2 * n_atom_basis, n_atom_basis, activation=activation),
                                     D_(n_atom_basis, 2 * n_atom_basis, activation=None))
        self.W_vu = D_(n_atom_basis, n_atom_basis, activation=None, bias=False)

    def reset_parameters(self):
        self.W_vu.reset_parameters()
        for l in self.gamma_m:
            l.reset_parameters()
        self.invalidate_packed()

    @torch.no_grad()
    def forward(self, h: Tensor, X: Tensor) -> Tuple[Tensor, Tensor]:
        """Reference EQFF.forward (gotennet.py:716-748): ``h`` [N,1,F] (or [N,F]), ``X`` [N,D,F] -> (h', X')."""
        _require_cuda(h, "EQFF")
        N, F_ = h.shape[0], self.n_atom_basis
        D = (self.lmax + 1) ** 2 - 1
        cfg = engine.Config(F=F_, L=1, R=0, H=1, lmax=self.lmax, M=1, cutoff=0.0, eps=float(self.epsilon),
                            scale_edge=False, sep_dir=False, sep_tensor=False, act=self.act_kind,
                            gemm_mode=engine.resolve_mode(getattr(self, "gemm_mode", None)))
        lw = self._layer_pack(_pack_eqff)
        ho, Xo = engine.eqff_layer(cfg, lw, h.reshape(N, F_).to(torch.float32).contiguous(),
                                   X.to(torch.float32).contiguous())
        return ho.reshape(h.shape), Xo


class GotenNet(nn.Module):
    """Drop-in for the reference ``GotenNet`` (gotennet.py:751-1010).

    ``forward(atomic_numbers, edge_index, edge_diff, edge_vec) -> (h [N,F], X [N,D,F])``.
    Differences from the reference, by design: inputs are left untouched (the
    reference normalises ``edge_vec`` in place, 978-980); inference only (attention
    dropout is inactive, as in ``eval()``); fp32; runs on a ROCm device only.
    """

    def __init__(self, n_atom_basis: int = 128, n_interactions: int = 8,
                 radial_basis: Union[Callable, str] = "expnorm", n_rbf: int = 32,
                 cutoff_fn: Optional[Union[Callable, str]] = None,
                 activation: Optional[Union[Callable, str]] = F.silu, max_z: int = 100, epsilon: float = 1e-8,
                 weight_init: Callable = nn.init.xavier_uniform_, bias_init: Callable = nn.init.zeros_,
                 layernorm: str = "", steerable_norm: str = "", num_heads: int = 8, attn_dropout: float = 0.0,
                 edge_updates: Union[bool, str] = True, scale_edge: bool = True, lmax: int = 1, aggr: str = "add",
                 evec_dim: Optional[int] = None, emlp_dim: Optional[int] = None, sep_htr: bool = True,
                 sep_dir: bool = False, sep_tensor: bool = False, edge_ln: str = ""):
        super().__init__()
        self._packed = self._packed_key = self._packed_params = None
        self.scale_edge = scale_edge
        if type(weight_init) == str:
            weight_init = get_weight_init_by_string(weight_init)
        if type(bias_init) == str:
            bias_init = get_weight_init_by_string(bias_init)
        self.act_kind = activation_kind(activation)
        activation = resolve_activation(activation)
        if not 1 <= lmax <= 8:                       # TensorInit is defined up to l = 8 (reference layers.py:805-1494)
            raise NotImplementedError("TensorInit (and the MI355X kernels) cover 1 <= lmax <= 8")

        self.n_atom_basis = self.hidden_dim = n_atom_basis
        self.n_interactions = n_interactions
        self.cutoff_fn = cutoff_fn
        self.cutoff = cutoff_fn.cutoff
        self.lmax, self.num_heads, self.n_rbf, self.epsilon = lmax, num_heads, n_rbf, epsilon
        self.sep_dir, self.sep_tensor = sep_dir, sep_tensor
        self.attn_dropout = attn_dropout

        self.node_init = NodeInit([self.hidden_dim, self.hidden_dim], n_rbf, self.cutoff, max_z=max_z,
                                  weight_init=weight_init, bias_init=bias_init, proj_ln="layer", activation=activation)
        self.edge_init = EdgeInit(n_rbf, self.hidden_dim)
        self.radial_basis = str2basis(radial_basis)(cutoff=self.cutoff, n_rbf=n_rbf)
        self.A_na = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.gata_list = nn.ModuleList([
            GATA(n_atom_basis=n_atom_basis, activation=activation, aggr=aggr, weight_init=weight_init,
                 bias_init=bias_init, layer_norm=layernorm, steerable_norm=steerable_norm, cutoff=self.cutoff,
                 epsilon=epsilon, num_heads=num_heads, dropout=attn_dropout, edge_updates=edge_updates,
                 last_layer=(i == n_interactions - 1), scale_edge=scale_edge, evec_dim=evec_dim,
                 emlp_dim=emlp_dim, sep_htr=sep_htr, sep_dir=sep_dir, sep_tensor=sep_tensor, lmax=lmax,
                 edge_ln=edge_ln)
            for i in range(n_interactions)])
        self.eqff_list = nn.ModuleList([
            EQFF(n_atom_basis=n_atom_basis, activation=activation, lmax=lmax, epsilon=epsilon,
                 weight_init=weight_init, bias_init=bias_init) for _ in range(n_interactions)])
        self.reset_parameters()

        #: set True when the caller guarantees ``edge_index[1]`` is non-decreasing (what
        #: radius_graph emits); skips the device->host sortedness check (one sync).
        self.assume_sorted_edges = False
        #: projection arithmetic of THIS model: None = ``engine.GEMM_MODE`` (the process default), or "f16x2" / "split" /
        #: "f32" (engine.py).  Carried in ``config()``: two models with different arithmetics may run from two threads.
        self.gemm_mode: Optional[str] = None
        #: True: lmax <= 4 runs on the degree-sliced kernel family as well (GN_LMAX_SLICED; tests hold the two families
        #: against each other)
        self.sliced_kernels = False
        #: the EQFF chain after X_p (context, gamma_m, update; and its input-gradient) as one kernel each way where covered
        #: (engine.eqff_fused_ok: F in {128, 256}, SiLU, a plane arithmetic).  None = auto: on for calls of at most
        #: engine.EQFF_FUSED_MAX_ATOMS atoms (a one-molecule step is launch-bound: 2.31 -> 1.91 ms; 32 molecules -2 %), off
        #: above (the 128-molecule batch: 7.735 vs 7.698 ms); True / False force it
        self.fuse_eqff = None
        self._warned_inference_only = False

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self):
        self.node_init.reset_parameters()
        self.edge_init.reset_parameters()
        for l in self.gata_list:
            l.reset_parameters()
        for l in self.eqff_list:
            l.reset_parameters()
        self.invalidate_packed()

    def invalidate_packed(self):
        """Drop the packed / transposed / bf16-split copies of the weights (rebuilt on the next forward).

        ``packed_weights`` notices parameter updates through autograd's version counter (optimizer steps,
        ``load_state_dict``, ``copy_`` / ``fill_`` under ``torch.no_grad()``) and through ``data_ptr`` (``.to()``,
        ``.cuda()``).  A write through ``param.data`` (``p.data.copy_(...)``, EMA weight swaps written that way) bumps
        NEITHER: call this method after such a write.  (A replaced parameter or submodule OBJECT is noticed: every parameter,
        buffer and submodule slot of the module tree is checked for identity on each call.)"""
        self._packed = self._packed_key = self._packed_params = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def _apply(self, fn, *args, **kwargs):              # .to() / .cuda() / .float() ...
        out = super()._apply(fn, *args, **kwargs)
        self._packed = self._packed_key = self._packed_params = None
        return out

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, device="cpu"):
        """Reference gotennet.py:904-946 (Lightning checkpoint -> representation)."""
        import os
        if not os.path.exists(checkpoint_path):
            raise FileNotFoundError(f"Checkpoint file {checkpoint_path} does not exist.")
        ck = torch.load(checkpoint_path, map_location=device)
        if "representation" in ck:
            ck = ck["representation"]
        assert "hyper_parameters" in ck, "Checkpoint must contain 'hyper_parameters' key."
        assert "representation" in ck["hyper_parameters"], "Hyperparameters must contain 'representation' key."
        conf = dict(ck["hyper_parameters"]["representation"])
        conf.pop("_target_", None)
        conf.pop("__target__", None)
        if isinstance(conf.get("cutoff_fn"), Mapping):
            conf["cutoff_fn"] = CosineCutoff(conf["cutoff_fn"]["cutoff"])
        assert "state_dict" in ck, "Checkpoint must contain 'state_dict' key."
        sd = {}
        for k, v in ck["state_dict"].items():
            if k.startswith("output_modules."):
                continue
            sd[k[len("representation."):] if k.startswith("representation.") else k] = v
        net = cls(**conf)
        net.load_state_dict(sd, strict=True)
        return net

    def config(self) -> engine.Config:
        """The kernels' view of the model.  A width that is not a power of two runs embedded in the next one (embed.py):
        ``F`` is then the padded width, ``F_model`` the real one."""
        cfg = self._config_real()
        if embed.needs_embedding(cfg.F):
            embed.check(cfg.F, cfg.H, cfg)
            cfg = embed.embedded_config(cfg)
        return cfg

    def _config_real(self) -> engine.Config:
        g0 = self.gata_list[0]
        return engine.Config(F=self.n_atom_basis, L=self.n_interactions, R=self.n_rbf, H=self.num_heads,
                             lmax=self.lmax, M=g0.multiplier, cutoff=float(self.cutoff), eps=float(self.epsilon),
                             scale_edge=bool(self.scale_edge), sep_dir=bool(self.sep_dir),
                             sep_tensor=bool(self.sep_tensor), basis=BASIS_CODE[type(self.radial_basis)][0],
                             htr_mode=g0.htr_mode, layernorm=bool(g0.layernorm_), steerable_norm=bool(g0.steerable_norm_),
                             composed_update=g0.composed_update, gate_kind=g0.gate_kind,
                             t_last_act=0 if g0.update_info["mlp"] else 3,
                             lin_w=g0.update_info["lin_w"], lin_ln=g0.update_info["lin_ln"],
                             evec=g0.edge_vec_dim, emlp=g0.edge_mlp_dim, act=self.act_kind,
                             gemm_mode=engine.resolve_mode(self.gemm_mode), sliced=bool(self.sliced_kernels),
                             fuse_eqff=self.fuse_eqff, aggr=g0.aggr_kind)

    def _param_slots(self):
        """(owner's ``_parameters`` / ``_buffers`` / ``_modules`` dict, name, object) of every parameter, buffer AND
        submodule slot of the tree: a replaced submodule (``net.gata_list[i] = new_layer``) leaves the old module's own
        dicts intact, so the parent's ``_modules`` slot is what has to be checked for it."""
        out = []
        for m in self.modules():
            out += [(m._parameters, n, t) for n, t in m._parameters.items() if t is not None]
            out += [(m._buffers, n, t) for n, t in m._buffers.items() if t is not None]
            out += [(m._modules, n, c) for n, c in m._modules.items() if c is not None]
        return out

    def packed_is_current(self) -> bool:
        """True when ``packed_weights()`` would return the cached pack (no rebuild, no allocation, no launch)."""
        slots = self._packed_params
        if slots is None or self._packed is None or any(d.get(n) is not t for d, n, t in slots):
            return False
        return self._packed_key == tuple((t.data_ptr(), t._version) for _, _, t in slots if isinstance(t, Tensor))

    def packed_weights(self) -> engine.PackedWeights:
        """Concatenate the projections that share an input into single GEMM operands
        (cached; rebuilt when any parameter is modified or moved)."""
        # (walking the module tree costs 0.3 ms -- a sixth of the host time of a one-molecule eager step: the walk is kept as
        #  (owner dict, name, object) triples over the parameter, buffer and submodule slots.  Every call checks that each
        #  slot still holds THAT object -- a replaced Parameter (torch.func.functional_call, a parent module's
        #  load_state_dict(assign=True), parametrize, ``layer.weight = nn.Parameter(...)``) or a replaced submodule
        #  (``net.gata_list[i] = layer``) fails the identity test and triggers a new walk -- and compares the
        #  addresses / version counters of the kept tensors: O(P) dictionary look-ups, no tree walk)
        slots = self._packed_params
        if slots is None or any(d.get(n) is not t for d, n, t in slots):
            slots = self._packed_params = self._param_slots()
            self._packed = None
        key = tuple((t.data_ptr(), t._version) for _, _, t in slots if isinstance(t, Tensor))
        if self._packed is not None and key == self._packed_key:
            return self._packed
        c = lambda *ts: torch.cat([t.detach() for t in ts], dim=0).contiguous()
        d = lambda t: t.detach().contiguous()
        ni, ei = self.node_init, self.edge_init
        mlp = ni.W_nrd_nru.dense_layers
        pw = engine.PackedWeights(
            A_na=d(self.A_na.weight), A_nbr=d(ni.A_nbr.weight),
            Winit=c(ni.W_ndp.dense_layers[0].weight, ei.W_erp.weight),
            binit=c(ni.W_ndp.dense_layers[0].bias, ei.W_erp.bias),
            Wa=d(mlp[0].weight), ba=d(mlp[0].bias), ln_w=d(mlp[0].norm.weight), ln_b=d(mlp[0].norm.bias),
            Wb=d(mlp[1].weight), bb=d(mlp[1].bias),
            rb0=d(getattr(self.radial_basis, BASIS_CODE[type(self.radial_basis)][1]).float()),
            rb1=d(getattr(self.radial_basis, BASIS_CODE[type(self.radial_basis)][2]).float()))
        for gata, eq in zip(self.gata_list, self.eqff_list):
            lw = _pack_eqff(eq, _pack_gata(gata))
            pw.layers.append(lw)
        if embed.needs_embedding(self.n_atom_basis):    # not a power of two: the equivalent model of the padded width
            real = self._config_real()
            embed.check(real.F, real.H, real)
            pw = embed.embed_pack(pw, real.F, real.H, real.M)
        self._packed, self._packed_key = pw, key
        return pw

    # ------------------------------------------------------------------ forward
    def _check_inputs(self, atomic_numbers, edge_index, edge_diff, edge_vec):
        if not atomic_numbers.is_cuda:
            raise GotenNetHipError("gotennet_amd runs on a ROCm device only: move the module and its inputs to "
                                   "'cuda' (there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
        p = next(self.parameters())
        if p.device != atomic_numbers.device or p.dtype != torch.float32:
            raise GotenNetHipError("module parameters must be fp32 on the same device as the inputs")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be int64 [2, E]")
        E = edge_index.shape[1]
        if edge_diff.shape != (E,):
            raise ValueError("edge_diff must be 1-D [E] (reference layers.py:745 unsqueezes it)")
        if edge_vec.shape != (E, 3):
            raise ValueError("edge_vec must be [E, 3]")
        if self.training and self.attn_dropout > 0:
            raise NotImplementedError("attention dropout (training mode) is not on the accelerated path; call .eval()")

    def _warn_if_training(self):
        """The hand-written backward produces INPUT gradients only (forces).  In a training loop ``loss.backward()``
        would succeed and leave every ``param.grad`` None -- optimizers skip those silently -- so say it once."""
        if self.training and torch.is_grad_enabled() and not self._warned_inference_only and \
                any(p.requires_grad for p in self.parameters()):
            import warnings
            warnings.warn("gotennet_amd.GotenNet is an inference / force-evaluation path: its backward returns gradients "
                          "w.r.t. positions (edge_vec, edge_diff) only, parameters receive NO gradient and the backward "
                          "is not double-differentiable.  Call .eval() (or requires_grad_(False)) to silence this.",
                          RuntimeWarning, stacklevel=3)
            self._warned_inference_only = True

    def forward(self, atomic_numbers: Tensor, edge_index: Tensor, edge_diff: Tensor, edge_vec: Tensor,
                _trace: Optional[list] = None, _sorted: Optional[bool] = None) -> Tuple[Tensor, Tensor]:
        """``_sorted`` (internal): this CALL's edge list is target-major (the wrapper's own radius graph); None = the
        module's ``assume_sorted_edges``.  A per-call argument, not module state: calls from several threads do not race."""
        self._check_inputs(atomic_numbers, edge_index, edge_diff, edge_vec)
        cfg, pw = self.config(), self.packed_weights()
        N = atomic_numbers.shape[0]
        edge_index = edge_index.contiguous()
        edge_diff = edge_diff.to(torch.float32).contiguous()
        edge_vec = edge_vec.to(torch.float32).contiguous()
        if not (self.assume_sorted_edges if _sorted is None else _sorted):   # one host sync; skipped for sorted lists
            edge_index, edge_diff, edge_vec, _ = engine.sorted_edges(edge_index, edge_diff, edge_vec, N)
        z32 = atomic_numbers.to(torch.int32)
        self._warn_if_training()
        if torch.is_grad_enabled() and (edge_vec.requires_grad or edge_diff.requires_grad):
            return _RepresentationFn.apply(edge_diff.contiguous(), edge_vec.contiguous(), self, z32, edge_index)
        with torch.no_grad():
            g = engine.Graph(cfg, pw, N, edge_index, edge_diff, edge_vec)
            h, X, _ = engine.forward(cfg, pw, z32, g, trace=_trace)
            return h, X


class GotenNetWrapper(GotenNet):
    """Reference gotennet.py:1013-1045: builds the radius graph from ``inputs.z/.pos/.batch``."""

    def __init__(self, *args, max_num_neighbors=32, **kwargs):
        super().__init__(*args, **kwargs)
        self.max_num_neighbors = max_num_neighbors

    def forward(self, inputs) -> Tuple[Tensor, Tensor]:
        from .graph import distance
        atomic_numbers, pos, batch = inputs.z, inputs.pos, inputs.batch
        if torch.is_grad_enabled() and pos.requires_grad:
            self._warn_if_training()
            self._check_inputs(atomic_numbers, torch.zeros((2, 0), dtype=torch.int64), pos.new_zeros(0), pos.new_zeros((0, 3)))
            return _RepresentationPosFn.apply(pos, self, atomic_numbers.to(torch.int32), batch)
        edge_index, edge_diff, edge_vec = distance(pos, batch, self.cutoff, self.max_num_neighbors)
        return super().forward(atomic_numbers, edge_index, edge_diff, edge_vec, _sorted=True)   # radius graph is target-major
