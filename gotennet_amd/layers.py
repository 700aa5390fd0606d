"""Host-side mirror of the reference layer primitives that sit on the hot path.

These classes are *parameter containers with the reference's names and
state_dict layout* (reference gotennet/models/components/layers.py: Dense
457-529, MLP 533-581, CosineCutoff 133-152, ExpNormalSmearing 703-746, NodeInit
1607-1675, EdgeInit 1677-1714).  They hold no compute: the arithmetic runs in
the HIP kernels behind gotennet_amd._lib, driven by gotennet_amd.engine.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


class CosineCutoff(nn.Module):
    """Carries ``.cutoff`` (the only thing GotenNet reads, reference gotennet.py:839)."""

    def __init__(self, cutoff: float):
        super().__init__()
        if isinstance(cutoff, torch.Tensor):
            cutoff = cutoff.item()
        self.cutoff = cutoff


#: activation kinds of libgotennet_hip (GN_ACT_* in include/gotennet_hip.h): the element-wise activations the reference's
#: ``str2act`` (layers.py:596-700) can name, or a caller can pass as callable / nn.Module
ACT_KINDS = {"silu": 0, "swish": 0, "softplus": 1, "shiftedsoftplus": 1, "ssp": 1, "relu": 2, "tanh": 3, "sigmoid": 4,
             "elu": 5, "selu": 6, "mish": 7, "gelu": 8, "leakyrelu": 10}
_ACT_TORCH = {0: F.silu, 2: F.relu, 3: torch.tanh, 4: torch.sigmoid, 5: F.elu, 6: F.selu, 7: F.mish, 8: F.gelu,
              10: F.leaky_relu}
_ACT_MODULES = {nn.SiLU: 0, nn.ReLU: 2, nn.Tanh: 3, nn.Sigmoid: 4, nn.ELU: 5, nn.SELU: 6, nn.Mish: 7, nn.GELU: 8,
                nn.LeakyReLU: 10}


def shifted_softplus(x: torch.Tensor) -> torch.Tensor:
    """Reference layers.py:40-50 (the default activation of the reference's Atomwise head)."""
    return F.softplus(x) - 0.6931471805599453


_ACT_TORCH[1] = shifted_softplus


def activation_kind(act) -> int:
    """GN_ACT_* code of an activation given the way the reference accepts it: a name (``str2act``: case-insensitive,
    '-', '_' and spaces ignored; 'softplus' is the reference's shifted softplus), a torch functional, an nn.Module
    instance or class.  Parameterised variants (ELU alpha != 1, LeakyReLU slope != 0.01, GELU tanh) are rejected."""
    if act is None:
        raise NotImplementedError("activation=None: the reference needs an activation here")
    if isinstance(act, str):
        key = act.lower().replace("-", "").replace("_", "").replace(" ", "")
        if key in ACT_KINDS:
            return ACT_KINDS[key]
        raise NotImplementedError(f"activation {act!r} is not on the MI355X path (have: {sorted(ACT_KINDS)})")
    if act is shifted_softplus or getattr(act, "__name__", "") == "shifted_softplus" or \
            type(act).__name__ == "ShiftedSoftplus":
        return 1
    for k, fn in _ACT_TORCH.items():
        if act is fn:
            return k
    for cls, k in _ACT_MODULES.items():
        if act is cls:
            return k
        if isinstance(act, cls):
            if (cls is nn.ELU and act.alpha != 1.0) or (cls is nn.LeakyReLU and act.negative_slope != 0.01) or \
                    (cls is nn.GELU and act.approximate != "none"):
                break
            return k
    raise NotImplementedError(f"activation {act!r} is not on the MI355X path (have: {sorted(ACT_KINDS)})")


def resolve_activation(act):
    """-> the torch functional of a supported activation (kept on the mirror modules as ``.activation``; the HIP
    kernels take its GN_ACT_* code, ``activation_kind``)."""
    return _ACT_TORCH[activation_kind(act)]


def glorot_orthogonal_(tensor: torch.Tensor, scale: float = 2.0) -> torch.Tensor:
    """'glo_orthogonal' (reference layers.py:360-371 over torch_geometric.nn.inits.glorot_orthogonal): an orthogonal
    matrix rescaled to the Glorot variance scale / (fan_in + fan_out)."""
    nn.init.orthogonal_(tensor)
    with torch.no_grad():
        fan = tensor.size(-2) + tensor.size(-1)
        tensor.mul_((scale / (fan * tensor.var())).sqrt())
    return tensor


def he_orthogonal_(tensor: torch.Tensor) -> torch.Tensor:
    """'he_orthogonal' (reference layers.py:374-423): orthogonal matrix, rows standardised to zero mean / unit
    variance over the input axis, then scaled by fan_in ** -0.5."""
    nn.init.orthogonal_(tensor)
    with torch.no_grad():
        three = tensor.dim() == 3
        axis = [0, 1] if three else 1
        fan_in = tensor.shape[:-1].numel() if three else tensor.shape[1]
        var, mean = torch.var_mean(tensor, dim=axis, unbiased=True, keepdim=True)
        tensor.copy_((tensor - mean) / (var + 1e-6) ** 0.5 * (1.0 / fan_in) ** 0.5)
    return tensor


def get_weight_init_by_string(name: str) -> Callable:
    """Reference layers.py:426-452."""
    if name == "":
        return lambda x: x
    if name == "zeros":
        return nn.init.zeros_
    if name == "xavier_uniform":
        return nn.init.xavier_uniform_
    if name == "glo_orthogonal":
        return glorot_orthogonal_
    if name == "he_orthogonal":
        return he_orthogonal_
    raise ValueError(f"Unknown initialization {name}")


class Dense(nn.Linear):
    """nn.Linear with the reference's extra attributes (weight [out, in])."""

    def __init__(self, in_features, out_features, bias=True, activation=None,
                 weight_init=nn.init.xavier_uniform_, bias_init=nn.init.zeros_, norm=None):
        self.weight_init = weight_init
        self.bias_init = bias_init
        super().__init__(in_features, out_features, bias)
        self.activation = activation
        self.norm = nn.LayerNorm(out_features) if norm == "layer" else None
        if norm not in (None, "", "layer"):
            raise NotImplementedError(f"Dense norm={norm!r} is not on the accelerated path")

    def reset_parameters(self):
        self.weight_init(self.weight)
        if self.bias is not None:
            self.bias_init(self.bias)


class MLP(nn.Module):
    """Registers its layers twice (``dense_layers`` and ``layers``) exactly like the
    reference (layers.py:566-571) so state_dict keys match with strict=True."""

    def __init__(self, hidden_dims: List[int], bias=True, activation=None, last_activation=None,
                 weight_init=nn.init.xavier_uniform_, bias_init=nn.init.zeros_, norm=""):
        super().__init__()
        n = len(hidden_dims)
        mk = lambda i, o, act, nm: Dense(i, o, bias=bias, activation=act, weight_init=weight_init,
                                         bias_init=bias_init, norm=nm)
        self.dense_layers = nn.ModuleList(
            [mk(hidden_dims[i], hidden_dims[i + 1], activation, norm) for i in range(n - 2)]
            + [mk(hidden_dims[-2], hidden_dims[-1], last_activation, None)])
        self.layers = nn.Sequential(*self.dense_layers)

    def reset_parameters(self):
        for m in self.dense_layers:
            m.reset_parameters()


class ExpNormalSmearing(nn.Module):
    """Buffers ``means``/``betas`` as in reference layers.py:714-737."""

    def __init__(self, cutoff=5.0, n_rbf=50, trainable=False):
        super().__init__()
        if trainable:
            raise NotImplementedError("trainable radial basis")
        self.cutoff, self.n_rbf = float(cutoff), n_rbf
        self.alpha = 5.0 / self.cutoff
        means, betas = self._initial_params()
        self.register_buffer("means", means)
        self.register_buffer("betas", betas)

    def _initial_params(self):
        start = torch.exp(torch.scalar_tensor(-self.cutoff))
        means = torch.linspace(start, 1, self.n_rbf)
        betas = torch.tensor([(2 / self.n_rbf * (1 - start)) ** -2] * self.n_rbf)
        return means, betas

    def reset_parameters(self):
        means, betas = self._initial_params()
        with torch.no_grad():                       # in-place through autograd's version counter (cache keys see it)
            self.means.copy_(means)
            self.betas.copy_(betas)


class GaussianRBF(nn.Module):
    """Buffers ``widths``/``offsets`` as in reference layers.py:294-326 (evaluated by gn_edge_geometry, basis 2)."""

    def __init__(self, n_rbf: int, cutoff: float, start: float = 0.0, trainable: bool = False):
        super().__init__()
        if trainable:
            raise NotImplementedError("trainable radial basis")
        self.n_rbf = n_rbf
        offset = torch.linspace(start, cutoff, n_rbf)
        widths = torch.abs(offset[1] - offset[0]) * torch.ones_like(offset)
        self.register_buffer("widths", widths)
        self.register_buffer("offsets", offset)


class BesselBasis(nn.Module):
    """Buffers ``freqs``/``norm1`` as in reference layers.py:329-347 (evaluated by gn_edge_geometry, basis 1)."""

    def __init__(self, cutoff=5.0, n_rbf=None, trainable=False):
        super().__init__()
        if n_rbf is None:
            raise ValueError("n_rbf must be specified for BesselBasis")
        self.n_rbf = n_rbf
        freqs = torch.arange(1, n_rbf + 1) * math.pi / cutoff
        self.register_buffer("freqs", freqs)
        self.register_buffer("norm1", torch.tensor(1.0))


#: gn_edge_geometry `basis` code and the two parameter vectors it reads, per radial-basis class
BASIS_CODE = {ExpNormalSmearing: (0, "means", "betas"), BesselBasis: (1, "freqs", "freqs"),
              GaussianRBF: (2, "offsets", "widths")}


def str2basis(basis: Union[str, Callable]):
    """Reference layers.py:749-776 (same spellings accepted)."""
    if not isinstance(basis, str):
        if basis not in BASIS_CODE:
            raise NotImplementedError(f"radial basis {basis!r} has no HIP kernel (expnorm, BesselBasis, GaussianRBF do)")
        return basis
    if basis.lower().replace("-", "").replace("_", "").replace(" ", "") == "besselbasis":
        return BesselBasis
    if basis == "GaussianRBF":
        return GaussianRBF
    if basis.lower() == "expnorm":
        return ExpNormalSmearing
    raise ValueError("Unknown radial basis: {}".format(basis))


class NodeInit(nn.Module):
    def __init__(self, hidden_channels, num_rbf, cutoff, max_z=100, activation=F.silu, proj_ln="",
                 weight_init=nn.init.xavier_uniform_, bias_init=nn.init.zeros_):
        super().__init__()
        if isinstance(hidden_channels, int):
            hidden_channels = [hidden_channels]
        last = hidden_channels[-1]
        self.A_nbr = nn.Embedding(max_z, last)
        self.W_ndp = MLP([num_rbf, last], activation=None, norm="", weight_init=weight_init, bias_init=bias_init)
        self.W_nrd_nru = MLP([2 * last] + hidden_channels, activation=activation, norm=proj_ln,
                             weight_init=weight_init, bias_init=bias_init)

    def reset_parameters(self):
        self.A_nbr.reset_parameters()
        self.W_ndp.reset_parameters()
        self.W_nrd_nru.reset_parameters()


class EdgeInit(nn.Module):
    def __init__(self, num_rbf, hidden_channels):
        super().__init__()
        self.W_erp = nn.Linear(num_rbf, hidden_channels)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.W_erp.weight)
        with torch.no_grad():
            self.W_erp.bias.fill_(0)
