"""Stand-in modules that let the *reference* GotenNet import in this container.

TEST INFRASTRUCTURE, survey-container only.  The reference
(/root/reference, read-only, never copied) needs torch_geometric,
torch_cluster, omegaconf and pytorch_lightning, none of which exist here (no
network).  This file registers minimal stand-ins in ``sys.modules`` that
implement the *documented PyG 2.x semantics* the reference's call sites rely on
(SURVEY.md section 8c):

* ``MessagePassing.propagate``: kwargs ``foo`` are gathered into ``foo_j``
  (``edge_index[0]``, the source) and ``foo_i`` (``edge_index[1]``, the
  target) along ``node_dim``; ``index = edge_index[1]``; ``dim_size = N``;
  then ``aggregate`` and ``update``.  ``edge_updater`` does the same for
  ``edge_update``.
* ``utils.softmax(src, index, ptr, num_nodes)``:
  ``exp(src - segmax) / (segsum + 1e-16)``.
* ``utils.scatter(src, index, dim, dim_size, reduce)``.
* ``torch_cluster.radius_graph(pos, r, batch, loop, max_num_neighbors)``:
  target-major, sources ascending, strict ``<r``, first ``max_num_neighbors``
  sources per target.

Nothing here ships to the GPU box and nothing in the product imports it.  It is
used by ``tools/make_golden.py`` (fixture generation) and by the CPU-only
tests that compare ``oracle/`` with the live reference when /root/reference is
present.
"""
from __future__ import annotations

import inspect
import sys
import types
from typing import Optional

import torch

REFERENCE_ROOT = "/root/reference"


def _scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    if dim < 0:
        dim += src.dim()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    if reduce in ("sum", "add"):
        return src.new_zeros(shape).scatter_add_(dim, idx, src)
    if reduce == "mean":
        out = src.new_zeros(shape).scatter_add_(dim, idx, src)
        cnt = src.new_zeros(dim_size).scatter_add_(0, index, src.new_ones(index.numel()))
        return out / cnt.clamp(min=1).view(view)
    if reduce == "max":
        out = src.new_full(shape, float("-inf"))
        out = out.scatter_reduce(dim, idx, src, reduce="amax", include_self=True)
        return torch.where(torch.isinf(out), torch.zeros_like(out), out)
    raise ValueError(reduce)


def _softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    assert index is not None
    n = int(num_nodes) if num_nodes is not None else int(index.max()) + 1
    shape = list(src.shape)
    shape[dim] = n
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    smax = src.new_full(shape, float("-inf")).scatter_reduce(
        dim, idx, src.detach(), reduce="amax", include_self=True)
    out = (src - smax.gather(dim, idx)).exp()
    ssum = src.new_zeros(shape).scatter_add_(dim, idx, out) + 1e-16
    return out / ssum.gather(dim, idx)


class _MessagePassing(torch.nn.Module):
    _special = {"edge_index", "index", "ptr", "dim_size", "size", "size_i", "size_j"}

    def __init__(self, aggr="add", *, flow="source_to_target", node_dim=-2, **_):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim

    def _collect(self, fn, edge_index, size, kwargs):
        params = list(inspect.signature(fn).parameters)
        j, i = (0, 1) if self.flow == "source_to_target" else (1, 0)
        n = size
        out = {}
        for name in params:
            if name in ("self",):
                continue
            if name == "edge_index":
                out[name] = edge_index
                continue
            if name in ("index",):
                out[name] = edge_index[i]
                continue
            if name == "ptr":
                out[name] = None
                continue
            if name in ("dim_size", "size_i"):
                out[name] = None  # filled below
                continue
            if name.endswith("_i") or name.endswith("_j"):
                base = name[:-2]
                data = kwargs[base]
                if n is None:
                    n = data.size(self.node_dim)
                sel = edge_index[i] if name.endswith("_i") else edge_index[j]
                out[name] = data.index_select(self.node_dim, sel)
            elif name in kwargs:
                out[name] = kwargs[name]
        if n is None:
            n = int(edge_index.max()) + 1 if edge_index.numel() else 0
        for k in ("dim_size", "size_i"):
            if k in out:
                out[k] = n
        return out, n

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = (0, 1) if self.flow == "source_to_target" else (1, 0)
        n = size[i] if isinstance(size, (tuple, list)) else size
        msg_kwargs, n = self._collect(self.message, edge_index, n, kwargs)
        msg = self.message(**msg_kwargs)
        ap = inspect.signature(self.aggregate).parameters
        akw = {}
        if "index" in ap:
            akw["index"] = edge_index[i]
        if "ptr" in ap:
            akw["ptr"] = None
        if "dim_size" in ap:
            akw["dim_size"] = n
        out = self.aggregate(msg, **akw)
        return self.update(out)

    def edge_updater(self, edge_index, size=None, **kwargs):
        kw, _ = self._collect(self.edge_update, edge_index, size, kwargs)
        return self.edge_update(**kw)

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return _scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=self.aggr)

    def update(self, inputs):
        return inputs

    def message(self, x_j):
        return x_j


def _radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32,
                  flow="source_to_target", num_workers=1, batch_size=None):
    n = x.size(0)
    if batch is None:
        batch = x.new_zeros(n, dtype=torch.long)
    d = torch.cdist(x.double(), x.double()) if False else None  # (not used: keep fp32 arithmetic below)
    diff = x.unsqueeze(1) - x.unsqueeze(0)          # [target? , source?] symmetric in norm
    dist2 = (diff * diff).sum(-1)
    ok = (dist2 < r * r) & (batch.unsqueeze(1) == batch.unsqueeze(0))
    if not loop:
        ok &= ~torch.eye(n, dtype=torch.bool, device=x.device)
    # row = target (centre) i, col = source j, sources ascending, first-k cap
    rank = ok.long().cumsum(dim=1)
    ok &= rank <= max_num_neighbors
    tgt, src = ok.nonzero(as_tuple=True)
    return torch.stack([src, tgt], dim=0)


def _glorot_orthogonal(tensor, scale=2.0):
    torch.nn.init.orthogonal_(tensor.data)
    s = scale / ((tensor.size(-2) + tensor.size(-1)) * tensor.var())
    tensor.data *= s.sqrt()
    return tensor


def install() -> None:
    """Register the stand-ins and put the reference on sys.path (idempotent)."""
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    if "torch_geometric" in sys.modules and getattr(sys.modules["torch_geometric"], "_gn_shim", False):
        return

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    tg = mod("torch_geometric"); tg._gn_shim = True
    tgnn = mod("torch_geometric.nn"); tgnn.MessagePassing = _MessagePassing
    tgin = mod("torch_geometric.nn.inits"); tgin.glorot_orthogonal = _glorot_orthogonal
    tgty = mod("torch_geometric.typing"); tgty.OptTensor = Optional[torch.Tensor]
    tgut = mod("torch_geometric.utils"); tgut.scatter = _scatter; tgut.softmax = _softmax
    tg.nn, tg.typing, tg.utils = tgnn, tgty, tgut
    tgnn.inits = tgin

    tc = mod("torch_cluster"); tc.radius_graph = _radius_graph

    oc = mod("omegaconf")
    oc.DictConfig = dict
    oc.OmegaConf = type("OmegaConf", (), {})

    pl = mod("pytorch_lightning")
    plu = mod("pytorch_lightning.utilities")

    def rank_zero_only(fn):
        return fn
    plu.rank_zero_only = rank_zero_only
    plu.rank_zero_warn = lambda *a, **k: None
    pl.utilities = plu
    pl.LightningModule = torch.nn.Module
    pl.Trainer = object
    pl.Callback = object
    pll = mod("pytorch_lightning.loggers"); pll.Logger = object
    pl.loggers = pll

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    """Return the reference's ``gotennet.models.representation.gotennet`` module."""
    install()
    import importlib
    return importlib.import_module("gotennet.models.representation.gotennet")
