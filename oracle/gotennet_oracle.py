"""CPU oracle for the GotenNet interaction hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the *checker*.  The product path
(``gotennet_amd``) never imports it and fails loudly when the HIP library is
missing.

This is a plain-torch (CPU, fp32 or fp64) restatement of the reference's
algorithm with no PyG / torch_cluster dependency.  Every function cites the
reference file:line (relative to the reference checkout) it follows.  Parity
pin: ``tools/make_golden.py`` runs the *real* reference (imported through
``tools/ref_shims.py`` in the survey container) and commits inputs, weights,
per-layer and final outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` holds this oracle to those vectors.  The
reference itself ships no tests or golden vectors, and
its PyG/torch_cluster boundaries are un-vendored and unpinned, so the shim
semantics documented in ``tools/ref_shims.py`` (= documented PyG 2.x behaviour)
are the definition used here.

Parameter naming follows the reference ``state_dict`` exactly, so a reference
checkpoint can be fed to the oracle unchanged.
"""
from __future__ import annotations

import functools
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- activations
def shifted_softplus(x: Tensor) -> Tensor:
    """layers.py:40-50."""
    return F.softplus(x) - math.log(2.0)


#: the names the reference's str2act resolves (layers.py:596-700), normalised (lower case, no '-', '_', ' ')
ACTIVATIONS = {"silu": F.silu, "swish": F.silu, "softplus": shifted_softplus, "relu": F.relu, "tanh": torch.tanh,
               "sigmoid": torch.sigmoid, "elu": F.elu, "selu": F.selu, "mish": F.mish, "gelu": F.gelu,
               "leakyrelu": F.leaky_relu}


def activation_of(cfg_or_name):
    """The activation function of a config (key ``activation``, default SiLU = the reference default) or of a name."""
    name = cfg_or_name.get("activation", "silu") if isinstance(cfg_or_name, dict) else cfg_or_name
    if callable(name):
        return name
    return ACTIVATIONS[str(name).lower().replace("-", "").replace("_", "").replace(" ", "")]


# --------------------------------------------------------------------------- config
def default_config(**kw) -> dict:
    """Hyper-parameters with the reference *class* defaults (gotennet.py:767-793)."""
    cfg = dict(
        n_atom_basis=128, n_interactions=8, n_rbf=32, cutoff=5.0, max_z=100,
        epsilon=1e-8, num_heads=8, scale_edge=True, lmax=1,
        sep_htr=True, sep_dir=False, sep_tensor=False,
        radial_basis="expnorm", edge_updates=True, layernorm="", steerable_norm="", aggr="add",
    )
    cfg.update(kw)
    return cfg


def edge_update_info(edge_updates) -> dict:
    """gotennet.py:139-190: the '_'-separated string form of ``edge_updates``."""
    info = dict(enabled=bool(edge_updates), rej=True, gate="", mlp=False, mlpa=False, lin_w=0, lin_ln=0)
    parts = edge_updates.split("_") if isinstance(edge_updates, str) else []
    allowed = ["gated", "gatedt", "norej", "norm", "mlp", "mlpa", "act", "linw", "linwa", "ln", "postln"]
    if not all(part in allowed for part in parts):
        raise ValueError(f"Invalid edge update parts. Allowed parts are {allowed}")
    if "gated" in parts:
        info["gate"] = "sigmoid"
    if "gatedt" in parts:
        info["gate"] = "tanh"
    if "act" in parts:
        info["gate"] = "silu"
    if "norej" in parts:
        info["rej"] = False
    info["mlp"], info["mlpa"] = "mlp" in parts, "mlpa" in parts
    if "linw" in parts:
        info["lin_w"] = 1
    if "linwa" in parts:
        info["lin_w"] = 2
    if "ln" in parts:
        info["lin_ln"] = 1
    if "postln" in parts:
        info["lin_ln"] = 2
    return info


def multiplier(cfg: dict) -> int:
    """gotennet.py:197-203."""
    m = 3
    if cfg["sep_dir"]:
        m += cfg["lmax"] - 1
    if cfg["sep_tensor"]:
        m += cfg["lmax"] - 1
    return m


def degree_sizes(lmax: int) -> List[int]:
    """gotennet.py:37-51."""
    return [2 * l + 1 for l in range(1, lmax + 1)]


# --------------------------------------------------------------------------- edge basis
def cosine_cutoff(d: Tensor, cutoff: float) -> Tensor:
    """layers.py:149-152."""
    c = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0)
    return c * (d < cutoff).to(d.dtype)


def expnorm_params(cutoff: float, n_rbf: int) -> Tuple[Tensor, Tensor]:
    """layers.py:733-737 (fp32 buffers, as the reference registers them)."""
    start = torch.exp(torch.scalar_tensor(-cutoff))
    means = torch.linspace(start, 1, n_rbf)
    betas = torch.tensor([(2 / n_rbf * (1 - start)) ** -2] * n_rbf)
    return means, betas


def expnorm_smearing(d: Tensor, means: Tensor, betas: Tensor, cutoff: float) -> Tensor:
    """layers.py:744-746.  d is 1-D [E]; returns [E, R]."""
    alpha = 5.0 / cutoff
    d = d.unsqueeze(-1)
    return cosine_cutoff(d, cutoff) * torch.exp(-betas * (torch.exp(alpha * (-d)) - means) ** 2)


def bessel_basis(d: Tensor, freqs: Tensor) -> Tensor:
    """layers.py:349-358 (BesselBasis.forward): sin(a d) / d, d = 0 divides by 1."""
    d = d.unsqueeze(-1)
    return torch.sin(d * freqs) / torch.where(d == 0, torch.ones_like(d), d)


def gaussian_rbf(d: Tensor, offsets: Tensor, widths: Tensor) -> Tensor:
    """layers.py:276-291 as called by GaussianRBF.forward (325-326) on a 1-D distance vector."""
    coeff = -0.5 / torch.pow(widths, 2)
    return torch.exp(coeff * torch.pow(d.unsqueeze(-1) - offsets, 2))


def radial_basis(sd: Dict[str, Tensor], cfg: dict, d: Tensor) -> Tensor:
    """str2basis (layers.py:749-776) + the selected module's forward; buffers come from the state_dict."""
    name = cfg.get("radial_basis", "expnorm")
    norm = name.lower().replace("-", "").replace("_", "").replace(" ", "")
    kind = "bessel" if norm == "besselbasis" else "gaussian" if name == "GaussianRBF" else norm
    if kind == "expnorm":
        return expnorm_smearing(d, sd["radial_basis.means"], sd["radial_basis.betas"], cfg["cutoff"])
    if kind == "bessel":
        return bessel_basis(d, sd["radial_basis.freqs"])
    if kind == "gaussian":
        return gaussian_rbf(d, sd["radial_basis.offsets"], sd["radial_basis.widths"])
    raise ValueError(f"Unknown radial basis: {kind}")


def _proper_harmonics(l: int, x, y, z):
    """Component-normalised real harmonics of degree l (numpy, homogeneous closed form): polar axis y, azimuth from z
    towards x, index m + l -- the basis the reference's degree-raising recursion couples in."""
    X, Y, Z = z, x, y
    r2 = X * X + Y * Y + Z * Z
    A, B = [np.ones_like(X)], [np.zeros_like(X)]
    for _ in range(l):
        A, B = A + [X * A[-1] - Y * B[-1]], B + [X * B[-1] + Y * A[-1]]
    out = [None] * (2 * l + 1)
    for m in range(l + 1):
        q0 = float(np.prod(np.arange(1, 2 * m, 2))) * np.ones_like(X)
        Q = q0
        if l > m:
            q1 = (2 * m + 1) * Z * q0
            for ll in range(m + 2, l + 1):
                q0, q1 = q1, ((2 * ll - 1) * Z * q1 - (ll + m - 1) * r2 * q0) / (ll - m)
            Q = q1
        if m == 0:
            out[l] = math.sqrt(2 * l + 1) * Q
        else:
            n = math.sqrt(2.0 * math.factorial(l - m) / math.factorial(l + m) * (2 * l + 1))
            out[l + m], out[l - m] = n * Q * A[m], n * Q * B[m]
    return np.stack(out, -1)


@functools.lru_cache(maxsize=None)
def harmonic_raise_table(l: int):
    """T_l [2l+1, 2l-1, 3] of the reference's recursion  sh_l[i] = sum_{j,a} T_l[i,j,a] sh_{l-1}[j] r_a
    (layers.py:934-1494, degrees 5..8; degree 4, layers.py:869-902, is the same construction and is the check in
    tests/test_oracle_golden.py).  T_l is the (l-1) x 1 -> l coupling of the real harmonics: computed here as the Gaunt
    integral  Int Y_l,i Y_{l-1},j r_a dOmega  (Gauss-Legendre x uniform quadrature, exact for these polynomials), scaled
    to the reference's leading coefficient T_l[0, 0, z] = sqrt((2l+1) / (2l)).  Pinned by the reference KAT
    tests/golden/kat_sh_l8.npz."""
    ct, wt = np.polynomial.legendre.leggauss(24)
    phi = (np.arange(48) + 0.5) * 2 * np.pi / 48
    CT, PH = np.meshgrid(ct, phi, indexing="ij")
    W = (np.repeat(wt[:, None], 48, 1) * (2 * np.pi / 48) / (4 * np.pi)).ravel()
    st = np.sqrt(1 - CT ** 2)
    x, y, z = (st * np.cos(PH)).ravel(), (st * np.sin(PH)).ravel(), CT.ravel()
    T = np.einsum("p,pi,pj,pa->ija", W, _proper_harmonics(l, x, y, z), _proper_harmonics(l - 1, x, y, z),
                  np.stack([x, y, z], -1))
    T[np.abs(T) < 1e-12] = 0.0
    return T * (math.sqrt((2 * l + 1) / (2 * l)) / T[0, 0, 2])


def real_harmonics(lmax: int, u: Tensor) -> Tensor:
    """layers.py:805-1494: literal polynomial formulas for degrees <= 4 on the unit vector ``u[..., 3]``, l = 0
    omitted, order (x, y, z) for l = 1; degrees 5..8 by the reference's recursion on the degree below with the
    coefficient tables of ``harmonic_raise_table``."""
    if not 1 <= lmax <= 8:
        raise NotImplementedError("TensorInit is defined for 1 <= lmax <= 8")
    full_lmax, lmax = lmax, min(lmax, 4)
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    out = [x, y, z]
    if lmax >= 2:
        r3 = math.sqrt(3.0)
        y2 = y.pow(2)
        x2z2 = x.pow(2) + z.pow(2)
        s2 = [r3 * x * z, r3 * x * y, y2 - 0.5 * x2z2, r3 * y * z, r3 / 2.0 * (z.pow(2) - x.pow(2))]
        out += s2
    if lmax >= 3:
        a = (1 / 6) * math.sqrt(42)
        b = math.sqrt(7)
        c = (1 / 8) * math.sqrt(168)
        s3 = [
            a * (s2[0] * z + s2[4] * x),
            b * s2[0] * y,
            c * (4.0 * y2 - x2z2) * x,
            (1 / 2) * b * y * (2.0 * y2 - 3.0 * x2z2),
            c * z * (4.0 * y2 - x2z2),
            b * s2[4] * y,
            a * (s2[4] * z - s2[0] * x),
        ]
        out += s3
    if lmax >= 4:
        q2, q6, q14, q21 = math.sqrt(2), math.sqrt(6), math.sqrt(14), math.sqrt(21)
        q42, q70, q105, q210, q7 = math.sqrt(42), math.sqrt(70), math.sqrt(105), math.sqrt(210), math.sqrt(7)
        s4 = [
            (3 / 4) * q2 * (s3[0] * z + s3[6] * x),
            (3 / 4) * s3[0] * y + (3 / 8) * q6 * s3[1] * z + (3 / 8) * q6 * s3[5] * x,
            (-3 / 56 * q14 * s3[0] * z + (3 / 14) * q21 * s3[1] * y + (3 / 56) * q210 * s3[2] * z
             + (3 / 56) * q210 * s3[4] * x + (3 / 56) * q14 * s3[6] * x),
            (-3 / 56 * q42 * s3[1] * z + (3 / 28) * q105 * s3[2] * y + (3 / 28) * q70 * s3[3] * x
             + (3 / 56) * q42 * s3[5] * x),
            -3 / 28 * q42 * s3[2] * x + (3 / 7) * q7 * s3[3] * y - 3 / 28 * q42 * s3[4] * z,
            (-3 / 56 * q42 * s3[1] * x + (3 / 28) * q70 * s3[3] * z + (3 / 28) * q105 * s3[4] * y
             - 3 / 56 * q42 * s3[5] * z),
            (-3 / 56 * q14 * s3[0] * x - 3 / 56 * q210 * s3[2] * x + (3 / 56) * q210 * s3[4] * z
             + (3 / 14) * q21 * s3[5] * y - 3 / 56 * q14 * s3[6] * z),
            -3 / 8 * q6 * s3[1] * x + (3 / 8) * q6 * s3[5] * z + (3 / 4) * s3[6] * y,
            (3 / 4) * q2 * (-s3[0] * x + s3[6] * z),
        ]
        out += s4
    prev = out[15:24] if full_lmax > 4 else None
    for l in range(5, full_lmax + 1):
        T = torch.as_tensor(harmonic_raise_table(l), dtype=u.dtype)
        r = (x, y, z)
        cur = []
        for i in range(2 * l + 1):
            acc = None
            for j in range(2 * l - 1):
                for a in range(3):
                    if T[i, j, a] != 0:
                        term = T[i, j, a] * (prev[j] * r[a])
                        acc = term if acc is None else acc + term
            cur.append(acc)
        out += cur
        prev = cur
    return torch.stack(out, dim=-1)


# --------------------------------------------------------------------------- dense helpers
def linear(x: Tensor, sd: Dict[str, Tensor], key: str) -> Tensor:
    """layers.py:457-529 (Dense without norm/activation): y = x W^T + b, W is [out, in]."""
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def segment_softmax(s: Tensor, index: Tensor, n: int) -> Tensor:
    """PyG ``utils.softmax`` (call site gotennet.py:503):
    exp(s - segmax) / (segsum + 1e-16), segments = target node, along dim 0."""
    shape = list(s.shape)
    shape[0] = n
    idx = index.view(-1, *([1] * (s.dim() - 1))).expand_as(s)
    smax = s.new_full(shape, float("-inf")).scatter_reduce(0, idx, s.detach(), reduce="amax", include_self=True)
    e = (s - smax.gather(0, idx)).exp()
    den = s.new_zeros(shape).scatter_add_(0, idx, e) + 1e-16
    return e / den.gather(0, idx)


# --------------------------------------------------------------------------- init layers
def node_init(sd, cfg, z, h0, edge_index, edge_diff, phi) -> Tensor:
    """layers.py:1658-1675 (NodeInit.forward/message) with proj_ln='layer' (gotennet.py:841-850)."""
    mask = edge_index[0] != edge_index[1]
    ei = edge_index[:, mask]
    r = edge_diff[mask]
    ph = phi[mask]
    h_src = sd["node_init.A_nbr.weight"][z]
    feat = linear(ph, sd, "node_init.W_ndp.dense_layers.0") * cosine_cutoff(r, cfg["cutoff"]).view(-1, 1)
    msg = h_src.index_select(0, ei[0]) * feat
    m = torch.zeros_like(h0).index_add_(0, ei[1], msg)
    y = linear(torch.cat([h0, m], dim=1), sd, "node_init.W_nrd_nru.dense_layers.0")
    y = F.layer_norm(y, (y.shape[-1],), sd["node_init.W_nrd_nru.dense_layers.0.norm.weight"],
                     sd["node_init.W_nrd_nru.dense_layers.0.norm.bias"], 1e-5)
    y = activation_of(cfg)(y)
    return linear(y, sd, "node_init.W_nrd_nru.dense_layers.1")


def edge_init(sd, edge_index, phi, h) -> Tensor:
    """layers.py:1704-1714: t_ij = (h_i + h_j) * W_erp(phi) on every edge (self-loops included)."""
    h_i = h.index_select(0, edge_index[1])
    h_j = h.index_select(0, edge_index[0])
    return (h_i + h_j) * linear(phi, sd, "edge_init.W_erp")


# --------------------------------------------------------------------------- GATA
def _mlp2(x, sd, key, act=F.silu):
    """nn.Sequential(Dense(act), Dense(act=None)) -- gotennet.py:209-224."""
    return linear(act(linear(x, sd, key + ".0")), sd, key + ".1")


def tensor_layernorm(X: Tensor, lmax: int, weight: Tensor, eps: float = 1e-12) -> Tensor:
    """layers.py:1529-1563 (TensorLayerNorm, max-min norm per degree block).  The reference's early
    ``(dist == 0).all() -> zeros`` return equals the formula below (0 / eps * relu(0) = 0)."""
    outs = []
    for part in torch.split(X, degree_sizes(lmax), dim=1):
        dist = torch.norm(part, dim=1, keepdim=True).clamp(min=eps)
        direct = part / dist
        max_val, _ = torch.max(dist, dim=-1)
        min_val, _ = torch.min(dist, dim=-1)
        delta = (max_val - min_val).view(-1)
        delta = torch.where(delta == 0, torch.ones_like(delta), delta)
        dist = (dist - min_val.view(-1, 1, 1)) / delta.view(-1, 1, 1)
        outs.append(F.relu(dist) * direct)
    return torch.cat(outs, dim=1) * weight.unsqueeze(0).unsqueeze(0)


def gata_input_norms(sd, cfg, p, h, X):
    """gotennet.py:397-398: optional nn.LayerNorm on h and TensorLayerNorm on X; the residuals at 426-427
    then add to the NORMALISED values."""
    if cfg.get("layernorm", ""):
        h = F.layer_norm(h, (h.shape[-1],), sd[p + "layernorm.weight"], sd[p + "layernorm.bias"], 1e-5)
    if cfg.get("steerable_norm", ""):
        X = tensor_layernorm(X, cfg["lmax"], sd[p + "tensor_layernorm.weight"])
    return h, X


def gata_message_aggregate(sd, cfg, p, edge_index, h, X, rl, t, r, n_edges):
    """gotennet.py:400-427 + message 452-559 + aggregate 613-640.

    h [N,F], X [N,D,F], rl [E,D], t [E,F], r [E], n_edges [E].  Returns h', X'.
    """
    Fd, H, lmax = cfg["n_atom_basis"], cfg["num_heads"], cfg["lmax"]
    M = multiplier(cfg)
    N = h.shape[0]
    j, i = edge_index[0], edge_index[1]
    q = linear(h, sd, p + "W_q").reshape(-1, H, Fd // H)
    k = linear(h, sd, p + "W_k").reshape(-1, H, Fd // H)
    act = activation_of(cfg)
    x = _mlp2(h, sd, p + "gamma_s", act)                         # [N, M F]
    v = _mlp2(h, sd, p + "gamma_v", act)
    t_attn = act(linear(t, sd, p + "W_re")).reshape(-1, H, Fd // H)
    t_filter = linear(t, sd, p + "W_rs")                         # [E, M F]

    attn = (q.index_select(0, i) * k.index_select(0, j) * t_attn).sum(dim=-1, keepdim=True)  # [E,H,1]
    attn = segment_softmax(attn, i, N)
    if cfg["scale_edge"]:
        norm = torch.sqrt(n_edges.reshape(-1, 1, 1)) / math.sqrt(Fd)
    else:
        norm = 1.0 / math.sqrt(Fd)
    attn = attn * norm
    sea = (attn * v.index_select(0, j).reshape(-1, H, (Fd * M) // H)).reshape(-1, Fd * M)
    spatial = t_filter * x.index_select(0, j) * cosine_cutoff(r, cfg["cutoff"]).unsqueeze(-1)
    out = spatial + sea
    comps = list(torch.split(out, Fd, dim=-1))
    o_s, comps = comps[0], comps[1:]
    sizes = degree_sizes(lmax)
    rl_split = torch.split(rl, sizes, dim=1)
    X_j = X.index_select(0, j)
    X_split = torch.split(X_j, sizes, dim=1)
    if cfg["sep_dir"]:
        o_d, comps = comps[:lmax], comps[lmax:]
        dX_R = torch.cat([rl_split[a].unsqueeze(-1) * o_d[a].unsqueeze(1) for a in range(lmax)], dim=1)
    else:
        o_d, comps = comps[0], comps[1:]
        dX_R = o_d.unsqueeze(1) * rl.unsqueeze(-1)
    if cfg["sep_tensor"]:
        o_t = comps[:lmax]
        dX_X = torch.cat([X_split[a] * o_t[a].unsqueeze(1) for a in range(lmax)], dim=1)
    else:
        dX_X = comps[0].unsqueeze(1) * X_j
    dX = dX_R + dX_X
    d_h, d_X = _aggregate(o_s, i, h, cfg.get("aggr", "add")), _aggregate(dX, i, X, cfg.get("aggr", "add"))
    return h + d_h, X + d_X


def _aggregate(msg: Tensor, index: Tensor, like: Tensor, aggr: str) -> Tensor:
    """GATA.aggregate (gotennet.py:638-639): PyG ``scatter(msg, index, dim=0, dim_size=N, reduce=aggr)`` -- "add" (every
    config), "mean" = sum / in-degree (count clamped to 1), "max" = element-wise maximum over the incoming edges; atoms
    without incoming edges get 0 in all three (PyG 2.x semantics, the definition of tools/ref_shims._scatter)."""
    out = torch.zeros_like(like).index_add_(0, index, msg)
    if aggr in ("add", "sum"):
        return out
    if aggr == "mean":
        cnt = torch.zeros(like.shape[0], dtype=like.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=like.dtype))
        return out / cnt.clamp(min=1).reshape([-1] + [1] * (like.dim() - 1))
    if aggr == "max":
        idx = index.reshape([-1] + [1] * (msg.dim() - 1)).expand_as(msg)
        mx = torch.full_like(like, float("-inf")).scatter_reduce(0, idx, msg, reduce="amax", include_self=True)
        return torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    raise ValueError(f"aggr={aggr!r}")


def _rejection(rep: Tensor, rl: Tensor) -> Tensor:
    """gotennet.py:351-364.  rep [E,m,F], rl [E,m]."""
    proj = (rep * rl.unsqueeze(2)).sum(dim=1, keepdim=True)
    return rep - proj * rl.unsqueeze(2)


def gata_htr(sd, cfg, p, edge_index, X, rl, t):
    """gotennet.py:429-445 + edge_update 561-611: rejection on/off, per-degree or joint (sep_htr);
    t' = t + gamma_t(t) * gamma_w(w)."""
    lmax = cfg["lmax"]
    info = edge_update_info(cfg.get("edge_updates", True))
    sizes = degree_sizes(lmax)
    j, i = edge_index[0], edge_index[1]
    EQ = F.linear(X, sd[p + "W_vq.weight"])
    if cfg["sep_htr"]:
        X_split = torch.split(X, sizes, dim=1)
        EK = torch.cat([F.linear(X_split[a], sd[p + f"W_vk.{a}.weight"]) for a in range(lmax)], dim=1)
        blocks = sizes
    else:
        EK = F.linear(X, sd[p + "W_vk.weight"])
        blocks = [sum(sizes)]
    EQ_i = torch.split(EQ.index_select(0, i), blocks, dim=1)
    EK_j = torch.split(EK.index_select(0, j), blocks, dim=1)
    rl_s = torch.split(rl, blocks, dim=1)
    w = None
    for a in range(len(blocks)):
        eq, ek = EQ_i[a], EK_j[a]
        if info["rej"]:
            eq = _rejection(eq, rl_s[a])
            ek = _rejection(ek, -rl_s[a])
        wl = (eq * ek).sum(dim=1)
        w = wl if w is None else w + wl
    return t + gamma_t(sd, cfg, p, info, t) * gamma_w(sd, p, info, w, activation_of(cfg))


def _layer_norm(x, sd, key):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def gamma_t(sd, cfg, p, info, t):
    """gotennet.py:236-251: MLP([F, F]) with SiLU, or with "mlp"/"mlpa" MLP([F, emlp, F]) whose hidden Dense
    carries the optional ``edge_ln`` LayerNorm (layers.py:518-529) and whose last activation is None for "mlp"."""
    k = p + "gamma_t.dense_layers."
    act = activation_of(cfg)
    if info["mlp"] or info["mlpa"]:
        u = linear(t, sd, k + "0")
        if (k + "0.norm.weight") in sd:
            u = _layer_norm(u, sd, k + "0.norm")
        y = linear(act(u), sd, k + "1")
        return y if info["mlp"] else act(y)
    return act(linear(t, sd, k + "0"))


def gamma_w(sd, p, info, w, act=F.silu):
    """gotennet.py:270-291: nn.Sequential([LayerNorm "ln"], [act "linwa"], W_edp [with norm "postln"], [gate])."""
    if info["lin_w"] > 0:
        if info["lin_ln"] == 1:
            w = _layer_norm(w, sd, p + "gamma_w.0")
        if info["lin_w"] == 2:
            w = act(w)                                            # "linwa": self.activation (gotennet.py:275)
        w = linear(w, sd, p + "W_edp")
        if info["lin_ln"] == 2:
            w = _layer_norm(w, sd, p + "W_edp.norm")
    if info["gate"] == "sigmoid":
        w = torch.sigmoid(w)
    elif info["gate"] == "tanh":
        w = torch.tanh(w)
    elif info["gate"] == "silu":
        w = F.silu(w)
    return w


def eqff(sd, cfg, p, h, X):
    """gotennet.py:716-748."""
    Fd = cfg["n_atom_basis"]
    X_p = F.linear(X, sd[p + "W_vu.weight"])
    X_pn = torch.sqrt(torch.sum(X_p ** 2, dim=-2) + cfg["epsilon"])
    ctx = torch.cat([h, X_pn], dim=-1)
    m = _mlp2(ctx, sd, p + "gamma_m", activation_of(cfg))
    m1, m2 = torch.split(m, Fd, dim=-1)
    return h + m1, X + m2.unsqueeze(1) * X_p


# --------------------------------------------------------------------------- stack driver
def gotennet_forward(sd: Dict[str, Tensor], cfg: dict, z: Tensor, edge_index: Tensor,
                     edge_diff: Tensor, edge_vec: Tensor, return_trace: bool = False):
    """gotennet.py:956-1010.  Inputs are NOT modified (the reference normalises
    ``edge_vec`` in place at 978-980; callers of the reference must clone)."""
    Fd, L, lmax = cfg["n_atom_basis"], cfg["n_interactions"], cfg["lmax"]
    dt = sd["A_na.weight"].dtype
    h = sd["A_na.weight"][z]
    phi = radial_basis(sd, cfg, edge_diff)
    h = node_init(sd, cfg, z, h, edge_index, edge_diff, phi)
    t = edge_init(sd, edge_index, phi, h)
    mask = (edge_index[0] != edge_index[1]).unsqueeze(1)
    nrm = torch.norm(edge_vec, dim=1, keepdim=True)
    unit = torch.where(mask, edge_vec / torch.where(mask, nrm, torch.ones_like(nrm)), edge_vec)
    rl = real_harmonics(lmax, unit)
    N = h.shape[0]
    deg = torch.zeros(N, dtype=edge_diff.dtype).index_add_(0, edge_index[0], torch.ones_like(edge_diff))
    n_edges = deg[edge_index[0]]
    D = (lmax + 1) ** 2 - 1
    X = torch.zeros((N, D, Fd), dtype=torch.promote_types(dt, torch.float32))
    trace = []
    for li in range(L):
        p = f"gata_list.{li}."
        h, X = gata_input_norms(sd, cfg, p, h, X)
        h, X = gata_message_aggregate(sd, cfg, p, edge_index, h, X, rl, t, edge_diff, n_edges)
        if li != L - 1 and cfg.get("edge_updates", True):
            t = gata_htr(sd, cfg, p, edge_index, X, rl, t)
        h, X = eqff(sd, cfg, f"eqff_list.{li}.", h, X)
        if return_trace:
            trace.append((h, X, t))
    if return_trace:
        return h, X, dict(phi=phi, rl=rl, layers=trace)
    return h, X


# --------------------------------------------------------------------------- graph builder (adjacent)
def radius_graph(pos: Tensor, batch: Tensor, cutoff: float, max_num_neighbors: int = 32,
                 loop: bool = True) -> Tensor:
    """torch_cluster.radius_graph as used at layers.py:1589-1590: edges j->i for
    ||pos_j - pos_i|| < cutoff within one molecule, target-major, sources
    ascending, at most ``max_num_neighbors`` sources per target (first-k)."""
    n = pos.shape[0]
    diff = pos.unsqueeze(1) - pos.unsqueeze(0)
    d2 = (diff * diff).sum(-1)
    ok = (d2 < cutoff * cutoff) & (batch.unsqueeze(1) == batch.unsqueeze(0))
    if not loop:
        ok &= ~torch.eye(n, dtype=torch.bool)
    ok &= ok.long().cumsum(dim=1) <= max_num_neighbors
    tgt, src = ok.nonzero(as_tuple=True)
    return torch.stack([src, tgt], dim=0)


def distance(pos: Tensor, batch: Tensor, cutoff: float, max_num_neighbors: int = 32):
    """layers.py:1588-1604 (Distance.forward, loop=True): edge_vec = pos[j]-pos[i];
    edge_weight = norm for non-self edges, 0 for self-loops."""
    ei = radius_graph(pos, batch, cutoff, max_num_neighbors, loop=True)
    vec = pos[ei[0]] - pos[ei[1]]
    mask = ei[0] != ei[1]
    # same value as the reference's masked assignment; written so autograd never
    # sees norm'(0) on the self-loops.
    safe = torch.where(mask.unsqueeze(1), vec, torch.ones_like(vec))
    w = torch.where(mask, torch.norm(safe, dim=-1), torch.zeros_like(vec[:, 0]))
    return ei, w, vec


# --------------------------------------------------------------------------- energy / forces
def atomwise_contributions(head: Dict[str, Tensor], h: Tensor, z: Optional[Tensor] = None,
                           activation: str = "silu") -> Tensor:
    """outputs.py:323-346 over SchnetMLP (layers.py:225-273): every layer but the last is activated;
    y_i = stddev * MLP(h_i) + mean [+ atomref[z_i]]."""
    act = activation_of(activation)
    n = 0
    while f"out_net.1.out_net.{n}.weight" in head:
        n += 1
    y = h
    for k in range(n):
        y = F.linear(y, head[f"out_net.1.out_net.{k}.weight"], head[f"out_net.1.out_net.{k}.bias"])
        if k + 1 < n:
            y = act(y)
    if "standardize.stddev" in head:
        y = y * head["standardize.stddev"] + head["standardize.mean"]
    if "atomref.weight" in head:
        y = y + head["atomref.weight"][z]
    return y


def atomwise_energy(head: Dict[str, Tensor], h: Tensor, batch: Tensor, n_mol: int,
                    activation: str = "silu", z: Optional[Tensor] = None, aggregation: str = "sum") -> Tensor:
    """outputs.py:348-358: scatter of the per-atom values by molecule, reduce "sum" or "mean"."""
    y = atomwise_contributions(head, h, z, activation)
    out = torch.zeros((n_mol, y.shape[1]), dtype=y.dtype).index_add_(0, batch, y)
    if aggregation == "mean":
        out = out / torch.bincount(batch, minlength=n_mol).clamp(min=1).to(y.dtype).unsqueeze(1)
    return out


def atomwise_v3(head: Dict[str, Tensor], h: Tensor, batch: Tensor, n_mol: int, mean: float, stddev: float,
                activation: str = "silu", z: Optional[Tensor] = None, aggregation: Optional[str] = "sum"):
    """AtomwiseV3.forward (outputs.py:186-229): y_i = MLP(h_i) * stddev [+ atomref[z_i]] (191-196); y = scatter(y_i, batch,
    reduce) or y_i (198-201); y = y + mean AFTER the aggregation (203).  ``mean`` / ``stddev`` are the constructor's
    attributes (153-157), not the ``standardize`` buffers.  -> (y, y_i)."""
    act = activation_of(activation)
    n = 0
    while f"out_net.1.out_net.{n}.weight" in head:
        n += 1
    yi = h
    for k in range(n):
        yi = F.linear(yi, head[f"out_net.1.out_net.{k}.weight"], head[f"out_net.1.out_net.{k}.bias"])
        if k + 1 < n:
            yi = act(yi)
    yi = yi * stddev
    if "atomref.weight" in head:
        yi = yi + head["atomref.weight"][z]
    if aggregation is None:
        return yi + mean, yi
    y = torch.zeros((n_mol, yi.shape[1]), dtype=yi.dtype).index_add_(0, batch, yi)
    if aggregation == "mean":
        y = y / torch.bincount(batch, minlength=n_mol).clamp(min=1).to(yi.dtype).unsqueeze(1)
    return y + mean, yi


def energy_and_forces(sd, cfg, head, z, pos, batch, n_mol, max_num_neighbors: int = 32,
                      activation: str = "silu", aggregation: str = "sum"):
    """GotenNetWrapper.forward (gotennet.py:1043-1045) + Atomwise with
    derivative (outputs.py:365-375): F = -dE/dpos."""
    pos = pos.detach().clone().requires_grad_(True)
    ei, w, vec = distance(pos, batch, cfg["cutoff"], max_num_neighbors)
    h, X = gotennet_forward(sd, cfg, z, ei, w, vec)
    e = atomwise_energy(head, h, batch, n_mol, activation, z=z, aggregation=aggregation)
    (g,) = torch.autograd.grad(e.sum(), pos)
    return e.detach(), -g, (h.detach(), X.detach(), ei)


# --------------------------------------------------------------------------- vector read-outs of the QM9 task
def gated_equivariant_block(sd: Dict[str, Tensor], prefix: str, scalars: Tensor, vectors: Tensor,
                            activation: str = "silu", sactivation: Optional[str] = None) -> Tuple[Tensor, Tensor]:
    """outputs.py:67-93: vmix = mix_vectors(vectors) -> (V, W); ctx = [scalars | ||V|| over the 3 components];
    x = scalar_net(ctx) -> (s, gate); v_out = gate * W; s_out = sactivation(s)."""
    act = activation_of(activation)
    vmix = F.linear(vectors, sd[prefix + "mix_vectors.weight"])
    n_vout = vmix.shape[-1] // 2
    V, W = vmix[..., :n_vout], vmix[..., n_vout:]
    ctx = torch.cat([scalars, torch.norm(V, dim=-2)], dim=-1)
    x = act(F.linear(ctx, sd[prefix + "scalar_net.0.weight"], sd[prefix + "scalar_net.0.bias"]))
    x = F.linear(x, sd[prefix + "scalar_net.1.weight"], sd[prefix + "scalar_net.1.bias"])
    n_sout = x.shape[-1] - n_vout
    s_out, gate = x[..., :n_sout], x[..., n_sout:]
    v_out = gate.unsqueeze(-2) * W
    if sactivation is not None:
        s_out = activation_of(sactivation)(s_out)
    return s_out, v_out


def dipole(sd: Dict[str, Tensor], h: Tensor, X: Tensor, pos: Tensor, batch: Tensor, n_mol: int,
           activation: str = "silu", mean=None, stddev=None, predict_magnitude: bool = False):
    """outputs.py:430-468 -> (y [n_mol, 3] or its norm [n_mol, 1], y_vector [n_mol, 3, 1])."""
    l0, l1 = h, X[:, :3, :]
    l0, l1 = gated_equivariant_block(sd, "equivariant_layers.0.", l0, l1, activation, activation)
    l0, l1 = gated_equivariant_block(sd, "equivariant_layers.1.", l0, l1, activation, None)
    if stddev is not None:
        l0 = stddev * l0 + mean
    y_atom = l1.squeeze(-1) + pos * l0
    y = torch.zeros((n_mol, 3), dtype=y_atom.dtype).index_add_(0, batch, y_atom)
    y_vec = torch.zeros((n_mol, 3, 1), dtype=l1.dtype).index_add_(0, batch, l1)
    if predict_magnitude:
        y = torch.norm(y, dim=1, keepdim=True)
    return y, y_vec


def electronic_spatial_extent(head: Dict[str, Tensor], h: Tensor, pos: Tensor, z: Tensor, batch: Tensor, n_mol: int,
                              activation: str = "softplus"):
    """outputs.py:516-545: x = out_net(h) (NOT standardised); c = mass-weighted centroid;
    y = sum_atoms |pos - c|^2 x.  -> (y [n_mol, 1], x [N, 1])."""
    raw = {k: v for k, v in head.items() if k.startswith("out_net.")}
    x = atomwise_contributions(raw, h, None, activation)
    mass = head["atomic_mass"].to(pos.dtype)[z].view(-1, 1)
    seg = lambda v: torch.zeros((n_mol, v.shape[1]), dtype=v.dtype).index_add_(0, batch, v)
    c = seg(mass * pos) / seg(mass)
    yi = torch.norm(pos - c[batch], dim=1, keepdim=True) ** 2 * x
    return seg(yi), x
