"""Feature widths the slot kernels do not tile natively: embed the model in the next power-of-two width.

The gather kernels give every edge a "slot" of F/4 lanes and cut a 256-thread workgroup into 1024/F slots, so F must
be a power of two; the reference accepts any ``n_atom_basis`` divisible by ``num_heads`` (gotennet.py:767-793).  A model
of width F that is not a power of two runs as a model of width Fp = 2^ceil(log2 F) whose extra channels are identically
zero: every weight is copied into a zero matrix of the padded shape, rows and columns placed by

    g(f) = (f // (F/H)) * (Fp/H) + f % (F/H)            (channel f of head h = f // (F/H) stays in head h)

-- every head boundary of the real model, for the F-wide vectors (q, k, t_attn: heads of F/H channels) and for the
flattened M*F-wide gate vectors (x, v, t_filter: heads of M*F/H channels), falls on a multiple of F/H, so ONE map keeps
every channel in its head and in its component block.  Linear maps, element-wise products, per-head sums and SiLU keep the
padded channels at exactly zero (or multiply them by zero weight columns), so the embedded model computes the real model's
numbers on the embedded channels.  Two things do not embed and are handled here:
  * the attention scale 1/sqrt(F) (gotennet.py:503-511): the kernels use 1/sqrt(Fp); sqrt(Fp/F) is folded into the last
    layer of gamma_v (a multiplies v and nothing else);
  * LayerNorm statistics (NodeInit, layers.py:1658-1675): that one intermediate stays COMPACT -- the product that feeds the
    norm writes F columns, the norm runs over F channels, the product after it reads K = F columns (engine: ``cfg.Fc``).
``GotenNet.forward`` returns (h, X) in the real layout (a column gather), the backward embeds the incoming gradients.
  * the optional GATA input norms (``layernorm``: nn.LayerNorm on h, ``steerable_norm``: TensorLayerNorm on X; gotennet.py:397-398,
    layers.py:1497-1563) take their statistics over the REAL channels: the engine gathers the real channels into a compact
    tensor, runs gn_layernorm / gn_tensor_norm (and their input-gradients) at the model's own width and scatters the result
    back into the padded layout (round 6; correct-first: two index launches per norm).
Cost: the work of the padded width (F = 192 runs as 256: +33 %).  ``edge_ln``, the composed / gated edge updates and
``evec_dim`` / ``emlp_dim`` are not embedded: NotImplementedError before any launch."""
from __future__ import annotations

import math
from dataclasses import replace
from typing import Optional

import torch

from . import engine


def padded_width(F: int) -> int:
    return max(16, 1 << (F - 1).bit_length())


def needs_embedding(F: int) -> bool:
    return F != padded_width(F)


def channel_map(F: int, H: int, device=None) -> torch.Tensor:
    """g(f) for f in [0, F): int64 [F]."""
    Fp = padded_width(F)
    f = torch.arange(F, device=device)
    return (f // (F // H)) * (Fp // H) + f % (F // H)


def check(F: int, H: int, cfg: "engine.Config") -> None:
    if F % H or F % 4 or F > 1024:
        raise NotImplementedError(f"n_atom_basis={F}: a multiple of 4 and of num_heads={H}, at most 1024, on the HIP path")
    if (cfg.composed_update or cfg.evec not in (0, F) or cfg.emlp not in (0, F) or (cfg.htr_mode >> 2)):
        raise NotImplementedError(
            f"n_atom_basis={F} is not a power of two: it runs embedded in width {padded_width(F)}, which covers the default "
            "layer family and the GATA input norms (layernorm / steerable_norm), not edge_ln, composed or gated edge updates, "
            "evec_dim / emlp_dim")


def _emb(W: Optional[torch.Tensor], idx: torch.Tensor, F: int, Fp: int, row_blocks: int = 0, col_blocks: int = 0,
         scale: float = 1.0) -> Optional[torch.Tensor]:
    """Copy W into a zero tensor whose row (dim 0) and / or column (dim 1) axis of ``blocks`` x F entries becomes
    ``blocks`` x Fp, block by block through ``idx`` (0 blocks: the axis is left alone)."""
    if W is None:
        return None
    out = W.detach()
    if scale != 1.0:
        out = out * scale
    if row_blocks:
        assert out.shape[0] == row_blocks * F
        ri = torch.cat([b * Fp + idx for b in range(row_blocks)])
        z = torch.zeros((row_blocks * Fp,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
        z.index_copy_(0, ri, out)
        out = z
    if col_blocks:
        assert out.shape[1] == col_blocks * F
        ci = torch.cat([b * Fp + idx for b in range(col_blocks)])
        z = torch.zeros((out.shape[0], col_blocks * Fp), dtype=out.dtype, device=out.device)
        z.index_copy_(1, ci, out)
        out = z
    return out.contiguous()


def embed_pack(pw: "engine.PackedWeights", F: int, H: int, M: int) -> "engine.PackedWeights":
    """The packed weights of the width-F model as those of the equivalent width-Fp model (module docstring)."""
    Fp = padded_width(F)
    idx = channel_map(F, H, pw.A_na.device)
    e = lambda W, r=0, c=0, s=1.0: _emb(W, idx, F, Fp, r, c, s)
    out = engine.PackedWeights(
        A_na=e(pw.A_na, 0, 1), A_nbr=e(pw.A_nbr, 0, 1),
        Winit=e(pw.Winit, 2, 0), binit=e(pw.binit, 2, 0),
        Wa=e(pw.Wa, 0, 2), ba=pw.ba, ln_w=pw.ln_w, ln_b=pw.ln_b,        # compact F outputs: the LayerNorm runs over the real channels
        Wb=e(pw.Wb, 1, 0), bb=e(pw.bb, 1, 0),
        rb0=pw.rb0, rb1=pw.rb1)
    vs = math.sqrt(Fp / F)                                               # 1/sqrt(F) of the attention, folded into gamma_v.1
    for lw in pw.layers:
        nl = engine.LayerWeights(
            Wn1=e(lw.Wn1, 4, 1), bn1=e(lw.bn1, 4, 0),
            Ws2=e(lw.Ws2, M, 1), bs2=e(lw.bs2, M, 0),
            Wv2=e(lw.Wv2, M, 1, vs), bv2=e(lw.bv2, M, 0, vs),
            We=e(lw.We, 1 + M, 1), be=e(lw.be, 1 + M, 0))
        nl.Wvu = e(lw.Wvu, 1, 1)
        nl.Wm0, nl.bm0 = e(lw.Wm0, 1, 2), e(lw.bm0, 1, 0)
        nl.Wm1, nl.bm1 = e(lw.Wm1, 2, 1), e(lw.bm1, 2, 0)
        # the GATA input norms (layernorm / steerable_norm) run on the COMPACT real channels (engine._input_norms): their
        # parameters stay in the model's own layout
        nl.ln_w, nl.ln_b, nl.tln_w = lw.ln_w, lw.ln_b, lw.tln_w
        if lw.Wt is not None:
            nl.Wt, nl.bt = e(lw.Wt, 1, 1), e(lw.bt, 1, 0)
            nl.Wvq = e(lw.Wvq, 1, 1)
            nl.Wvk = [e(w, 1, 1) for w in lw.Wvk]
        out.layers.append(nl)
    out.emb_idx, out.F_model = idx, F
    return out


def embedded_config(cfg: "engine.Config") -> "engine.Config":
    return replace(cfg, F=padded_width(cfg.F), Fc=cfg.F, F_model=cfg.F, evec=0, emlp=0)


def unembed(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """[..., Fp] -> [..., F] (the real channels)."""
    return t.index_select(t.dim() - 1, idx)


def embed_grad(g: torch.Tensor, idx: torch.Tensor, Fp: int) -> torch.Tensor:
    """[..., F] -> [..., Fp] with zeros in the padding."""
    z = torch.zeros(tuple(g.shape[:-1]) + (Fp,), dtype=g.dtype, device=g.device)
    z.index_copy_(g.dim() - 1, idx, g.contiguous())
    return z
