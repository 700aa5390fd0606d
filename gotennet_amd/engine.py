"""Launch sequencer for the HIP hot path.

Takes raw device tensors + packed weights and issues the C-ABI calls of
include/gotennet_hip.h on the current PyTorch-ROCm stream.  PyTorch is plumbing
here (device memory from the caching allocator, the stream); all arithmetic is
in libgotennet_hip.so.  Nothing in this file synchronises with the host, so a
forward can be captured in a hipGraph (torch.cuda.CUDAGraph).

Call order = the reference's op order in GotenNet.forward (gotennet.py:956-1010)
and GATA.forward (366-450) / EQFF.forward (716-748).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr


@dataclass
class LayerWeights:
    Wn1: torch.Tensor; bn1: torch.Tensor          # [W_q; W_k; gamma_s.0; gamma_v.0]  [4F, F]
    Ws2: torch.Tensor; bs2: torch.Tensor          # gamma_s.1 [MF, F]
    Wv2: torch.Tensor; bv2: torch.Tensor          # gamma_v.1 [MF, F]
    We: torch.Tensor; be: torch.Tensor            # [W_re; W_rs] [(1+M)F, F]
    Wt: Optional[torch.Tensor] = None; bt: Optional[torch.Tensor] = None   # gamma_t
    Wvq: Optional[torch.Tensor] = None
    Wvk: List[torch.Tensor] = field(default_factory=list)
    Wvu: torch.Tensor = None
    Wm0: torch.Tensor = None; bm0: torch.Tensor = None
    Wm1: torch.Tensor = None; bm1: torch.Tensor = None


@dataclass
class PackedWeights:
    A_na: torch.Tensor; A_nbr: torch.Tensor
    Winit: torch.Tensor; binit: torch.Tensor      # [W_ndp; W_erp] [2F, R]
    Wa: torch.Tensor; ba: torch.Tensor; ln_w: torch.Tensor; ln_b: torch.Tensor
    Wb: torch.Tensor; bb: torch.Tensor
    means: torch.Tensor; betas: torch.Tensor
    layers: List[LayerWeights] = field(default_factory=list)


@dataclass
class Config:
    F: int; L: int; R: int; H: int; lmax: int; M: int
    cutoff: float; eps: float
    scale_edge: bool; sep_dir: bool; sep_tensor: bool

    @property
    def D(self) -> int:
        return (self.lmax + 1) ** 2 - 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def gemm(A, lda, W, bias, C, ldc, rows, nout, K, act=(0, 0), rowmap=(1, 1, 0), res=None, gate=None, a_off=0):
    """C = epi(A W^T + bias); ``a_off`` = float offset of the first A column."""
    a_ptr = A.data_ptr() + 4 * a_off
    call("gn_gemm", a_ptr, lda, ptr(W), ptr(bias), ptr(C), ldc, rows, nout, K, act[0], act[1],
         rowmap[0], rowmap[1], rowmap[2], ptr(res), ptr(gate), _stream())


class Graph:
    """CSR-by-target view of an edge list + per-edge geometry (K1)."""

    def __init__(self, cfg: Config, pw: PackedWeights, n_atoms: int, edge_index: torch.Tensor,
                 edge_diff: torch.Tensor, edge_vec: torch.Tensor):
        dev = edge_index.device
        E = edge_index.shape[1]
        self.N, self.E = n_atoms, E
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.src = torch.empty(E, **i32)
        self.dst = torch.empty(E, **i32)
        self.rowptr = torch.empty(n_atoms + 1, **i32)
        st = _stream()
        call("gn_build_csr", ptr(edge_index), E, n_atoms, ptr(self.src), ptr(self.dst), ptr(self.rowptr), st)
        self.outdeg = None
        if cfg.scale_edge:
            self.outdeg = torch.zeros(n_atoms, **i32)
            call("gn_out_degree", ptr(self.src), E, ptr(self.outdeg), st)
        self.rl = torch.empty((E, cfg.D), **f32)
        self.phi = torch.empty((E, cfg.R), **f32)
        self.cut = torch.empty(E, **f32)
        call("gn_edge_geometry", ptr(edge_vec), ptr(edge_diff), ptr(self.src), ptr(self.dst), E,
             cfg.lmax, cfg.R, ptr(pw.means), ptr(pw.betas), float(cfg.cutoff),
             ptr(self.rl), ptr(self.phi), ptr(self.cut), st)


def forward(cfg: Config, pw: PackedWeights, z32: torch.Tensor, g: Graph,
            trace: Optional[list] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(h [N,F], X [N,D,F]) for target-sorted edges.  ``trace`` (tests only) collects
    per-layer clones of (h, X, t)."""
    F_, R, H, D, M, lmax = cfg.F, cfg.R, cfg.H, cfg.D, cfg.M, cfg.lmax
    N, E = g.N, g.E
    dev = z32.device
    f32 = dict(dtype=torch.float32, device=dev)
    st = _stream()
    new = lambda *shape: torch.empty(shape, **f32)

    # ---- init (gotennet.py:973-977) -------------------------------------------------
    feat = new(E, 2 * F_)
    gemm(g.phi, R, pw.Winit, pw.binit, feat, 2 * F_, E, 2 * F_, R)
    ctx = new(N, 2 * F_)
    call("gn_node_init", ptr(z32), ptr(g.rowptr), ptr(g.src), ptr(feat), 2 * F_, ptr(g.cut),
         ptr(pw.A_na), ptr(pw.A_nbr), N, F_, ptr(ctx), st)
    y = new(N, F_)
    gemm(ctx, 2 * F_, pw.Wa, pw.ba, y, F_, N, F_, 2 * F_)
    call("gn_layernorm_silu", ptr(y), ptr(pw.ln_w), ptr(pw.ln_b), 1e-5, N, F_, ptr(y), st)
    h = new(N, F_)
    gemm(y, F_, pw.Wb, pw.bb, h, F_, N, F_, F_)
    t = new(E, F_)
    call("gn_edge_init", ptr(h), ptr(g.rowptr), ptr(g.src), feat.data_ptr() + 4 * F_, 2 * F_, N, F_, ptr(t), st)
    del feat

    X = torch.zeros((N, D, F_), **f32)            # gotennet.py:992
    h2, X2, t2 = new(N, F_), new(N, D, F_), new(E, F_)
    nproj = new(N, 4 * F_)
    xs, vs = new(N, M * F_), new(N, M * F_)
    eproj = new(E, (1 + M) * F_)
    attn = new(E, H)
    EQ, EK, Xp = new(N, D, F_), new(N, D, F_), new(N, D, F_)
    w = new(E, F_)
    g1, mm = new(N, F_), new(N, 2 * F_)
    lde = (1 + M) * F_

    for li, lw in enumerate(pw.layers):
        # ---- GATA projections (gotennet.py:400-407)
        gemm(h, F_, lw.Wn1, lw.bn1, nproj, 4 * F_, N, 4 * F_, F_, act=(2 * F_, 4 * F_))
        gemm(nproj, 4 * F_, lw.Ws2, lw.bs2, xs, M * F_, N, M * F_, F_, a_off=2 * F_)
        gemm(nproj, 4 * F_, lw.Wv2, lw.bv2, vs, M * F_, N, M * F_, F_, a_off=3 * F_)
        gemm(t, F_, lw.We, lw.be, eproj, lde, E, lde, F_, act=(0, F_))
        # ---- message / softmax / aggregate / residual (452-559, 613-640, 426-427)
        call("gn_attn_softmax", ptr(nproj), nproj.data_ptr() + 4 * F_, 4 * F_, ptr(eproj), lde,
             ptr(g.rowptr), ptr(g.src), ptr(g.outdeg), N, F_, H, ptr(attn), st)
        call("gn_message_aggregate", ptr(xs), ptr(vs), M * F_, eproj.data_ptr() + 4 * F_, lde,
             ptr(attn), ptr(g.rl), ptr(g.cut), ptr(g.rowptr), ptr(g.src),
             ptr(h), ptr(X), ptr(h2), ptr(X2), N, F_, H, lmax, int(cfg.sep_dir), int(cfg.sep_tensor), st)
        h, h2 = h2, h
        X, X2 = X2, X
        # ---- HTR (429-445, 561-611)
        if lw.Wt is not None:
            gemm(X, F_, lw.Wvq, None, EQ, F_, N * D, F_, F_)
            off = 0
            for l in range(1, lmax + 1):
                cnt = 2 * l + 1
                gemm(X, F_, lw.Wvk[l - 1], None, EK, F_, N * cnt, F_, F_, rowmap=(cnt, D, off))
                off += cnt
            call("gn_htr_edge", ptr(EQ), ptr(EK), ptr(g.rl), ptr(g.rowptr), ptr(g.src), N, F_, lmax, ptr(w), st)
            gemm(t, F_, lw.Wt, lw.bt, t2, F_, E, F_, F_, act=(0, F_), res=t, gate=w)
            t, t2 = t2, t
        # ---- EQFF (716-748)
        gemm(X, F_, lw.Wvu, None, Xp, F_, N * D, F_, F_)
        call("gn_eqff_context", ptr(h), ptr(Xp), float(cfg.eps), N, F_, D, ptr(ctx), st)
        gemm(ctx, 2 * F_, lw.Wm0, lw.bm0, g1, F_, N, F_, 2 * F_, act=(0, F_))
        gemm(g1, F_, lw.Wm1, lw.bm1, mm, 2 * F_, N, 2 * F_, F_)
        call("gn_eqff_update", ptr(mm), ptr(Xp), N, F_, D, ptr(h), ptr(X), st)
        if trace is not None:
            trace.append((h.clone(), X.clone(), t.clone()))
    return h, X
