"""Launch sequencer for the HIP hot path (forward, and the force backward).

Takes raw device tensors + packed weights and issues the C-ABI calls of
include/gotennet_hip.h on the current PyTorch-ROCm stream.  PyTorch is plumbing
here (device memory from the caching allocator, the stream, integer index
sorting for the CSC view); all floating-point arithmetic is in
libgotennet_hip.so.  Nothing in ``forward``/``backward`` synchronises with the
host, so a step can be captured in a hipGraph (torch.cuda.CUDAGraph).

Call order = the reference's op order in GotenNet.forward (gotennet.py:956-1010),
GATA.forward (366-450) and EQFF.forward (716-748); ``backward`` walks it in
reverse (what torch.autograd.grad does for the reference at outputs.py:365-375).
The backward's copies of activations are PRE-activation (SiLU' needs them); the forward applies SiLU once, in the
producing GEMM's epilogue.  Independent projections are issued as grouped launches (gn_gemm_group).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr


@dataclass
class LayerWeights:
    Wn1: torch.Tensor; bn1: torch.Tensor          # [W_q; W_k; gamma_s.0; gamma_v.0]  [4F, F]
    Ws2: torch.Tensor; bs2: torch.Tensor          # gamma_s.1 [MF, F]
    Wv2: torch.Tensor; bv2: torch.Tensor          # gamma_v.1 [MF, F]
    We: torch.Tensor; be: torch.Tensor            # [W_re; W_rs] [(1+M)F, F]
    Wvu: Optional[torch.Tensor] = None                                         # EQFF (None in a stand-alone GATA pack)
    Wm0: Optional[torch.Tensor] = None; bm0: Optional[torch.Tensor] = None
    Wm1: Optional[torch.Tensor] = None; bm1: Optional[torch.Tensor] = None
    Wt: Optional[torch.Tensor] = None; bt: Optional[torch.Tensor] = None   # gamma_t
    Wvq: Optional[torch.Tensor] = None
    Wvk: List[torch.Tensor] = field(default_factory=list)   # one per degree, or ONE shared weight (sep_htr=False)
    ln_w: Optional[torch.Tensor] = None; ln_b: Optional[torch.Tensor] = None   # optional nn.LayerNorm on h
    tln_w: Optional[torch.Tensor] = None                                       # optional TensorLayerNorm weight
    # composed edge update (edge_updates "mlp"/"mlpa"/"linw"/"linwa"/"ln"/"postln", gotennet.py:236-291)
    Wt0: Optional[torch.Tensor] = None; bt0: Optional[torch.Tensor] = None     # gamma_t hidden layer
    t_ln_w: Optional[torch.Tensor] = None; t_ln_b: Optional[torch.Tensor] = None   # its LayerNorm (edge_ln)
    Wedp: Optional[torch.Tensor] = None; bedp: Optional[torch.Tensor] = None   # W_edp
    w_ln_w: Optional[torch.Tensor] = None; w_ln_b: Optional[torch.Tensor] = None   # LayerNorm before / after W_edp
    # prefixes for the first interaction (X_in = 0: no tensor-gate blocks), views made on first use
    We0: Optional[torch.Tensor] = None; be0: Optional[torch.Tensor] = None
    Ws20: Optional[torch.Tensor] = None; Wv20: Optional[torch.Tensor] = None
    T: dict = field(default_factory=dict)         # lazily built transposes for the backward


@dataclass
class PackedWeights:
    A_na: torch.Tensor; A_nbr: torch.Tensor
    Winit: torch.Tensor; binit: torch.Tensor      # [W_ndp; W_erp] [2F, R]
    Wa: torch.Tensor; ba: torch.Tensor; ln_w: torch.Tensor; ln_b: torch.Tensor
    Wb: torch.Tensor; bb: torch.Tensor
    rb0: torch.Tensor; rb1: torch.Tensor          # radial-basis parameter vectors (means/betas, freqs, offsets/widths)
    layers: List[LayerWeights] = field(default_factory=list)
    T: dict = field(default_factory=dict)
    emb_idx: Optional[torch.Tensor] = None        # model embedded in a power-of-two width (embed.py): real channel f sits at emb_idx[f]
    F_model: int = 0                              # ... and its real width (0: not embedded)


#: A/B and test switch: False runs the first interaction through the general kernels on the zero tensor
ZERO_X_FIRST = os.environ.get("GN_ZERO_X_FIRST", "1") != "0"
#: test switch: True withholds the head-sum workspace from gn_message_backward where the entry point allows it (monolithic
#: launches and every first interaction), which then runs the by-target / by-source kernel pair instead of the merged kernel
MSG_BWD_PAIR = False


def zero_X_in(cfg: "Config", li: int) -> bool:
    """Layer ``li`` of ``forward`` sees the all-zero X that forward itself creates (gotennet.py:992) and the kernels have
    the zero-X_in form (register-tiled SiLU kernels, lmax <= 4): every tensor-gate term of that layer is 0 * gate, so its
    blocks of the edge projection are neither computed nor read, and nothing consumes the gradient w.r.t. X_in."""
    return (ZERO_X_FIRST and li == 0 and cfg.lmax <= 4 and cfg.act == 0 and not cfg.steerable_norm and not cfg.sliced
            and cfg.aggr == 0 and not cfg.wide)


def _We_first(cfg: "Config", lw) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Rows of [W_re; W_rs] without the tensor-gate blocks (a prefix: attention, scalar, direction gates)."""
    n0 = (2 + (cfg.lmax if cfg.sep_dir else 1)) * cfg.F
    if lw.We0 is None:
        lw.We0, lw.be0 = lw.We[:n0], (lw.be[:n0] if lw.be is not None else None)
    return lw.We0, lw.be0, n0


def _value_first(cfg: "Config", lw) -> int:
    """Rows of gamma_s.1 / gamma_v.1 without the tensor-gate blocks (a prefix: scalar, direction gates)."""
    n0 = (1 + (cfg.lmax if cfg.sep_dir else 1)) * cfg.F
    if lw.Ws20 is None:
        lw.Ws20, lw.Wv20 = lw.Ws2[:n0], lw.Wv2[:n0]
    return n0


def _Tqk(lw) -> torch.Tensor:
    """Transposed q | k rows of W_n1 ([F, 2F]) for the input-gradient product, cached."""
    t = lw.T.get("Wn1_qk")
    if t is None:
        F2 = lw.Wn1.shape[0] // 2
        t = lw.T["Wn1_qk"] = lw.Wn1[:F2].t().contiguous()
    return t


def _Tsv(lw) -> torch.Tensor:
    """Transposed gamma_s.0 | gamma_v.0 rows of W_n1 ([F, 2F]), cached."""
    t = lw.T.get("Wn1_sv")
    if t is None:
        F2 = lw.Wn1.shape[0] // 2
        t = lw.T["Wn1_sv"] = lw.Wn1[F2:].t().contiguous()
    return t


def _T(holder, name: str) -> torch.Tensor:
    """Transposed copy ([in, out] -> the GEMM's [out', in'] layout for input-gradients), cached."""
    t = holder.T.get(name)
    if t is None:
        t = getattr(holder, name).t().contiguous()
        holder.T[name] = t
    return t


@dataclass
class Config:
    F: int; L: int; R: int; H: int; lmax: int; M: int
    cutoff: float; eps: float
    scale_edge: bool; sep_dir: bool; sep_tensor: bool
    basis: int = 0            # gn_edge_geometry basis code: 0 expnorm, 1 Bessel, 2 Gaussian
    htr_mode: int = 0         # GN_HTR_* bits (sep_htr=False, "norej", gamma_w gate)
    layernorm: bool = False   # nn.LayerNorm on h at the GATA input (gotennet.py:397)
    steerable_norm: bool = False   # TensorLayerNorm on X at the GATA input (gotennet.py:398)
    composed_update: bool = False  # gamma_t 2-layer MLP and/or W_edp in gamma_w: sequenced by _edge_update_composed
    gate_kind: int = 0        # gamma_w's final element-wise gate: 0 none, 1 sigmoid, 2 tanh, 3 SiLU
    t_last_act: int = 3       # gamma_t's last layer: 3 = activated (with ``act``), 0 = linear ("mlp")
    lin_w: int = 0            # 0 no W_edp, 1 "linw", 2 "linwa" (SiLU before W_edp)
    lin_ln: int = 0           # 0 none, 1 "ln" (LayerNorm before W_edp), 2 "postln" (inside the W_edp Dense)
    act: int = 0              # GN_ACT_* kind of the ``activation`` argument (0 = SiLU / swish, the reference default)
    evec: int = 0             # evec_dim: width of EQ / EK / w (0 = F; != F needs W_edp to map back to F)
    emlp: int = 0             # emlp_dim: hidden width of the 2-layer gamma_t (0 = F)
    gemm_mode: str = ""       # projection arithmetic of THIS model ("f16x2" | "split" | "f32"; "" = engine.GEMM_MODE, the default)
    sliced: bool = False      # run lmax <= 4 on the degree-sliced kernel family too (GN_LMAX_SLICED in the lmax argument)
    aggr: int = 0             # the reference's `aggr` (gotennet.py:84,638): 0 "add", 1 "mean", 2 "max"
    Fc: int = 0               # width of the NodeInit LayerNorm intermediate when the model is embedded in a power-of-two F (embed.py; 0 = F)
    F_model: int = 0          # the model's real n_atom_basis when embedded (0 = F)
    fuse_eqff: Optional[bool] = None   # the node-local EQFF chain as ONE kernel each way where covered (eqff_fused_ok).  None =
                              # auto: on for systems of at most EQFF_FUSED_MAX_ATOMS atoms (launch-bound: -17 % on a one-molecule
                              # step, -2 % at 32 molecules), off above (a wash at the 128-molecule batch)

    @property
    def Fe(self) -> int:
        return self.evec or self.F

    @property
    def Fm(self) -> int:
        return self.emlp or self.F

    @property
    def D(self) -> int:
        return (self.lmax + 1) ** 2 - 1

    @property
    def lmax_arg(self) -> int:
        """The ``lmax`` argument of the message / HTR entry points: GN_LMAX_SLICED rides in it."""
        return self.lmax | (_lib.LMAX_SLICED if self.sliced else 0)

    @property
    def wide(self) -> bool:
        """A slot (one edge row, F / 4 lanes) spans several waves: the input-gradient kernels are the degree-sliced family
        and the per-edge scalar gradients come as F / 256 partial slices per call."""
        return self.F > 256 or self.Fe > 256

    @property
    def lmax_arg_bwd(self) -> int:
        """... of the BACKWARD entry points: F > 256 runs the degree-sliced kernels there."""
        return self.lmax_arg | (_lib.LMAX_SLICED if self.wide else 0)

    @property
    def lmax_arg_msg_bwd(self) -> int:
        return self.lmax_arg_msg | (_lib.LMAX_SLICED if self.wide else 0)

    @property
    def lmax_arg_msg(self) -> int:
        """... of the two MESSAGE entry points, which also carry the aggregation (GN_LMAX_MEAN / GN_LMAX_MAX)."""
        return self.lmax_arg | {0: 0, 1: _lib.LMAX_MEAN, 2: _lib.LMAX_MAX}[self.aggr]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The current stream's handle.  (``torch.cuda.current_stream().cuda_stream`` builds a Stream object per call: 4 us, a
    hundred times per step -- a fifth of the host time of a one-molecule eager step.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


#: DEFAULT projection arithmetic of a model that does not choose one (``GotenNet.gemm_mode = None``; env GN_GEMM_MODE
#: sets the default at import; every GPU parity test runs in all three).  The arithmetic and the activation kind of a
#: call are carried by ``Config`` (``cfg.gemm_mode``, ``cfg.act``) and bound per call by ``_Proj`` -- there is no
#: per-call module state, so two models with different arithmetics can run from two threads.
#:   "f16x2" (default) -- every fp32 operand as two fp16 planes scaled by block exponents (A: per staging wave and
#:       32-column K-slab = 16 or 32 neighbouring rows of the workgroup tile, running maximum, accumulators rescaled when
#:       it grows; W: per tensor), THREE fp16 MFMAs per product, fp32 accumulate: <= 3e-7 of the output's max-norm vs an
#:       fp64 product.  Results depend on which rows share a group at the 1e-7 level (a molecule's energy moves by ~1e-6
#:       relative when its position in the batch, or the batch size, changes); identical inputs give identical bits.
#:   "split" -- three bf16 planes, six bf16 MFMAs per product: same error class, row-wise arithmetic independent of the
#:       batch layout (bit-exact batch independence), 14 % slower on the C2 step.
#:   "f32"   -- exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain), 40 % slower.
GEMM_MODE = os.environ.get("GN_GEMM_MODE", "f16x2")
MODES = ("f16x2", "split", "f32")


#: projection arithmetic -> (single-product entry point, grouped entry point, weight packer, its size query, dtype)
_PLANE_MODES = {"split": ("gn_gemm_split", "gn_gemm_group_split", "gn_split_bf16x3", "gn_split_bf16x3_size", torch.bfloat16),
                "f16x2": ("gn_gemm_f16x2", "gn_gemm_group_f16x2", "gn_split_f16x2", "gn_split_f16x2_size", torch.float16)}


def resolve_mode(mode: Optional[str]) -> str:
    mode = mode or GEMM_MODE
    if mode not in MODES:
        raise ValueError(f"projection arithmetic {mode!r}: one of {MODES}")
    return mode


def split_weight(W: torch.Tensor, mode: Optional[str] = None) -> torch.Tensor:
    """Operand planes of a weight [N, K] for the arithmetic ``mode`` in MFMA-fragment-major order (gn_split_bf16x3: bf16
    hi/mid/lo; gn_split_f16x2: header + fp16 hi/lo), cached ON the tensor object per mode (the packed weights are
    long-lived; GotenNet.invalidate_packed() drops them with the pack)."""
    mode = resolve_mode(mode)
    key = "_gn_split_" + mode
    cached = getattr(W, key, None)
    if cached is not None and cached[0] == W._version and cached[1] == W.data_ptr():
        return cached[2]
    _, _, packer, sizer, dt = _PLANE_MODES[mode]
    N, K = W.shape
    planes = torch.empty(getattr(_lib.load(), sizer)(N, K), dtype=dt, device=W.device)
    call(packer, ptr(W.contiguous()), N, K, ptr(planes), _stream())
    setattr(W, key, (W._version, W.data_ptr(), planes))
    return planes


def gemm(A, lda, W, bias, C, ldc, rows, nout, K, act=(0, 0), rowmap=(1, 1, 0), res=None, gate=None,
         a_off=0, c_off=0, pre_out=None, pro=(0, 0, 0), a_pre=None, ldp=0, p_off=0, a_gate=None, ldg=0,
         dgate=None, g_off=0, kind=None, mode=None):
    """C = epi(pro(A) W^T + bias); ``kind``: GN_ACT_* of the activated columns / SiLU' gates / prologues (None = SiLU);
    ``mode``: projection arithmetic (None = the module default).  ``a_off`` / ``c_off`` / ``p_off`` / ``g_off``: float
    offsets of the first column.  ``dgate``: multiply the output by SiLU'(dgate) (same addressing as C)."""
    mode = resolve_mode(mode)
    name = "gn_gemm_ex"
    if mode in _PLANE_MODES:
        name, W = _PLANE_MODES[mode][0], split_weight(W, mode)
    call(name, A.data_ptr() + 4 * a_off, lda, ptr(W), ptr(bias), C.data_ptr() + 4 * c_off, ldc,
         rows, nout, K, act[0], act[1], rowmap[0], rowmap[1], rowmap[2], ptr(res),
         (dgate.data_ptr() + 4 * g_off) if dgate is not None else ptr(gate), 1 if dgate is not None else 0, ptr(pre_out),
         pro[0], pro[1], pro[2], (a_pre.data_ptr() + 4 * p_off) if a_pre is not None else None, ldp,
         ptr(a_gate), ldg, 0 if kind is None else kind, _stream())


def gemm_group(problems, mode=None, kind=None):
    """Several INDEPENDENT ``gemm(...)`` calls (a list of argument dicts) as ONE launch (gn_gemm_group, or
    gn_gemm_group_split / _f16x2 with the weights replaced by their cached planes)."""
    mode = resolve_mode(mode)
    kind = 0 if kind is None else kind
    problems = [q for q in problems if q is not None]
    split = mode in _PLANE_MODES
    for i0 in range(0, len(problems), 4):
        chunk = problems[i0:i0 + 4]
        arr = (_lib.GemmDesc * len(chunk))()
        for d, q in zip(arr, chunk):
            g = q.get                                # (a lambda around it was 39 Python calls per problem, 41 groups per step)
            act, rowmap, pro = g("act", (0, 0)), g("rowmap", (1, 1, 0)), g("pro", (0, 0, 0))
            dgate = g("dgate")
            d.A = q["A"].data_ptr() + 4 * g("a_off", 0); d.lda = q["lda"]
            d.W = ptr(split_weight(q["W"], mode) if split else q["W"]); d.bias = ptr(g("bias"))
            d.C = q["C"].data_ptr() + 4 * g("c_off", 0); d.ldc = q["ldc"]
            d.M, d.N, d.K = q["rows"], q["nout"], q["K"]
            d.act_lo, d.act_hi = act
            d.row_cnt, d.row_gstride, d.row_goff = rowmap
            d.res = ptr(g("res"))
            d.gate = (dgate.data_ptr() + 4 * g("g_off", 0)) if dgate is not None else ptr(g("gate"))
            d.gate_mode = 1 if dgate is not None else 0
            d.pre_out = ptr(g("pre_out"))
            d.pro_mode, d.pro_lo, d.pro_hi = pro
            a_pre = g("a_pre")
            d.a_pre = (a_pre.data_ptr() + 4 * g("p_off", 0)) if a_pre is not None else None
            d.ldp = g("ldp", 0)
            d.a_gate = ptr(g("a_gate")); d.ldg = g("ldg", 0)
            d.A2, d.A3, d.a_seg = ptr(g("A2")), ptr(g("A3")), g("a_seg", 0)
            d.act_kind = g("kind", kind)
        call(_PLANE_MODES[mode][1] if split else "gn_gemm_group", arr, len(chunk), _stream())


class _Proj:
    """The projection launchers bound to ONE model's arithmetic and activation kind (``cfg.gemm_mode``, ``cfg.act``)."""
    __slots__ = ("mode", "act")

    def __init__(self, cfg: "Config"):
        self.mode, self.act = resolve_mode(cfg.gemm_mode), cfg.act

    def gemm(self, *args, **kw):
        kw.setdefault("kind", self.act)
        kw.setdefault("mode", self.mode)
        return gemm(*args, **kw)

    def group(self, problems):
        return gemm_group(problems, mode=self.mode, kind=self.act)


def validate_edges(edge_index: torch.Tensor, n_atoms: int) -> int:
    """Bit 0: ``edge_index[1]`` is not non-decreasing (needs a stable sort by target); bit 1: an index is outside
    [0, n_atoms).  One tiny kernel + ONE host read of its flag (a stream synchronisation)."""
    flag = torch.zeros(1, dtype=torch.int32, device=edge_index.device)
    call("gn_check_edges", ptr(edge_index), edge_index.shape[1], n_atoms, ptr(flag), _stream())
    return int(flag.item())


def sorted_edges(edge_index, edge_diff, edge_vec, n_atoms: int):
    """Validate a caller-supplied edge list; returns it target-major (stable: the order inside a target row is kept)
    plus the permutation applied (None when it already was).  Raises on out-of-range indices."""
    if edge_index.shape[1] == 0:
        return edge_index, edge_diff, edge_vec, None
    bits = validate_edges(edge_index, n_atoms)
    if bits & 2:
        raise ValueError(f"edge_index holds indices outside [0, {n_atoms})")
    if not bits & 1:
        return edge_index, edge_diff, edge_vec, None
    order = torch.sort(edge_index[1], stable=True).indices
    return edge_index[:, order].contiguous(), edge_diff[order], edge_vec[order], order


class Graph:
    """CSR-by-target view of a target-sorted edge list + per-edge geometry (K1);
    ``csc()`` adds the by-source view the backward needs.  Topology (index arrays) and geometry (rl, phi, cut)
    are separate: ``set_geometry`` re-evaluates the geometry of a fixed edge list (static-topology steps that are
    replayed from a hipGraph, pipeline.CapturedStep)."""

    def __init__(self, cfg: Config, pw: PackedWeights, n_atoms: int, edge_index: torch.Tensor,
                 edge_diff: Optional[torch.Tensor] = None, edge_vec: Optional[torch.Tensor] = None):
        dev = edge_index.device
        E = edge_index.shape[1]
        self.N, self.E = n_atoms, E
        self.cfg, self.pw = cfg, pw
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.src = torch.empty(E, **i32)
        self.dst = torch.empty(E, **i32)
        self.rowptr = torch.empty(n_atoms + 1, **i32)
        call("gn_build_csr", ptr(edge_index), E, n_atoms, ptr(self.src), ptr(self.dst), ptr(self.rowptr), _stream())
        self.outdeg = None
        if cfg.scale_edge:
            self.outdeg = torch.zeros(n_atoms, **i32)
            call("gn_out_degree", ptr(self.src), E, ptr(self.outdeg), _stream())
        self.rl = torch.empty((E, cfg.D), **f32)
        self.phi = torch.empty((E, max(cfg.R, 0)), **f32)
        self.cut = torch.empty(E, **f32)
        self.perm = self.colptr = self.tgt_by_src = None
        self.edge_diff = self.edge_vec = None
        if edge_vec is not None:
            self.set_geometry(edge_diff, edge_vec)

    def set_geometry(self, edge_diff: torch.Tensor, edge_vec: torch.Tensor):
        """K1 on the current edge list: unit vectors, real harmonics, radial basis, cutoff."""
        cfg, pw = self.cfg, self.pw
        self.edge_diff, self.edge_vec = edge_diff, edge_vec
        call("gn_edge_geometry", ptr(edge_vec), ptr(edge_diff), ptr(self.src), ptr(self.dst), self.E,
             cfg.lmax, cfg.R, cfg.basis, ptr(pw.rb0), ptr(pw.rb1), float(cfg.cutoff),
             ptr(self.rl), ptr(self.phi), ptr(self.cut), _stream())

    def set_positions(self, pos: torch.Tensor):
        """Edge vectors of the fixed edge list for new positions (gn_edge_vectors), then the geometry."""
        if self.edge_vec is None:
            self.edge_vec = torch.empty((self.E, 3), dtype=torch.float32, device=pos.device)
            self.edge_diff = torch.empty(self.E, dtype=torch.float32, device=pos.device)
        call("gn_edge_vectors", ptr(pos), ptr(self.src), ptr(self.dst), self.E, ptr(self.edge_vec), ptr(self.edge_diff),
             _stream())
        self.set_geometry(self.edge_diff, self.edge_vec)

    def csc(self):
        """Edges grouped by source (stable order: gn_build_csc), no host sync."""
        if self.perm is None:
            i32 = dict(dtype=torch.int32, device=self.src.device)
            self.colptr, self.perm = torch.empty(self.N + 1, **i32), torch.empty(self.E, **i32)
            self.tgt_by_src = torch.empty(self.E, **i32)        # target of each by-source entry
            work = torch.empty(self.N + self.E, **i32)
            call("gn_build_csc", ptr(self.src), ptr(self.dst), self.E, self.N, ptr(self.colptr), ptr(self.perm),
                 ptr(self.tgt_by_src), ptr(work), _stream())
        return self.colptr, self.perm


@dataclass
class LayerTape:
    h_in: torch.Tensor = None; X_in: torch.Tensor = None; t_in: torch.Tensor = None
    nproj: torch.Tensor = None; xs: torch.Tensor = None; vs: torch.Tensor = None
    eproj: torch.Tensor = None; attn: torch.Tensor = None
    EQ: torch.Tensor = None; EK: torch.Tensor = None; w: torch.Tensor = None; pre_t: torch.Tensor = None
    w_raw: torch.Tensor = None; h_raw: torch.Tensor = None; X_raw: torch.Tensor = None
    upd: dict = None                               # intermediates of the composed edge update
    Xp: torch.Tensor = None; ctx: torch.Tensor = None; pre_g1: torch.Tensor = None; mm: torch.Tensor = None


@dataclass
class Tape:
    feat: torch.Tensor = None; y_pre: torch.Tensor = None; h0: torch.Tensor = None
    layers: List[LayerTape] = field(default_factory=list)


def check_backward_supported(cfg: Config) -> None:
    """Raise NotImplementedError -- BEFORE any launch -- for the configurations whose force path does not exist."""
    if cfg.wide and (cfg.F // 4) // cfg.H > 64:
        raise NotImplementedError(f"n_atom_basis={cfg.F_model or cfg.F} with {cfg.H} heads: the input-gradient kernels keep one "
                                  "attention head inside one wave (at most 256 channels per head)")


def _forward_impl(cfg: Config, pw: PackedWeights, z32: torch.Tensor, g: Graph, save: bool = False,
            trace: Optional[list] = None):
    """-> (h [N,F], X [N,D,F], tape or None).  ``save`` keeps what ``backward`` needs;
    ``trace`` (tests only) collects per-layer clones of (h, X, t)."""
    F_, R, H, D, M, lmax = cfg.F, cfg.R, cfg.H, cfg.D, cfg.M, cfg.lmax
    Fe = cfg.Fe
    N, E = g.N, g.E
    if save:
        check_backward_supported(cfg)               # before any launch: a saving forward is only run for a backward
    proj = _Proj(cfg)
    gemm, gemm_group = proj.gemm, proj.group
    dev = z32.device
    f32 = dict(dtype=torch.float32, device=dev)
    new = lambda *shape: torch.empty(shape, **f32)
    tape = Tape() if save else None

    # ---- init (gotennet.py:973-977) -------------------------------------------------
    feat = new(E, 2 * F_)
    gemm(g.phi, R, pw.Winit, pw.binit, feat, 2 * F_, E, 2 * F_, R)
    ctx0 = new(N, 2 * F_)
    call("gn_node_init", ptr(z32), ptr(g.rowptr), ptr(g.src), ptr(feat), 2 * F_, ptr(g.cut),
         ptr(pw.A_na), ptr(pw.A_nbr), N, F_, ptr(ctx0), _stream())
    Fc = cfg.Fc or F_                              # (an embedded model keeps this intermediate compact: LayerNorm over the real channels)
    y_pre = new(N, Fc)
    gemm(ctx0, 2 * F_, pw.Wa, pw.ba, y_pre, Fc, N, Fc, 2 * F_)
    y = new(N, Fc)
    call("gn_layernorm_silu", ptr(y_pre), ptr(pw.ln_w), ptr(pw.ln_b), 1e-5, N, Fc, ptr(y), cfg.act, _stream())
    h = new(N, F_)
    gemm(y, Fc, pw.Wb, pw.bb, h, F_, N, F_, Fc)
    t = new(E, F_)
    call("gn_edge_init", ptr(h), ptr(g.rowptr), ptr(g.src), feat.data_ptr() + 4 * F_, 2 * F_, N, F_, ptr(t), _stream())
    if save:
        tape.feat, tape.y_pre, tape.h0 = feat, y_pre, h

    # gotennet.py:992: X starts as the zero tensor.  Where the first interaction runs the zero-X_in kernels nothing ever reads
    # it (message stage and message backward get a null X_in): no fill launch
    X = new(N, D, F_) if (zero_X_in(cfg, 0) and pw.layers) else torch.zeros((N, D, F_), **f32)
    lde = (1 + M) * F_
    nact, g1act = new(N, 4 * F_), new(N, F_)       # activated copies (scratch, shared by all layers)
    eq_fused, eq_arith = eqff_fused_ok(cfg, N), (1 if proj.mode == "split" else 2)
    if not save:                                   # inference: ping-pong work buffers, reused by every layer
        h2, X2, t2 = new(N, F_), new(N, D, F_), new(E, F_)
        nproj, xs, vs = new(N, 4 * F_), new(N, M * F_), new(N, M * F_)
        eproj, attn = new(E, lde), new(E, H)
        EQ, EK, Xp, w = new(N, D, Fe), new(N, D, Fe), new(N, D, F_), new(E, Fe)
        ctx, pre_g1, mm = new(N, 2 * F_), new(N, F_), new(N, 2 * F_)

    for li, lw in enumerate(pw.layers):
        last = lw.Wt is None
        h_raw, X_raw = h, X
        if cfg.layernorm:                          # gotennet.py:397-398: the layer (and its residuals) see the normalised values
            h = _norm_h(cfg, pw, lw, h)
        if cfg.steerable_norm:
            X = _norm_X(cfg, pw, lw, X)
        if save:                                   # every layer keeps its own activations
            lt = LayerTape(h_in=h, X_in=X, t_in=t, h_raw=h_raw, X_raw=X_raw)
            tape.layers.append(lt)
            h2, X2 = new(N, F_), new(N, D, F_)
            t2 = None if last else new(E, F_)
            nproj, xs, vs = new(N, 4 * F_), new(N, M * F_), new(N, M * F_)
            eproj, attn = new(E, lde), new(E, H)
            Xp, ctx, pre_g1, mm = new(N, D, F_), new(N, 2 * F_), new(N, F_), new(N, 2 * F_)
            if not last:
                EQ, EK, w = new(N, D, Fe), new(N, D, Fe), new(E, Fe)
                lt.pre_t = new(E, F_)
                lt.EQ, lt.EK, lt.w = EQ, EK, w
                lt.w_raw = new(E, Fe) if (cfg.htr_mode >> 2) else None
            lt.nproj, lt.xs, lt.vs, lt.eproj, lt.attn = nproj, xs, vs, eproj, attn
            lt.Xp, lt.ctx, lt.pre_g1, lt.mm = Xp, ctx, pre_g1, mm
        # ---- GATA projections (gotennet.py:400-407).  The atom-sized node projection rides in the edge projection's
        # launch (its 168 tiles fill the tail of the 5100-tile grid).  SiLU of the two hidden blocks is applied ONCE by
        # the epilogue (a SiLU prologue in the two products below would redo it for each of their 4M column tiles); the
        # pre-activation copy is what the backward needs.
        first = zero_X_in(cfg, li)                  # X is the zero tensor made above: no tensor-gate blocks
        We, be, ne = _We_first(cfg, lw) if first else (lw.We, lw.be, lde)
        gemm_group([dict(A=t, lda=F_, W=We, bias=be, C=eproj, ldc=lde, rows=E, nout=ne, K=F_),
                    dict(A=h, lda=F_, W=lw.Wn1, bias=lw.bn1, C=nact, ldc=4 * F_, rows=N, nout=4 * F_, K=F_,
                         act=(2 * F_, 4 * F_), pre_out=nproj if save else None)])
        nv = _value_first(cfg, lw) if first else M * F_
        gemm_group([dict(A=nact, lda=4 * F_, W=lw.Ws20 if first else lw.Ws2, bias=lw.bs2, C=xs, ldc=M * F_, rows=N, nout=nv,
                         K=F_, a_off=2 * F_),
                    dict(A=nact, lda=4 * F_, W=lw.Wv20 if first else lw.Wv2, bias=lw.bv2, C=vs, ldc=M * F_, rows=N, nout=nv,
                         K=F_, a_off=3 * F_)])
        # ---- message / softmax / aggregate / residual (452-559, 613-640, 426-427)
        message_stage(cfg, g, nact, xs, vs, eproj, attn, h, None if first else X, h2, X2)
        h, h2 = h2, h
        X, X2 = X2, X
        # every product of the updated X (X W_vu^T for EQFF; EQ and the per-degree EK_l for HTR) in one launch
        xprods = [dict(A=X, lda=F_, W=lw.Wvu, C=Xp, ldc=F_, rows=N * D, nout=F_, K=F_)]
        if not last:
            xprods.append(dict(A=X, lda=F_, W=lw.Wvq, C=EQ, ldc=Fe, rows=N * D, nout=Fe, K=F_))
            if cfg.htr_mode & 1:                   # sep_htr=False: one W_vk for every row
                xprods.append(dict(A=X, lda=F_, W=lw.Wvk[0], C=EK, ldc=Fe, rows=N * D, nout=Fe, K=F_))
            else:
                off = 0
                for l in range(1, lmax + 1):
                    cnt = 2 * l + 1
                    xprods.append(dict(A=X, lda=F_, W=lw.Wvk[l - 1], C=EK, ldc=Fe, rows=N * cnt, nout=Fe, K=F_,
                                       rowmap=(cnt, D, off)))
                    off += cnt
        gemm_group(xprods)
        # ---- EQFF (731-746) and HTR edge weights (561-611).  Where covered the EQFF chain after X_p is ONE kernel
        # (context, both gamma_m layers, update); else: context kernel, the first gamma_m layer riding in the launch of the
        # edge-sized gamma_t product, the second layer, update kernel
        m0 = None
        if eq_fused:
            call("gn_eqff_fused_forward", ptr(Xp), ptr(split_weight(lw.Wm0, proj.mode)), ptr(lw.bm0),
                 ptr(split_weight(lw.Wm1, proj.mode)), ptr(lw.bm1), float(cfg.eps), N, F_, D, ptr(h), ptr(X),
                 ptr(ctx) if save else None, ptr(pre_g1) if save else None, ptr(mm) if save else None, eq_arith, _stream())
        else:
            call("gn_eqff_context", ptr(h), ptr(Xp), float(cfg.eps), N, F_, D, ptr(ctx), _stream())
            m0 = dict(A=ctx, lda=2 * F_, W=lw.Wm0, bias=lw.bm0, C=g1act, ldc=F_, rows=N, nout=F_, K=2 * F_, act=(0, F_),
                      pre_out=pre_g1 if save else None)
        if not last:
            call("gn_htr_edge", ptr(EQ), ptr(EK), ptr(g.rl), ptr(g.rowptr), ptr(g.src), N, Fe, cfg.lmax_arg, cfg.htr_mode,
                 ptr(lt.w_raw) if save else None, ptr(w), _stream())
            if cfg.composed_update:
                upd = _edge_update_composed(cfg, lw, t, w, t2, E, lt.pre_t if save else None)
                if save:
                    lt.upd = upd
                gemm_group([m0])
            else:
                gemm_group([dict(A=t, lda=F_, W=lw.Wt, bias=lw.bt, C=t2, ldc=F_, rows=E, nout=F_, K=F_, act=(0, F_), res=t,
                                 gate=w, pre_out=lt.pre_t if save else None), m0])
            t, t2 = t2, t
        else:
            gemm_group([m0])
        if not eq_fused:
            gemm(g1act, F_, lw.Wm1, lw.bm1, mm, 2 * F_, N, 2 * F_, F_)
            call("gn_eqff_update", ptr(mm), ptr(Xp), N, F_, D, ptr(h), ptr(X), _stream())
        if trace is not None:
            trace.append((h.clone(), X.clone(), t.clone()) if pw.emb_idx is None else
                         tuple(v.index_select(v.dim() - 1, pw.emb_idx) for v in (h, X, t)))
    if pw.emb_idx is not None:                     # embedded model: the real channels, in the model's own order
        h, X = h.index_select(1, pw.emb_idx), X.index_select(2, pw.emb_idx)
    return h, X, tape


def _norm_h(cfg: Config, pw: PackedWeights, lw: LayerWeights, h: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm on h at the GATA input (gotennet.py:397).  Embedded model: over the real channels (embed.py)."""
    N = h.shape[0]
    if pw.emb_idx is None:
        hn = torch.empty_like(h)
        call("gn_layernorm", ptr(h), ptr(lw.ln_w), ptr(lw.ln_b), 1e-5, N, cfg.F, ptr(hn), _stream())
        return hn
    hc = h.index_select(1, pw.emb_idx)
    yc = torch.empty_like(hc)
    call("gn_layernorm", ptr(hc), ptr(lw.ln_w), ptr(lw.ln_b), 1e-5, N, cfg.F_model, ptr(yc), _stream())
    return torch.zeros_like(h).index_copy_(1, pw.emb_idx, yc)


def _norm_X(cfg: Config, pw: PackedWeights, lw: LayerWeights, X: torch.Tensor) -> torch.Tensor:
    """TensorLayerNorm on X at the GATA input (gotennet.py:398, layers.py:1497-1563)."""
    N = X.shape[0]
    if pw.emb_idx is None:
        Xn = torch.empty_like(X)
        call("gn_tensor_norm", ptr(X), ptr(lw.tln_w), 1e-12, N, cfg.F, cfg.lmax, ptr(Xn), _stream())
        return Xn
    Xc = X.index_select(2, pw.emb_idx)
    Yc = torch.empty_like(Xc)
    call("gn_tensor_norm", ptr(Xc), ptr(lw.tln_w), 1e-12, N, cfg.F_model, cfg.lmax, ptr(Yc), _stream())
    return torch.zeros_like(X).index_copy_(2, pw.emb_idx, Yc)


def gata_input_norms(cfg: Config, lw: LayerWeights, h: torch.Tensor, X: torch.Tensor):
    """The optional input norms of a GATA layer (gotennet.py:397-398) on their own: -> (h or LN(h), X or TLN(X))."""
    N, F_ = h.shape[0], cfg.F
    if cfg.layernorm and N:
        hn = torch.empty_like(h)
        call("gn_layernorm", ptr(h), ptr(lw.ln_w), ptr(lw.ln_b), 1e-5, N, F_, ptr(hn), _stream())
        h = hn
    if cfg.steerable_norm and N:
        Xn = torch.empty_like(X)
        call("gn_tensor_norm", ptr(X), ptr(lw.tln_w), 1e-12, N, F_, cfg.lmax, ptr(Xn), _stream())
        X = Xn
    return h, X


def _gata_layer_impl(cfg: Config, lw: LayerWeights, g: "Graph", h: torch.Tensor, X: torch.Tensor, t: torch.Tensor):
    """ONE GATA layer (gotennet.py:366-450) on its own, inference only: what ``GATA.forward`` of the mirror module runs
    when a caller composes layers directly.  Same kernels as ``forward`` (which additionally fuses the neighbouring EQFF
    launches into the grouped GEMMs).  ``g`` carries the CSR view, rl and the cosine cutoff.  -> (h', X', t')."""
    F_, H, D, M, lmax, Fe = cfg.F, cfg.H, cfg.D, cfg.M, cfg.lmax, cfg.Fe
    N, E = g.N, g.E
    proj = _Proj(cfg)
    gemm, gemm_group = proj.gemm, proj.group
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=h.device)
    last = lw.Wt is None
    if cfg.layernorm:
        hn = new(N, F_)
        call("gn_layernorm", ptr(h), ptr(lw.ln_w), ptr(lw.ln_b), 1e-5, N, F_, ptr(hn), _stream())
        h = hn
    if cfg.steerable_norm:
        Xn = new(N, D, F_)
        call("gn_tensor_norm", ptr(X), ptr(lw.tln_w), 1e-12, N, F_, lmax, ptr(Xn), _stream())
        X = Xn
    lde = (1 + M) * F_
    nact, xs, vs = new(N, 4 * F_), new(N, M * F_), new(N, M * F_)
    eproj, attn = new(E, lde), new(E, H)
    h2, X2 = new(N, F_), new(N, D, F_)
    gemm_group([dict(A=t, lda=F_, W=lw.We, bias=lw.be, C=eproj, ldc=lde, rows=E, nout=lde, K=F_),
                dict(A=h, lda=F_, W=lw.Wn1, bias=lw.bn1, C=nact, ldc=4 * F_, rows=N, nout=4 * F_, K=F_,
                     act=(2 * F_, 4 * F_))])
    gemm_group([dict(A=nact, lda=4 * F_, W=lw.Ws2, bias=lw.bs2, C=xs, ldc=M * F_, rows=N, nout=M * F_, K=F_, a_off=2 * F_),
                dict(A=nact, lda=4 * F_, W=lw.Wv2, bias=lw.bv2, C=vs, ldc=M * F_, rows=N, nout=M * F_, K=F_, a_off=3 * F_)])
    message_stage(cfg, g, nact, xs, vs, eproj, attn, h, X, h2, X2)
    h, X = h2, X2
    if last:
        return h, X, t
    EQ, EK, w, t2 = new(N, D, Fe), new(N, D, Fe), new(E, Fe), new(E, F_)
    xprods = [dict(A=X, lda=F_, W=lw.Wvq, C=EQ, ldc=Fe, rows=N * D, nout=Fe, K=F_)]
    if cfg.htr_mode & 1:
        xprods.append(dict(A=X, lda=F_, W=lw.Wvk[0], C=EK, ldc=Fe, rows=N * D, nout=Fe, K=F_))
    else:
        off = 0
        for l in range(1, lmax + 1):
            cnt = 2 * l + 1
            xprods.append(dict(A=X, lda=F_, W=lw.Wvk[l - 1], C=EK, ldc=Fe, rows=N * cnt, nout=Fe, K=F_,
                               rowmap=(cnt, D, off)))
            off += cnt
    gemm_group(xprods)
    call("gn_htr_edge", ptr(EQ), ptr(EK), ptr(g.rl), ptr(g.rowptr), ptr(g.src), N, Fe, cfg.lmax_arg, cfg.htr_mode, None, ptr(w),
         _stream())
    if cfg.composed_update:
        _edge_update_composed(cfg, lw, t, w, t2, E, None)
    else:
        gemm(t, F_, lw.Wt, lw.bt, t2, F_, E, F_, F_, act=(0, F_), res=t, gate=w)
    return h, X, t2


def _eqff_layer_impl(cfg: Config, lw: LayerWeights, h: torch.Tensor, X: torch.Tensor):
    """ONE EQFF block (gotennet.py:716-748) on its own, inference only (``EQFF.forward`` of the mirror module).
    Returns NEW tensors (h', X')."""
    F_, D = cfg.F, cfg.D
    N = h.shape[0]
    gemm = _Proj(cfg).gemm
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=h.device)
    Xp, ctx, g1, mm = new(N, D, F_), new(N, 2 * F_), new(N, F_), new(N, 2 * F_)
    gemm(X, F_, lw.Wvu, None, Xp, F_, N * D, F_, F_)
    call("gn_eqff_context", ptr(h), ptr(Xp), float(cfg.eps), N, F_, D, ptr(ctx), _stream())
    gemm(ctx, 2 * F_, lw.Wm0, lw.bm0, g1, F_, N, F_, 2 * F_, act=(0, F_))
    gemm(g1, F_, lw.Wm1, lw.bm1, mm, 2 * F_, N, 2 * F_, F_)
    h, X = h.clone(), X.clone()
    call("gn_eqff_update", ptr(mm), ptr(Xp), N, F_, D, ptr(h), ptr(X), _stream())
    return h, X


def message_stage(cfg: Config, g: "Graph", nact, xs, vs, eproj, attn, h, X, h2, X2):
    """GATA message stage (gotennet.py:452-559, 613-640, 426-427): scores + segment softmax, message, aggregate,
    residual.  q | k are columns [0, 2F) of ``nact``; t_attn (pre-activation) columns [0, F) of ``eproj``, t_filter the
    rest."""
    F_, H, M = cfg.F, cfg.H, cfg.M
    lde = (1 + M) * F_
    call("gn_attn_softmax", ptr(nact), nact.data_ptr() + 4 * F_, 4 * F_, ptr(eproj), lde,
         ptr(g.rowptr), ptr(g.src), ptr(g.outdeg), g.N, F_, H, ptr(attn), cfg.act, _stream())
    call("gn_message_aggregate", ptr(xs), ptr(vs), M * F_, eproj.data_ptr() + 4 * F_, lde,
         ptr(attn), ptr(g.rl), ptr(g.cut), ptr(g.rowptr), ptr(g.src),
         ptr(h), ptr(X), ptr(h2), ptr(X2), g.N, F_, H, cfg.lmax_arg_msg, int(cfg.sep_dir), int(cfg.sep_tensor), _stream())


#: ``fuse_eqff = None`` (auto): fused up to this many atoms per call (measured crossover between 672 and 2688 atoms at F = 256)
EQFF_FUSED_MAX_ATOMS = 1024


def eqff_fused_ok(cfg: Config, n_atoms: Optional[int] = None) -> bool:
    """The EQFF chains run as one kernel each way (gn_eqff_fused_forward / _backward): F in {128, 256}, SiLU, a plane
    arithmetic; everything else keeps the launch sequence.  ``cfg.fuse_eqff`` None: decided by the system size."""
    mode = resolve_mode(cfg.gemm_mode)
    want = cfg.fuse_eqff if cfg.fuse_eqff is not None else (n_atoms is not None and n_atoms <= EQFF_FUSED_MAX_ATOMS)
    if not want or mode not in _PLANE_MODES:
        return False
    return bool(_lib.load().gn_eqff_fused_supported(cfg.F, cfg.act, 1 if mode == "split" else 2))


def _edge_update_composed(cfg: Config, lw: LayerWeights, t, w_raw, t2, E: int, pre_t):
    """t2 = t + gamma_t(t) * gamma_w(w) for the non-default variants (gotennet.py:236-291, 611):
    gamma_w = [LayerNorm "ln"] -> [SiLU "linwa"] -> W_edp "linw"/"linwa" [-> LayerNorm "postln"] -> [gate];
    gamma_t = Dense -> [LayerNorm edge_ln] -> SiLU -> Dense [-> SiLU unless "mlp"]  ("mlp"/"mlpa"), or the
    default SiLU(Dense).  Returns the intermediates the backward needs."""
    F_, Fe, Fm = cfg.F, cfg.Fe, cfg.Fm
    gemm = _Proj(cfg).gemm
    new = lambda width=F_: torch.empty((E, width), dtype=torch.float32, device=t.device)
    st = _stream()
    u = dict(w_raw=w_raw)
    x = w_raw
    if cfg.lin_w:
        a_in = x
        if cfg.lin_ln == 1:
            a_in = new(Fe)
            call("gn_layernorm", ptr(x), ptr(lw.w_ln_w), ptr(lw.w_ln_b), 1e-5, E, Fe, ptr(a_in), st)
        lin = new()
        gemm(a_in, Fe, lw.Wedp, lw.bedp, lin, F_, E, F_, Fe, pro=(1, 0, Fe) if cfg.lin_w == 2 else (0, 0, 0))
        x = lin
        if cfg.lin_ln == 2:
            x = new()
            call("gn_layernorm", ptr(lin), ptr(lw.w_ln_w), ptr(lw.w_ln_b), 1e-5, E, F_, ptr(x), st)
        u.update(a_in=a_in, lin=lin)
    u["pre_gate"] = x
    if cfg.gate_kind:
        wg = new()
        call("gn_gate", ptr(x), cfg.gate_kind, E * F_, ptr(wg), st)
    else:
        wg = x
    u["wg"] = wg
    act = (0, F_) if cfg.t_last_act == 3 else (0, 0)
    if lw.Wt0 is not None:
        hid = new(Fm)
        gemm(t, F_, lw.Wt0, lw.bt0, hid, Fm, E, Fm, F_)
        u_in = hid
        if lw.t_ln_w is not None:
            u_in = new(Fm)
            call("gn_layernorm", ptr(hid), ptr(lw.t_ln_w), ptr(lw.t_ln_b), 1e-5, E, Fm, ptr(u_in), st)
        gemm(u_in, Fm, lw.Wt, lw.bt, t2, F_, E, F_, Fm, act=act, res=t, gate=wg, pre_out=pre_t, pro=(1, 0, Fm))
        u.update(hid=hid, u_in=u_in)
    else:
        gemm(t, F_, lw.Wt, lw.bt, t2, F_, E, F_, F_, act=act, res=t, gate=wg, pre_out=pre_t)
    return u


def _edge_update_composed_backward(cfg: Config, lw: LayerWeights, lt, gt, gt_a, E: int):
    """Input-gradients of _edge_update_composed: writes gt_a = gt + (d/dt through gamma_t) and returns dL/dw [E,F]."""
    F_, Fe, Fm = cfg.F, cfg.Fe, cfg.Fm
    gemm = _Proj(cfg).gemm
    new = lambda width=F_: torch.empty((E, width), dtype=torch.float32, device=gt.device)
    st = _stream()
    u = lt.upd
    g_pre, g_wg = new(), new()
    call("gn_edge_gate_backward", ptr(gt), ptr(lt.pre_t), cfg.act if cfg.t_last_act else -1, ptr(u["wg"]), E * F_, ptr(g_pre), ptr(g_wg), st)
    if lw.Wt0 is not None:
        g_u = new(Fm)
        gemm(g_pre, F_, _T(lw, "Wt"), None, g_u, Fm, E, Fm, F_, dgate=u["u_in"])
        if lw.t_ln_w is not None:
            g_h = new(Fm)
            call("gn_layernorm_backward", ptr(u["hid"]), ptr(lw.t_ln_w), 1e-5, ptr(g_u), E, Fm, ptr(g_h), st)
            g_u = g_h
        gemm(g_u, Fm, _T(lw, "Wt0"), None, gt_a, F_, E, F_, Fm, res=gt)
    else:
        gemm(g_pre, F_, _T(lw, "Wt"), None, gt_a, F_, E, F_, F_, res=gt)
    gq = g_wg
    if cfg.gate_kind:
        g2 = new()
        call("gn_gate_backward", ptr(gq), ptr(u["pre_gate"]), cfg.gate_kind, E * F_, ptr(g2), st)
        gq = g2
    if cfg.lin_w:
        if cfg.lin_ln == 2:
            g2 = new()
            call("gn_layernorm_backward", ptr(u["lin"]), ptr(lw.w_ln_w), 1e-5, ptr(gq), E, F_, ptr(g2), st)
            gq = g2
        g3 = new(Fe)
        gemm(gq, F_, _T(lw, "Wedp"), None, g3, Fe, E, Fe, F_, dgate=u["a_in"] if cfg.lin_w == 2 else None)
        gq = g3
        if cfg.lin_ln == 1:
            g4 = new(Fe)
            call("gn_layernorm_backward", ptr(u["w_raw"]), ptr(lw.w_ln_w), 1e-5, ptr(gq), E, Fe, ptr(g4), st)
            gq = g4
    return gq


def _backward_impl(cfg: Config, pw: PackedWeights, z32: torch.Tensor, g: Graph, tape: Tape,
             gh: torch.Tensor, gX: Optional[torch.Tensor]):
    """Input-gradients of ``forward``: given dL/dh [N,F] and dL/dX [N,D,F] (or None = 0)
    returns (g_edge_vec [E,3], g_edge_diff [E]) in the CSR edge order of ``g``."""
    F_, R, H, D, M, lmax = cfg.F, cfg.R, cfg.H, cfg.D, cfg.M, cfg.lmax
    Fe = cfg.Fe
    N, E = g.N, g.E
    proj = _Proj(cfg)
    gemm, gemm_group = proj.gemm, proj.group
    f32 = dict(dtype=torch.float32, device=z32.device)
    new = lambda *shape: torch.empty(shape, **f32)
    check_backward_supported(cfg)
    colptr, perm = g.csc()
    lde = (1 + M) * F_
    eq_fused, eq_arith = eqff_fused_ok(cfg, N), (1 if proj.mode == "split" else 2)

    # every contributing kernel writes its own slice; the geometry backward sums them in a fixed order
    L = len(pw.layers)
    G = _lib.load().gn_message_backward_groups(cfg.lmax_arg_msg_bwd, int(cfg.sep_dir), int(cfg.sep_tensor), cfg.act)
    # W: partial slices per writing call (a slot wider than a wave writes one per 64-lane part: F / 256, Fe / 256 for HTR)
    Wm, Wh = max(1, F_ // 256), max(1, Fe // 256)
    n_htr = sum(lw.Wt is not None for lw in pw.layers)
    n_rl, n_cut = Wm * L + Wh * n_htr, Wm * (G * L + 1)
    g_rl_parts, g_cut_parts = new(n_rl, E, D), new(n_cut, E)
    ga_parts = new(G, E, H)                        # head sums of g_a: G partial slices (degree groups), or the merged kernel's one
    if cfg.aggr == 2:                              # aggr = "max": the per-message gradient workspace of the routing kernel
        ga_parts = new(E, 1 + D, F_)
    rl_slice = lambda q: g_rl_parts.data_ptr() + 4 * q * E * D
    cut_slice = lambda q: g_cut_parts.data_ptr() + 4 * q * E
    htr_slice = {}                                 # layer -> first g_rl slice of its HTR backward (after the message slices)
    for li_, lw_ in enumerate(pw.layers):
        if lw_.Wt is not None:
            htr_slice[li_] = Wm * L + Wh * len(htr_slice)
    if pw.emb_idx is not None:                     # embedded model: gradients arrive in the real layout
        z_ = torch.zeros((N, F_), **f32)
        gh = z_.index_copy_(1, pw.emb_idx, gh.contiguous())
        if gX is not None:
            gX = torch.zeros((N, D, F_), **f32).index_copy_(2, pw.emb_idx, gX.contiguous())
    gh = gh.contiguous()
    # dL/dX = None (an energy head reads h only): the un-fused EQFF backward takes a null pointer for it (no zero-filled
    # [N,D,F] tensor is made, written or read); the fused kernel wants the tensor
    if gX is None:
        gX = torch.zeros((N, D, F_), **f32) if eq_fused else None
    else:
        gX = gX.contiguous()
    gh_caller, gX_caller = gh, gX                  # read-only: never enter the work-buffer rotation below
    gt = None                                      # dL/dt of the layer output (None = 0)

    gm, gXp, g_g1, g_ctx = new(N, 2 * F_), new(N, D, F_), new(N, F_), new(N, 2 * F_)
    gh1, gX1, gX2, gh2, gh_qk = new(N, F_), new(N, D, F_), new(N, D, F_), new(N, F_), new(N, F_)
    gEQ, gEK = new(N, D, Fe), new(N, D, Fe)
    g_eproj, g_s = new(E, lde), new(E, H)
    g_nproj, g_x, g_v = new(N, 4 * F_), new(N, M * F_), new(N, M * F_)
    gt_a, gt_b, g_pre_t = new(E, F_), new(E, F_), new(E, F_)

    for li in reversed(range(len(pw.layers))):
        lw, lt = pw.layers[li], tape.layers[li]
        last = lw.Wt is None
        first = zero_X_in(cfg, li)
        # ---- EQFF backward (one kernel where covered; else its first half here); HTR backward kernels (independent of it)
        m1 = None
        if eq_fused:
            call("gn_eqff_fused_backward", ptr(gh), ptr(gX), ptr(lt.mm), ptr(lt.Xp), ptr(lt.ctx), ptr(lt.pre_g1),
                 ptr(split_weight(_T(lw, "Wm1"), proj.mode)), ptr(split_weight(_T(lw, "Wm0"), proj.mode)), N, F_, D,
                 ptr(gXp), ptr(gh1), eq_arith, _stream())
        else:
            call("gn_eqff_backward_a", ptr(gh), ptr(gX), ptr(lt.mm), ptr(lt.Xp), N, F_, D, ptr(gm), ptr(gXp), _stream())
            m1 = dict(A=gm, lda=2 * F_, W=_T(lw, "Wm1"), C=g_g1, ldc=F_, rows=N, nout=F_, K=2 * F_,
                      dgate=lt.pre_g1)             # * SiLU'(pre) in the epilogue
        if not last:
            if gt is None:
                raise RuntimeError("internal: missing edge gradient")
            if cfg.composed_update:                # gt_a = gt + gamma_t backward; g_w = gamma_w backward
                g_w = _edge_update_composed_backward(cfg, lw, lt, gt, gt_a, E)
                call("gn_htr_backward", ptr(g_w), None, None, None, ptr(lt.EQ), ptr(lt.EK), ptr(g.rl),
                     ptr(g.rowptr), ptr(g.src), ptr(g.tgt_by_src), ptr(colptr), ptr(perm), N, Fe, cfg.lmax_arg_bwd, cfg.htr_mode | 16,
                     ptr(gEQ), ptr(gEK), rl_slice(htr_slice[li]), None, cfg.act, _stream())
                gemm_group([m1])
            else:
                call("gn_htr_backward", ptr(gt), ptr(lt.pre_t), ptr(lt.w), ptr(lt.w_raw), ptr(lt.EQ), ptr(lt.EK),
                     ptr(g.rl), ptr(g.rowptr), ptr(g.src), ptr(g.tgt_by_src), ptr(colptr), ptr(perm), N, Fe, cfg.lmax_arg_bwd,
                     cfg.htr_mode, ptr(gEQ), ptr(gEK), rl_slice(htr_slice[li]), ptr(g_pre_t), cfg.act, _stream())
                # gt_a = gt + ((gt * w) * SiLU'(pre_t)) Wt; the atom-sized gamma_m product rides in its launch
                gemm_group([dict(A=g_pre_t, lda=F_, W=_T(lw, "Wt"), C=gt_a, ldc=F_, rows=E, nout=F_, K=F_, res=gt), m1])
            gt_in = gt_a
        else:
            gemm_group([m1])
            gt_in = gt                             # no edge update in this layer: t passes through unchanged
        # ---- EQFF backward, second half
        if not eq_fused:
            gemm(g_g1, F_, _T(lw, "Wm0"), None, g_ctx, 2 * F_, N, 2 * F_, F_)
            call("gn_eqff_backward_b", ptr(g_ctx), ptr(lt.ctx), ptr(lt.Xp), ptr(gh), N, F_, D, ptr(gXp), ptr(gh1), _stream())
        # ---- gX1 = gX + gXp W_vu (+ gEQ W_vq + gEK_l W_vk_l)
        joint = bool(cfg.htr_mode & 1)
        if last:
            gemm(gXp, F_, _T(lw, "Wvu"), None, gX1, F_, N * D, F_, F_, res=gX)
        elif Fe == F_:
            # one launch; per degree block: A = [gXp | gEQ | gEK] (K-segmented), W = [W_vu^T | W_vq^T | W_vk_l^T] along K
            probs, off = [], 0
            for l in range(1, lmax + 1):
                cnt = D if joint else 2 * l + 1
                wcat = lw.T.get(("Xcat", l))
                if wcat is None:
                    wcat = torch.cat([_T(lw, "Wvu"), _T(lw, "Wvq"), lw.Wvk[l - 1].t()], dim=1).contiguous()
                    lw.T[("Xcat", l)] = wcat
                probs.append(dict(A=gXp, A2=gEQ, A3=gEK, a_seg=F_, lda=F_, W=wcat, C=gX1, ldc=F_, rows=N * cnt,
                                  nout=F_, K=3 * F_, rowmap=(cnt, D, off), res=gX))
                off += cnt
                if joint:
                    break
            gemm_group(probs)
        else:                                      # evec_dim != F: three chained products
            gemm(gXp, F_, _T(lw, "Wvu"), None, gX1, F_, N * D, F_, F_, res=gX)
            gemm(gEQ, Fe, _T(lw, "Wvq"), None, gX1, F_, N * D, F_, Fe, res=gX1)
            off = 0
            for l in range(1, lmax + 1):
                cnt = D if joint else 2 * l + 1
                wkT = lw.T.get(("Wvk", l))
                if wkT is None:
                    wkT = lw.Wvk[l - 1].t().contiguous()
                    lw.T[("Wvk", l)] = wkT
                gemm(gEK, Fe, wkT, None, gX1, F_, N * cnt, F_, Fe, rowmap=(cnt, D, off), res=gX1)
                off += cnt
                if joint:
                    break
        # ---- message backward
        if first and G > 1:                        # one launch instead of G degree groups: one g_cut slice is written
            g_cut_parts[G * li + 1:G * li + G].zero_()       # (never together with wide slots: zero_X_in excludes them)
        call("gn_message_backward", ptr(lt.xs), ptr(lt.vs), M * F_, ptr(lt.eproj), lde, ptr(lt.attn),
             ptr(lt.nproj), 4 * F_, None if first else ptr(lt.X_in), ptr(g.rl), ptr(g.cut), ptr(g.outdeg),
             ptr(gh1), ptr(gX1), ptr(g.rowptr), ptr(g.src), ptr(g.tgt_by_src), ptr(colptr), ptr(perm),
             ptr(g_eproj), ptr(g_s), ptr(g_nproj), 4 * F_, ptr(g_x), ptr(g_v), None if first else ptr(gX2),
             rl_slice(Wm * li), cut_slice(Wm * G * li),
             None if (MSG_BWD_PAIR and (first or G == 1)) else ptr(ga_parts), E,
             N, F_, H, cfg.lmax_arg_msg_bwd, int(cfg.sep_dir), int(cfg.sep_tensor), cfg.act, _stream())
        # the edge-sized W_e^T product leaves 0.7 of its last tile round idle: the two K-heavy atom-sized products
        # (g_x W_s2, g_v W_v2; 60 us as a launch of their own) ride there; W_n1^T needs their output and follows alone
        if first:                                  # the tensor-gate columns of g_eproj were not written: K-prefix
            _, _, ke = _We_first(cfg, lw)
            _value_first(cfg, lw)                  # (Ws20 / Wv20 exist before the group below names them)
        gemm_group([dict(A=g_eproj, lda=lde, W=_T(lw, "We0" if first else "We"), C=gt_b, ldc=F_, rows=E, nout=F_,
                         K=ke if first else lde, res=gt_in),
                    dict(A=g_x, lda=M * F_, W=_T(lw, "Ws20" if first else "Ws2"), C=g_nproj, ldc=4 * F_, rows=N, nout=F_,
                         K=_value_first(cfg, lw) if first else M * F_, c_off=2 * F_, dgate=lt.nproj, g_off=2 * F_),
                    dict(A=g_v, lda=M * F_, W=_T(lw, "Wv20" if first else "Wv2"), C=g_nproj, ldc=4 * F_, rows=N, nout=F_,
                         K=_value_first(cfg, lw) if first else M * F_,
                         c_off=3 * F_, dgate=lt.nproj, g_off=3 * F_),
                    # the q | k half of the W_n1^T product needs only the message backward's g_q | g_k: it rides here
                    # too and halves the K of the product that has to wait for the two riders above (33 -> 20 us)
                    dict(A=g_nproj, lda=4 * F_, W=_Tqk(lw), C=gh_qk, ldc=F_, rows=N, nout=F_, K=2 * F_, res=gh1)])
        gemm(g_nproj, 4 * F_, _Tsv(lw), None, gh2, F_, N, F_, 2 * F_, res=gh_qk, a_off=2 * F_)
        gh, gh2 = gh2, gh
        gX, gX2 = gX2, gX
        if gh2 is gh_caller:
            gh2 = new(N, F_)
        if gX2 is gX_caller:
            gX2 = new(N, D, F_)
        gt, gt_b = gt_b, (gt if gt is not None else new(E, F_))
        # ---- optional input norms (gotennet.py:397-398): back to the un-normalised h / X
        if cfg.layernorm:
            if pw.emb_idx is None:
                call("gn_layernorm_backward", ptr(lt.h_raw), ptr(lw.ln_w), 1e-5, ptr(gh), N, F_, ptr(gh2), _stream())
            else:                                  # statistics over the real channels: compact -> kernel -> padded layout
                gc = torch.empty((N, cfg.F_model), **f32)
                xc, gyc = lt.h_raw.index_select(1, pw.emb_idx), gh.index_select(1, pw.emb_idx)   # (named: alive until the launch)
                call("gn_layernorm_backward", ptr(xc), ptr(lw.ln_w), 1e-5, ptr(gyc), N, cfg.F_model, ptr(gc), _stream())
                gh2.zero_().index_copy_(1, pw.emb_idx, gc)
            gh, gh2 = gh2, gh
        if cfg.steerable_norm:
            if pw.emb_idx is None:
                call("gn_tensor_norm_backward", ptr(lt.X_raw), ptr(lw.tln_w), ptr(gX), 1e-12, N, F_, lmax, ptr(gX2), _stream())
            else:
                gc = torch.empty((N, D, cfg.F_model), **f32)
                xc, gyc = lt.X_raw.index_select(2, pw.emb_idx), gX.index_select(2, pw.emb_idx)
                call("gn_tensor_norm_backward", ptr(xc), ptr(lw.tln_w), ptr(gyc), 1e-12, N, cfg.F_model, lmax, ptr(gc), _stream())
                gX2.zero_().index_copy_(2, pw.emb_idx, gc)
            gX, gX2 = gX2, gX

    # ---- init backward (layers.py:1658-1714) ------------------------------------------
    g_feat = new(E, 2 * F_)
    call("gn_edge_init_backward", ptr(gt), ptr(tape.h0), ptr(tape.feat), 2 * F_, ptr(g.rowptr), ptr(g.src),
         ptr(colptr), ptr(perm), N, F_, ptr(g_feat), ptr(gh), _stream())
    Fc = cfg.Fc or F_
    gy = new(N, Fc)
    gemm(gh, F_, _T(pw, "Wb"), None, gy, Fc, N, Fc, F_)
    gy1 = new(N, Fc)
    call("gn_layernorm_silu_backward", ptr(tape.y_pre), ptr(pw.ln_w), ptr(pw.ln_b), 1e-5, ptr(gy), N, Fc, ptr(gy1), cfg.act, _stream())
    gemm(gy1, Fc, _T(pw, "Wa"), None, g_ctx, 2 * F_, N, 2 * F_, Fc)
    call("gn_node_init_backward", ptr(g_ctx), ptr(z32), ptr(tape.feat), 2 * F_, ptr(g.cut), ptr(pw.A_nbr),
         ptr(g.rowptr), ptr(g.src), N, F_, ptr(g_feat), cut_slice(Wm * G * L), _stream())
    g_phi = new(E, R)
    gemm(g_feat, 2 * F_, _T(pw, "Winit"), None, g_phi, R, E, R, 2 * F_)
    g_vec, g_diff = new(E, 3), new(E)
    call("gn_edge_geometry_backward", ptr(g.edge_vec), ptr(g.edge_diff), ptr(g.src), ptr(g.dst), E, lmax, R,
         cfg.basis, ptr(pw.rb0), ptr(pw.rb1), float(cfg.cutoff), ptr(g_rl_parts), n_rl, ptr(g_cut_parts), n_cut,
         ptr(g_phi), ptr(g_vec), ptr(g_diff), _stream())
    return g_vec, g_diff


def pos_gradient(g: Graph, g_vec: torch.Tensor, g_diff: torch.Tensor, sign: float = 1.0) -> torch.Tensor:
    """sign * dL/dpos for edge_vec = pos[j] - pos[i], edge_diff = |edge_vec| (Distance, layers.py:1593-1600)."""
    colptr, perm = g.csc()
    out = torch.empty((g.N, 3), dtype=torch.float32, device=g_vec.device)
    call("gn_pos_scatter", ptr(g_vec), ptr(g_diff), ptr(g.edge_vec), ptr(g.rowptr), ptr(colptr), ptr(perm),
         g.N, float(sign), ptr(out), _stream())
    return out


forward, backward, gata_layer, eqff_layer = _forward_impl, _backward_impl, _gata_layer_impl, _eqff_layer_impl
