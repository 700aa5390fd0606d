"""Fused energy + force step: representation forward (with tape) -> Atomwise head ->
head gradient -> hand-written backward -> force scatter, all through the C ABI with no
autograd graph and no host synchronisation (hipGraph-capturable).  This is the path
bench.py times; the autograd.Function wrappers in gotennet.py expose the same kernels to
reference-style callers (GotenModel / torch.autograd.grad)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import engine
from .gotennet import GotenNet
from .outputs import Atomwise, molecule_ptr


class EnergyForces:
    """``check_edges`` (default on): validate the caller's edge list on the device (index range, target-major order;
    one host sync per call) and stable-sort it by target when needed -- energies and forces are per molecule / per atom,
    so the order of the edge list never shows in the result; the ``batch`` vector must be non-decreasing (molecules
    contiguous) and is checked too.  Callers that pass radius-graph output
    (``gotennet_amd.graph.distance``: target-major by construction) switch it off and stay sync-free."""

    def __init__(self, representation: GotenNet, head: Atomwise, check_edges: bool = True, cache_topology: bool = True,
                 replay: bool = False, replay_after: int = 2):
        self.rep, self.head, self.check_edges = representation, head, check_edges
        #: ``replay``: after ``replay_after`` consecutive energy+force calls on the SAME topology (a ``cache_topology``
        #: hit: same edge_index tensor, same sizes, same packed weights and configuration) the step -- geometry kernel,
        #: forward, head, backward, force scatter: every launch of the eager step, nothing skipped -- is recorded into
        #: ONE hipGraph (torch.cuda.CUDAGraph) and later calls replay it on fresh copies of ``edge_diff`` / ``edge_vec`` /
        #: ``z`` / ``mol_ptr``: the ~190 launches of a step are dispatched by the GPU's command processor instead of the
        #: host (-2.8 % on the C2 batch, 2.3 -> 1.9 ms for one molecule), bit-identical to the eager step.  The graph's
        #: private memory pool (the step's work buffers, ~4.5 GB at C2) stays allocated until the topology changes or
        #: ``clear_cache()``.  Returned tensors are copies.  Off by default (a library should not capture behind the
        #: caller's back); an MD driver or a benchmark on a fixed neighbour list switches it on.
        self.replay, self.replay_after = bool(replay), max(1, int(replay_after))
        self._hits, self._graph_state = 0, None
        #: ``cache_topology``: a repeated call with the SAME ``edge_index`` tensor (same storage, shape and PyTorch
        #: version counter -- an MD loop or a benchmark on a fixed neighbour list) reuses the CSR / CSC index arrays,
        #: the validation verdict and the stable-sort permutation of the previous call: no sort / scan / fill launch and
        #: no host read on such a step, only the geometry kernel runs on the new ``edge_diff`` / ``edge_vec``.  The cache
        #: holds a reference to the tensor (its memory cannot be recycled under the key) and the E-sized index / geometry
        #: buffers of its Graph until the next different edge list or ``clear_cache()``; writes that bypass PyTorch's
        #: version counter (raw pointers, ``.data``) are not seen: pass ``cache_topology=False`` for such callers.
        #: Inference tensors (made under ``torch.inference_mode()``) have no version counter and are never cached.
        self.cache_topology = cache_topology
        self._topo = None
        self._mol_ptr = None                         # (key, batch tensor, offsets) of the last call without mol_ptr

    def clear_cache(self):
        """Drop the cached topology (it keeps the last ``edge_index`` tensor and E-sized index / geometry buffers alive)
        and the recorded hipGraph with its memory pool."""
        self._topo = self._mol_ptr = None
        self._hits, self._graph_state = 0, None

    def _graph(self, cfg, pw, N, edge_index, edge_diff, edge_vec, need_csc):
        key = None
        if self.cache_topology and not edge_index.is_inference():
            # (inference tensors -- a neighbour list built under torch.inference_mode() -- carry no version counter:
            #  reading ``_version`` raises, and a write to one cannot be seen, so they are never cached)
            key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()), N,
                   id(pw), cfg.lmax, cfg.R, cfg.basis, bool(cfg.scale_edge), bool(self.check_edges))
        hit = key is not None and self._topo is not None and self._topo[0] == key and self._topo[1] is edge_index
        self._hits = self._hits + 1 if hit else 0
        if not hit:
            self._graph_state = None
        if hit:
            _, _, g, order = self._topo
            g.cfg = cfg                              # cutoff / eps may have changed under the same packed weights
            if order is not None:
                edge_diff, edge_vec = edge_diff[order], edge_vec[order]
            g.set_geometry(edge_diff.contiguous(), edge_vec.contiguous())
        else:
            ei, order = edge_index.contiguous(), None
            if self.check_edges:
                ei, edge_diff, edge_vec, order = engine.sorted_edges(ei, edge_diff, edge_vec, N)
            g = engine.Graph(cfg, pw, N, ei, edge_diff.contiguous(), edge_vec.contiguous())
            self._topo = (key, edge_index, g, order) if key is not None else None
        if need_csc:
            g.csc()
        return g

    @torch.no_grad()
    def __call__(self, z: torch.Tensor, edge_index: torch.Tensor, edge_diff: torch.Tensor, edge_vec: torch.Tensor,
                 batch: torch.Tensor, n_mol: int, mol_ptr: Optional[torch.Tensor] = None,
                 forces: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """-> (energy [n_mol,1], forces [N,3])."""
        rep = self.rep
        cfg, pw = rep.config(), rep.packed_weights()
        N = z.shape[0]
        if mol_ptr is None:
            # (bincount + cumsum + two fills per call otherwise: kept per batch tensor, like the topology)
            mkey = None if (batch.is_inference() or not self.cache_topology) else (batch.data_ptr(), batch._version, tuple(batch.shape), n_mol)
            if mkey is not None and self._mol_ptr is not None and self._mol_ptr[0] == mkey and self._mol_ptr[1] is batch:
                mol_ptr = self._mol_ptr[2]
            else:
                if self.check_edges and N > 1 and bool((batch[1:] < batch[:-1]).any()):
                    # (a host read, like the edge validation; not repeated for a cached batch vector)
                    raise ValueError("batch must be non-decreasing (each molecule's atoms contiguous)")
                mol_ptr = molecule_ptr(batch, n_mol)
                self._mol_ptr = (mkey, batch, mol_ptr) if mkey is not None else None
        gs = self._graph_state
        if self.replay and gs is not None and forces and self._replay_ok(gs, cfg, pw, N, edge_index, n_mol):
            return self._replay(gs, z, edge_diff, edge_vec, mol_ptr)
        z32 = z.to(torch.int32)
        g = self._graph(cfg, pw, N, edge_index, edge_diff, edge_vec, forces)
        if (self.replay and forces and self._hits >= self.replay_after and self._topo is not None
                and not torch.cuda.is_current_stream_capturing()):
            gs = self._capture(cfg, pw, g, z32, edge_index, n_mol, mol_ptr)
            return self._replay(gs, z, edge_diff, edge_vec, mol_ptr)
        return self._step(cfg, pw, g, z32, mol_ptr, n_mol, forces)

    def _step(self, cfg, pw, g, z32, mol_ptr, n_mol, forces):
        h, X, tape = engine.forward(cfg, pw, z32, g, save=forces)
        e, y, pre1 = self.head.energy_raw(h, z32, mol_ptr, n_mol, mode=cfg.gemm_mode)
        if not forces:
            return e, None
        gh = self.head.grad_h_raw(pre1, cfg.F_model or cfg.F, mode=cfg.gemm_mode)
        g_vec, g_diff = engine.backward(cfg, pw, z32, g, tape, gh, None)
        return e, engine.pos_gradient(g, g_vec, g_diff, sign=-1.0)

    # ---- hipGraph replay of the eager step on a cached topology --------------------------------------------------
    def _replay_ok(self, gs, cfg, pw, N, edge_index, n_mol) -> bool:
        # (the head's scalars -- last bias, scale / shift, AtomwiseV3's molecule shift -- are baked into the captured kernel
        #  arguments and its transposed weights are captured pointers: its cache key is part of the replay key)
        ok = (self._topo is not None and self._topo[1] is edge_index and not edge_index.is_inference()
              and gs["key"] == (self._topo[0], cfg, id(pw), n_mol, self._head_key()) and self._topo[0][1] == edge_index._version)
        if not ok:
            self._graph_state = None
        return ok

    def _head_key(self):
        """Everything of the head a recorded step depends on: its packed-cache key (parameter addresses / versions,
        incl. the ScaleShift buffers) and the host scalars that become kernel arguments."""
        packed = self.head._packed()
        c = packed[1] if isinstance(packed, tuple) else packed
        return (id(self.head), c["key"], c.get("scale"), c.get("shift"), c.get("b2"), c.get("mol_shift"),
                tuple(c.get("scales") or ()), tuple(c.get("shifts") or ()), tuple(c.get("b2s") or ()))

    def _capture(self, cfg, pw, g, z32, edge_index, n_mol, mol_ptr):
        dev = z32.device
        order = self._topo[3]
        st = dict(key=(self._topo[0], cfg, id(pw), n_mol, self._head_key()), g=g, order=order, z32=z32.clone(), mol_ptr=mol_ptr.clone(),
                  ed=torch.empty(g.E, dtype=torch.float32, device=dev), ev=torch.empty((g.E, 3), dtype=torch.float32, device=dev))
        st["ed"].copy_(g.edge_diff)
        st["ev"].copy_(g.edge_vec)

        def body():
            g.set_geometry(st["ed"], st["ev"])
            return self._step(cfg, pw, g, st["z32"], st["mol_ptr"], n_mol, True)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                  # one run off the capture: weight planes / transposes are built here
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        st["graph"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(st["graph"]):
            st["e"], st["f"] = body()
        self._graph_state = st
        return st

    def _replay(self, st, z, edge_diff, edge_vec, mol_ptr):
        order = st["order"]
        st["ed"].copy_(edge_diff if order is None else edge_diff[order])
        st["ev"].copy_(edge_vec if order is None else edge_vec[order])
        st["z32"].copy_(z)
        st["mol_ptr"].copy_(mol_ptr)
        st["graph"].replay()
        return st["e"].clone(), st["f"].clone()


class InFlight:
    """Several independent energy+force evaluations in flight at once: ``lanes`` ``EnergyForces`` objects, each on its own
    HIP stream, fed round-robin -- call k runs on lane k % lanes while call k - 1 is still executing.

    Why it pays on MI355X: the projection launches of a step run at the socket power cap, the
    gather / scatter launches between them do not come near it, and every launch of ONE step depends on the one before.  Two
    steps on two streams let the command processor run one step's memory-bound kernels beside the other's matrix
    kernels: 7.46 -> 6.73 ms per 128-molecule batch (lmax 2), 13.77 -> 12.56 (lmax 4), measured with hipGraph replays.  The
    price is latency (a batch takes as long as before, or longer) and a second set of work buffers.

        lanes = InFlight(rep, head, lanes=2, check_edges=False)
        for batch in stream_of_batches:
            results.append(lanes(z, edge_index, edge_diff, edge_vec, batch, n_mol))     # returns at once
        lanes.wait()                      # the CURRENT stream now waits for every lane: results are safe to read

    Inputs must be ready on the current stream when a call is made (each lane waits for the current stream first).
    Every lane keeps its own topology cache.

    Memory lifetime across the streams (PyTorch's caching allocator hands a freed block back to the stream it was
    allocated on, without waiting for OTHER streams that still read it):
      * every tensor argument of a call is ``record_stream``-ed on the lane's stream, so the caller may drop or overwrite-by-
        reallocation its inputs right after the call returns (the normal data-loader loop);
      * the tensors a call returns were allocated on the lane's stream; the object keeps a reference to them until the next
        ``wait()``, which marks them as used on the stream that waits -- read results only after ``wait()``, on that stream;
      * a weight update (a stale pack) makes the call wait for ALL lanes before the old pack's operands are freed."""

    def __init__(self, representation: GotenNet, head: Atomwise, lanes: int = 2, **kw):
        dev = next(representation.parameters()).device
        self.lanes = [EnergyForces(representation, head, **kw) for _ in range(max(1, int(lanes)))]
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.lanes]
        self._k, self._seen = 0, set()
        self._pending = []                           # tensors returned since the last wait()
        self._pack = None                            # the weight pack the lanes were last fed with (kept alive across a rebuild)

    @staticmethod
    def _tensors(objs):
        for o in objs:
            if isinstance(o, torch.Tensor):
                yield o
            elif isinstance(o, (tuple, list)):
                yield from InFlight._tensors(o)

    def __call__(self, *args, _then=None, **kw):
        """``_then(energy, forces)`` (optional) runs inside the lane's stream context right after the step is enqueued -- e.g. the
        per-step collective of a multi-GPU run."""
        k = self._k % len(self.lanes)
        self._k += 1
        st = self.streams[k]
        rep = self.lanes[k].rep
        pw = rep.packed_weights()                    # (a stale pack is rebuilt HERE, on the caller's stream, before the lane waits for it)
        if pw is not self._pack:
            # a weight update: the old pack's operands (cat'ed weights, fp16 planes, transposes) may still be read by kernels
            # queued on other lanes.  This object holds the OLD pack until every lane has drained into the current stream, so
            # its blocks are not handed to the rebuild above or to anything after it while they are in use
            self.wait()
            self._pack = pw
        pack_id = id(pw)
        st.wait_stream(torch.cuda.current_stream(st.device))
        for t in self._tensors(list(args) + list(kw.values())):
            if t.is_cuda:
                t.record_stream(st)                  # read on the lane's stream: not reusable by the caller's stream before that
        # The first call of a kind (with / without forces) builds lazily cached operands that ALL lanes share -- packed
        # weights, their fp16 planes, the transposes of the backward -- on THIS lane's stream: it runs alone, fenced against
        # the other lanes on both sides.  Later calls find the caches filled and overlap freely.
        kind = (bool(kw.get("forces", True)), pack_id)       # (a weight update makes a new pack: cold again)
        cold = kind not in self._seen
        if cold:
            for other in self.streams:
                if other is not st:
                    st.wait_stream(other)
        with torch.cuda.stream(st):
            out = self.lanes[k](*args, **kw)
            if _then is not None:
                _then(*out)
        if cold:
            self._seen.add(kind)
            for other in self.streams:
                if other is not st:
                    other.wait_stream(st)
        self._pending.extend(self._tensors(out))
        return out

    @property
    def next_lane(self) -> int:
        return self._k % len(self.lanes)

    def wait(self):
        """The CURRENT stream waits for every lane; the tensors returned since the last ``wait()`` are marked as used on it."""
        cur = torch.cuda.current_stream(self.streams[0].device)
        for st in self.streams:
            cur.wait_stream(st)
        for t in self._pending:
            if t.is_cuda:
                t.record_stream(cur)
        self._pending = []

    def clear_cache(self):
        for ef in self.lanes:
            ef.clear_cache()


class CapturedStep:
    """Energy + forces for a FIXED edge list (static topology: an MD trajectory of molecules whose neighbour lists do
    not change, e.g. any molecule smaller than the cutoff) as ONE hipGraph replay per step.

    A step of the fused path is ~190 kernel launches; for one 21-atom molecule the eager path is launch-bound
    (2.7 ms on MI355X) while the kernels themselves need a fraction of that.  ``CapturedStep`` builds the CSR / CSC
    topology once, records edge vectors -> geometry -> forward -> head -> backward -> force scatter into a
    ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) and replays it for every new set of positions.  Same kernels, same
    order, same buffers: results are bit-identical to ``EnergyForces`` on the same edge list.

        step = CapturedStep(EnergyForces(rep, head), z, edge_index, batch, n_mol)
        energy, forces = step(pos)          # views of static output buffers: copy them to keep them
    """

    def __init__(self, ef: EnergyForces, z: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor, n_mol: int,
                 warmup: int = 2):
        rep, head = ef.rep, ef.head
        self.cfg, self.pw = rep.config(), rep.packed_weights()
        dev = z.device
        self.z32 = z.to(torch.int32)
        self.N, self.n_mol = z.shape[0], n_mol
        self.head = head
        self.mol_ptr = molecule_ptr(batch, n_mol)
        self.pos = torch.zeros((self.N, 3), dtype=torch.float32, device=dev)      # static input buffer
        edge_index = edge_index.contiguous()
        if ef.check_edges:                           # once, at construction: the topology is static
            bits = engine.validate_edges(edge_index, self.N)
            if bits & 2:
                raise ValueError(f"edge_index holds indices outside [0, {self.N})")
            if bits & 1:
                edge_index = edge_index[:, torch.sort(edge_index[1], stable=True).indices].contiguous()
        self.g = engine.Graph(self.cfg, self.pw, self.N, edge_index)
        self.g.csc()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():                             # warm-up off the capture
            for _ in range(max(1, warmup)):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.energy, self.forces = self._body()

    def _body(self):
        cfg, pw, g = self.cfg, self.pw, self.g
        g.set_positions(self.pos)
        h, X, tape = engine.forward(cfg, pw, self.z32, g, save=True)
        e, y, pre1 = self.head.energy_raw(h, self.z32, self.mol_ptr, self.n_mol, mode=cfg.gemm_mode)
        gh = self.head.grad_h_raw(pre1, cfg.F_model or cfg.F, mode=cfg.gemm_mode)
        g_vec, g_diff = engine.backward(cfg, pw, self.z32, g, tape, gh, None)
        return e, engine.pos_gradient(g, g_vec, g_diff, sign=-1.0)

    @torch.no_grad()
    def __call__(self, pos: torch.Tensor):
        self.pos.copy_(pos)
        self.graph.replay()
        return self.energy, self.forces
