"""CPU: host-side logic and the C-ABI surface (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gotennet_amd import _lib
    from gotennet_amd.build import build_library
    build_library()
    hdr = open(os.path.join(ROOT, "include", "gotennet_hip.h")).read()
    declared = set(re.findall(r"^(?:int|long) (gn_\w+)\(", hdr, flags=re.M))
    assert declared, "header parse failed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    arch = ctypes.c_char_p()
    assert _lib.load().gn_abi_version(ctypes.byref(arch)) == _lib.ABI_VERSION
    assert arch.value == b"gfx950"


def test_state_dict_layout_matches_golden_checkpoint():
    """Reference weights load with strict=True (key names incl. the duplicated MLP keys)."""
    import gotennet_amd
    from tests.golden_util import case_names, load_case
    for name in case_names():
        cfg, sd, _, _ = load_case(name)
        net = gotennet_amd.GotenNet(
            n_atom_basis=cfg["n_atom_basis"], n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"],
            cutoff_fn=gotennet_amd.CosineCutoff(cfg["cutoff"]), max_z=cfg["max_z"], num_heads=cfg["num_heads"],
            scale_edge=cfg["scale_edge"], lmax=cfg["lmax"], sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"],
            sep_htr=cfg.get("sep_htr", True), radial_basis=cfg.get("radial_basis", "expnorm"),
            edge_updates=cfg.get("edge_updates", True), layernorm=cfg.get("layernorm", ""),
            steerable_norm=cfg.get("steerable_norm", ""), edge_ln=cfg.get("edge_ln", ""),
            activation=cfg.get("activation", "silu"), evec_dim=cfg.get("evec_dim"), emlp_dim=cfg.get("emlp_dim"))
        assert sorted(net.state_dict().keys()) == sorted(sd.keys()), name
        net.load_state_dict(sd, strict=True)
        assert net.hidden_dim == cfg["n_atom_basis"] and net.cutoff == cfg["cutoff"]


def test_load_from_lightning_style_checkpoint(tmp_path):
    """GotenNet.load_from_checkpoint (reference gotennet.py:904-946): Lightning layout with the ``representation.``
    prefix, head weights to skip, hyper-parameters incl. a cutoff_fn mapping and the Hydra target key."""
    import gotennet_amd
    from tests.golden_util import load_case
    cfg, sd, head, _ = load_case("opt_mlp_linwa_ln_gated")
    hp = dict(n_atom_basis=cfg["n_atom_basis"], n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"],
              cutoff_fn={"cutoff": cfg["cutoff"]}, max_z=cfg["max_z"], num_heads=cfg["num_heads"],
              scale_edge=cfg["scale_edge"], lmax=cfg["lmax"], sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"],
              edge_updates=cfg["edge_updates"], edge_ln=cfg["edge_ln"], activation=cfg["activation"],
              __target__="gotennet.models.representation.gotennet.GotenNetWrapper")
    state = {"representation." + k: v for k, v in sd.items()}
    state.update({"output_modules.0." + k: v for k, v in head.items()})
    path = tmp_path / "model.ckpt"
    torch.save({"hyper_parameters": {"representation": hp}, "state_dict": state}, path)
    net = gotennet_amd.GotenNet.load_from_checkpoint(str(path))
    got = net.state_dict()
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert net.cutoff == cfg["cutoff"] and net.gata_list[0].composed_update
    with pytest.raises(FileNotFoundError):
        gotennet_amd.GotenNet.load_from_checkpoint(str(tmp_path / "missing.ckpt"))


def test_product_path_fails_loudly_on_cpu():
    import gotennet_amd
    from gotennet_amd._lib import GotenNetHipError
    net = gotennet_amd.GotenNet(n_atom_basis=32, n_interactions=1, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0))
    ei = torch.zeros((2, 1), dtype=torch.long)
    with pytest.raises(GotenNetHipError):
        net(torch.ones(1, dtype=torch.long), ei, torch.zeros(1), torch.zeros(1, 3))


def test_unsupported_flags_raise_before_launch():
    import gotennet_amd
    cut = gotennet_amd.CosineCutoff(5.0)
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(cutoff_fn=cut, activation="hardswish")      # not one of the GN_ACT_* kinds
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(cutoff_fn=cut, activation=torch.nn.ELU(alpha=0.5))
    for act in ("relu", "tanh", "softplus", "Swish", "leaky_relu", torch.nn.functional.gelu, torch.nn.Mish()):
        assert gotennet_amd.GotenNet(cutoff_fn=cut, n_atom_basis=32, n_interactions=1, n_rbf=8, activation=act).act_kind >= 0
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(cutoff_fn=cut, lmax=9)          # the reference's TensorInit stops at l = 8
    assert gotennet_amd.GotenNet(cutoff_fn=cut, n_atom_basis=32, n_interactions=1, n_rbf=8, lmax=8).config().D == 80
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(cutoff_fn=cut, edge_ln="batch")
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(cutoff_fn=cut, evec_dim=24, edge_updates="linw")
    for aggr, kind in (("add", 0), ("mean", 1), ("max", 2)):  # gotennet.py:84,638: the PyG reduce of GATA.aggregate
        assert gotennet_amd.GotenNet(cutoff_fn=cut, n_atom_basis=32, n_interactions=1, n_rbf=8, aggr=aggr).config().aggr == kind
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(cutoff_fn=cut, aggr="min")
    with pytest.raises(ValueError):
        gotennet_amd.GotenNet(cutoff_fn=cut, edge_updates="bogus")
    with pytest.raises(ValueError):
        gotennet_amd.GotenNet(cutoff_fn=cut, radial_basis="nope")


def test_no_oracle_import_in_product():
    """The product package must never import oracle/ (or the reference)."""
    pkg = os.path.join(ROOT, "gotennet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
            assert "/root/reference" not in src, fn


def test_packed_weights_layout():
    """Projections that share an input are concatenated in the order the kernels index them."""
    import gotennet_amd
    torch.manual_seed(0)
    F, M = 32, 5
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=2, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                lmax=2, sep_dir=True, sep_tensor=True)
    pw = net.packed_weights()
    g0 = net.gata_list[0]
    assert net.config().M == M and net.config().D == 8
    assert torch.equal(pw.layers[0].Wn1[:F], g0.W_q.weight) and torch.equal(pw.layers[0].Wn1[F:2 * F], g0.W_k.weight)
    assert torch.equal(pw.layers[0].Wn1[2 * F:3 * F], g0.gamma_s[0].weight)
    assert torch.equal(pw.layers[0].Wn1[3 * F:], g0.gamma_v[0].weight)
    assert torch.equal(pw.layers[0].We[:F], g0.W_re.weight) and torch.equal(pw.layers[0].We[F:], g0.W_rs.weight)
    assert pw.layers[0].We.shape == ((1 + M) * F, F)
    assert pw.layers[0].Wt is not None and pw.layers[1].Wt is None          # last layer has no HTR
    assert torch.equal(pw.Winit[:F], net.node_init.W_ndp.dense_layers[0].weight)
    assert torch.equal(pw.Winit[F:], net.edge_init.W_erp.weight)
    assert net.packed_weights() is pw                                       # cached ...
    with torch.no_grad():
        g0.W_q.weight.add_(1.0)
    pw2 = net.packed_weights()                                              # ... until a parameter changes
    assert pw2 is not pw and torch.equal(pw2.layers[0].Wn1[:F], g0.W_q.weight)


def test_synthetic_shards_are_consistent():
    """Rank r's shard is the same set of molecules whatever the world size (bench.py sharding)."""
    from gotennet_amd import synthetic
    from gotennet_amd.parallel import shard_range
    pos, batch, z = synthetic.make_batch("rmd17_aspirin", 8, seed=0)
    for world in (2, 4):
        parts = []
        for r in range(world):
            first, cnt = shard_range(r, world, 8)
            p, b, zz = synthetic.make_batch("rmd17_aspirin", cnt, seed=0, first_molecule=first)
            parts.append((p, zz))
        assert torch.equal(torch.cat([p for p, _ in parts]), pos)
        assert torch.equal(torch.cat([q for _, q in parts]), z)
    assert pos.shape == (8 * 21, 3) and int(batch.max()) == 7


def test_bench_line_fits_driver_tail():
    """The driver parses the LAST stdout line out of an 8 KB tail (round 3's 21 KB line was not parsed).  Build the
    compact line from a canned full record (the committed round-3 run, the largest record the bench ever produced)
    and hold it to the budget and the contract keys."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    assert len(json.dumps(full)) > 8000                      # the canned record is the oversized one
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_BUDGET < 8000
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_gather_scatter", "roofline_htr_edge",
              "roofline_message_backward", "cpu_baseline", "also"):
        assert k in out, k
    assert "workload" in out["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
        assert k in out["roofline_gather_scatter"], k
    assert not any(isinstance(v, dict) and "stages" in v for v in out.values())     # one level, no nested copies
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    # the side workloads keep their four family fractions
    for wl in ("lmax4", "md22_ac_ala3_b64", "md22_nanotube_b8_lmax3"):
        for k in ("value", "ms_per_step", "gather_frac", "htr_frac", "msg_bwd_frac", "gemm_frac"):
            assert k in out["also"][wl], (wl, k)
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    # round 6: the latency figure and the north-star target shape's record are TOP-LEVEL keys of the line
    full6 = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_full.json")))
    full6.update(value_one_at_a_time=19341.0, ms_per_batch_one_at_a_time=6.618, lanes_consistent=True,
                 roofline_target={"config": "rmd17_aspirin batch=128/GPU lmax=4", "gather_frac": 0.42, "gather_frac_general": 0.43,
                                  "htr_frac": 0.31, "msg_bwd_frac": 0.44, "htr_bwd_frac": 0.23, "ms_per_step": 12.2, "value": 10500.0})
    line6 = bench.compact_line(full6)
    out6 = json.loads(line6)
    assert len(line6) < bench.LINE_BUDGET - 300 and "also" in out6 and "lmax4" not in out6["also"]    # (headroom: the line never sheds `also`)
    assert out6["value_one_at_a_time"] == 19341.0 and out6["ms_per_batch_one_at_a_time"] == 6.618
    assert out6["roofline_target"]["gather_frac"] == 0.42 and "lmax=4" in out6["roofline_target"]["config"]
    # a pathological record (a huge `also`) sheds `also`, never the contract keys
    fat = dict(full)
    fat["also"] = {f"w{i}": dict(full["also"]["lmax4"], config="x" * 80) for i in range(40)}
    line = bench.compact_line(fat)
    assert len(line) < bench.LINE_BUDGET and "roofline" in json.loads(line) and "cpu_baseline" in json.loads(line)


def test_library_reads_no_environment():
    """include/gotennet_hip.h: "re-entrant, no global mutable state" -- no entry point may read the process environment
    (round 3 read GN_ATTN_WAVE / GN_GEMM_* / GN_FORCE_HIGHL into function-local statics) and the wrong-result probe
    paths are gone from the product translation units."""
    import glob
    csrc = os.path.join(ROOT, "gotennet_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*")):
        text = open(path).read()
        assert "getenv" not in text, path
        for probe in ("GN_SPLIT_ABL", "GN_SPLIT_NOSTORE", "GN_SPLIT_TRACE"):
            assert probe not in text, (path, probe)
    from gotennet_amd import engine
    assert not hasattr(engine, "ACT") and not hasattr(engine, "_act_scope")


def test_embedding_map_keeps_heads_and_blocks():
    """gotennet_amd/embed.py: the channel map g of a width that is not a power of two is injective, keeps every channel in its
    attention head -- for the F-wide vectors (heads of F/H channels) AND for the flattened M*F-wide gate vectors (heads of
    M*F/H channels: reference gotennet.py:516-529) -- and embed_pack places weights accordingly (zero padding, compact
    LayerNorm intermediate, sqrt(Fp/F) folded into gamma_v's last layer)."""
    import math
    import gotennet_amd
    from gotennet_amd import embed
    for F, H, lmax in ((192, 8, 2), (96, 4, 3), (200, 8, 1), (48, 4, 2), (384, 8, 2), (24, 2, 1)):
        Fp = embed.padded_width(F)
        g = embed.channel_map(F, H)
        assert Fp >= F and (Fp & (Fp - 1)) == 0 and g.numel() == F and g.unique().numel() == F and int(g.max()) < Fp
        f = torch.arange(F)
        assert torch.equal(g // (Fp // H), f // (F // H))                        # heads over the F-wide vectors
        M = 1 + 2 * lmax
        for m in range(M):                                                       # heads over the flattened M F-wide vectors
            assert torch.equal((m * Fp + g) // (M * Fp // H), (m * F + f) // (M * F // H)), (F, H, m)
    net = gotennet_amd.GotenNet(n_atom_basis=192, n_interactions=2, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8,
                                lmax=2, sep_dir=True, sep_tensor=True)
    cfg, pw = net.config(), net.packed_weights()
    assert (cfg.F, cfg.Fc, cfg.F_model) == (256, 192, 192) and pw.F_model == 192
    idx = pw.emb_idx
    g0 = net.gata_list[0]
    assert pw.Wa.shape == (192, 512) and pw.Wb.shape == (256, 192)               # compact LayerNorm intermediate
    Wq = pw.layers[0].Wn1[:256]
    assert torch.equal(Wq[idx][:, idx], g0.W_q.weight.detach())
    mask = torch.ones(256, dtype=torch.bool); mask[idx] = False
    assert float(Wq[mask].abs().max()) == 0.0 and float(Wq[:, mask].abs().max()) == 0.0
    Wv2 = pw.layers[0].Wv2                                                        # [M Fp, Fp], scaled by sqrt(Fp / F)
    blk = Wv2[256:512][idx][:, idx]
    assert torch.allclose(blk, g0.gamma_v[1].weight.detach()[192:384] * math.sqrt(256 / 192), rtol=0, atol=0)
    # the GATA input norms run on the compact real channels (round 6) ...
    assert gotennet_amd.GotenNet(n_atom_basis=192, n_interactions=1, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8,
                                 layernorm="layer", steerable_norm="layer").config().F == 256
    with pytest.raises(NotImplementedError):                                     # ... the composed edge updates are not embedded
        gotennet_amd.GotenNet(n_atom_basis=192, n_interactions=2, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8,
                              edge_updates="mlp").config()
    with pytest.raises(NotImplementedError):
        gotennet_amd.GotenNet(n_atom_basis=100, n_interactions=1, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8).config()
