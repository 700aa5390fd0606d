"""CPU, build container only: the oracle against the LIVE reference (imported from /root/reference through
tools/ref_shims.py) on fresh seeded cases, in fp64.  Skipped wherever the reference checkout is absent
(e.g. the GPU box); the committed goldens in tests/golden/ carry the same pin there."""
import os

import pytest
import torch

REF = "/root/reference/gotennet"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


@pytest.mark.parametrize("lmax,sep,scale", [(1, False, True), (2, True, False), (3, True, True), (4, True, False)])
def test_oracle_matches_live_reference_fp64(lmax, sep, scale):
    from tools import ref_shims
    ref = ref_shims.import_reference()
    from gotennet.models.components import layers as ref_layers
    from oracle import gotennet_oracle as orc
    torch.manual_seed(100 + lmax)
    hp = dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=lmax, num_heads=4, scale_edge=scale,
              sep_dir=sep, sep_tensor=sep, max_z=12)
    net = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(5.0), **hp).double().eval()
    with torch.no_grad():
        for n, p in net.named_parameters():            # non-zero biases / affine so nothing hides behind init
            if p.dim() == 1:
                p.uniform_(0.9, 1.1) if n.endswith("norm.weight") else p.uniform_(-0.1, 0.1)
    g = torch.Generator().manual_seed(lmax)
    pos = torch.cat([torch.rand((7, 3), generator=g, dtype=torch.float64) * 3.0 + 10.0 * b for b in range(2)])
    batch = torch.arange(2).repeat_interleave(7)
    z = torch.randint(1, 9, (14,), generator=g)
    cfg = orc.default_config(cutoff=5.0, **{k: v for k, v in hp.items() if k != "max_z"})
    ei, w, vec = orc.distance(pos, batch, 5.0)
    with torch.no_grad():
        h_ref, X_ref = net(z, ei, w.clone(), vec.clone())      # the reference normalises edge_vec in place
    sd = {k: v for k, v in net.state_dict().items()}
    h, X = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    assert float((h - h_ref).abs().max() / h_ref.abs().max()) < 1e-12
    assert float((X - X_ref).abs().max() / X_ref.abs().max()) < 1e-12
