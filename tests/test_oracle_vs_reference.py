"""CPU, build container only: oracle vs the LIVE reference imported from /root/reference.
Skipped wherever the reference is not mounted (e.g. on the GPU box)."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models", "superglue")),
                                reason="reference not mounted")


def _ref_module(cfg, sd):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.superglue.superglue import SuperGlue
    ref = SuperGlue(cfg)
    ref.load_state_dict(sd, strict=True)
    return ref.eval()


@pytest.mark.parametrize("m,n,kw", [
    (50, 73, dict(descriptor_dim=64, num_stages=2, num_heads=4, num_iters=3, side_info_size=6)),
    (128, 96, dict(descriptor_dim=128, num_stages=2, num_heads=4, num_iters=15, side_info_size=1, use_offset=True)),
])
def test_oracle_equals_live_reference(m, n, kw):
    from openglue_amd import synthetic as syn
    from oracle import superglue_oracle as orc
    cfg = syn.make_config(**kw)
    sd = syn.make_state_dict(cfg, seed=3)
    data = syn.make_batch(2, m, n, cfg["descriptor_dim"], cfg["positional_encoding"]["side_info_size"], seed=9)
    with torch.no_grad():
        r = _ref_module(cfg, sd)(data)
        o = orc.superglue_forward(sd, cfg, data)
        # the `image0` tensor path (size()[-2:] = (H, W)) equals the image0_size=[W,H] path
        d2 = {k: v for k, v in data.items() if not k.endswith("_size")}
        d2["image0"] = torch.empty(2, 1, 720, 960)
        d2["image1"] = torch.empty(2, 1, 720, 960)
        o2 = orc.superglue_forward(sd, cfg, d2)
    assert (r["scores"] - o["scores"]).abs().max() < 1e-4
    assert (r["context_descriptors0"] - o["context_descriptors0"]).abs().max() < 2e-5
    assert torch.equal(o["scores"], o2["scores"])
