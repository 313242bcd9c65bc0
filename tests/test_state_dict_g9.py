"""A11 (optional projection, nwhead/nw.py:74-79) and the CLIP kernel's doubly registered parameter
(nw.py:82,85) against fixture G9, captured from the reference's NWNet: state_dict key lists, shapes and
parameter names on the host; forward / predict('full') with the reference's weights on the MI355X."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import T, load_golden

TAGS = {"proj": dict(feat_dim=16, proj_dim=8),
        "clip": dict(kernel_type="clip"),
        "projclip": dict(feat_dim=16, proj_dim=8, kernel_type="clip")}


class _DS(torch.utils.data.Dataset):
    def __init__(self, data, targets, C):
        self.data, self.targets, self.num_classes = data, list(targets), C

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return self.data[i], self.targets[i]


def _build(tag, device):
    from nwhead_amd.nwhead.nw import NWNet
    g = load_golden("g9_state_dict.npz")
    C = int(g["C"])
    ds = _DS(T(g["ds_data"]), g["ds_targets"].tolist(), C)
    feat = nn.Sequential(nn.Flatten(), nn.Linear(48, 16))
    net = NWNet(feat, C, support_dataset=ds, n_shot=2, n_shot_full=5, device=device, **TAGS[tag])
    return net, g


@pytest.mark.parametrize("tag", list(TAGS))
def test_state_dict_keys_shapes_and_parameter_names(tag):
    net, g = _build(tag, "cpu")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f"{tag}_keys"]]            # same names, same order
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g[f"{tag}_shapes"]]
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g[f"{tag}_param_names"]]
    # a reference checkpoint loads strictly (both kernel.* and nwhead.kernel.* present for CLIP)
    ref_sd = {str(k): T(g[f"{tag}_sd_{k}"]) for k in g[f"{tag}_keys"]}
    missing, unexpected = net.load_state_dict(ref_sd, strict=True)
    assert not missing and not unexpected
    if "clip" in tag:
        assert net.kernel.logit_scale is net.nwhead.kernel.logit_scale      # one parameter, two names


def test_projection_needs_feat_dim():
    from nwhead_amd.nwhead.nw import NWNet
    with pytest.raises(AssertionError, match="Feature dimension"):
        NWNet(nn.Flatten(), 3, proj_dim=4)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(TAGS))
def test_forward_and_full_prediction_with_reference_weights(tag):
    net, g = _build(tag, "cuda:0")
    net.load_state_dict({str(k): T(g[f"{tag}_sd_{k}"]) for k in g[f"{tag}_keys"]})
    net = net.to("cuda:0").eval()
    xq, yq = T(g["xq"]).cuda(), T(g["yq"]).cuda()
    with torch.no_grad():
        out = net(xq, yq, support_data=(T(g["sx"]), T(g["sy"]), None))
        np.testing.assert_allclose(out.cpu().numpy(), g[f"{tag}_fwd"], rtol=1e-5, atol=3e-5)
        net.precompute()
        assert list(net.full_feat.shape) == g[f"{tag}_full_feat_shape"].tolist()
        np.testing.assert_allclose(net.predict(xq, "full").cpu().numpy(), g[f"{tag}_pred_full"], rtol=1e-5, atol=3e-5)
    # the CLIP scale trains through the head (reference: kernel.py:38 is an nn.Parameter)
    if "clip" in tag:
        net.train()
        out = net(xq, yq, support_data=(T(g["sx"]), T(g["sy"]), None))
        torch.nn.functional.nll_loss(out, yq).backward()
        assert net.kernel.logit_scale.grad is not None and torch.isfinite(net.kernel.logit_scale.grad)
