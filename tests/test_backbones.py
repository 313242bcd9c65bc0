"""A10: backbone definitions reproduce the reference's modules (fixture G6, procedural weights)."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden
from procedural import fill_procedural


@pytest.mark.parametrize("name", ["resnet18", "CIFAR_ResNet18", "densenet121", "CIFAR_DenseNet121"])
def test_backbone_matches_reference(name):
    from nwhead_amd.model import load_model
    g = load_golden("g6_backbones.npz")
    torch.manual_seed(0)
    net = fill_procedural(load_model(name))
    assert len(net.state_dict()) == int(g[f"{name}_nkeys"])          # same parameter/buffer names
    assert str(g[f"{name}_rm_name"]) in net.state_dict()
    x = T(g[f"{name}_x"])
    with torch.no_grad():
        net.eval()
        ev = net(x)                                   # DenseNet: concat-free slab path
        np.testing.assert_allclose(ev.numpy(), g[f"{name}_eval"], rtol=1e-4, atol=1e-5)
    # the autograd (torch.cat) path of the DenseNet block gives the same features as the slab path
    xg = x.clone().requires_grad_(True)
    np.testing.assert_allclose(net(xg).detach().numpy(), ev.numpy(), rtol=1e-5, atol=1e-6)
    # training forward = autograd enabled (nw_step, train.py:409): batch statistics, running stats
    # updated.  (The batch-2 DenseNet fixture ends on 2x2 maps: 8 samples per channel make the
    # batch-norm statistics ill-conditioned, so only the exact training path is compared.)
    net.train()
    tr = net(x.clone().requires_grad_(True)).detach()
    np.testing.assert_allclose(tr.numpy(), g[f"{name}_train"], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        np.testing.assert_allclose(net.state_dict()[str(g[f"{name}_rm_name"])].numpy(), g[f"{name}_rm_after"],
                                   rtol=1e-5, atol=1e-6)


def test_load_model_contract():
    from nwhead_amd.model import load_model
    with pytest.raises(KeyError):
        load_model("no_such_net")
    assert load_model("resnet18")(torch.zeros(1, 3, 64, 64)).shape == (1, 512)


def test_fold_batchnorm_matches_eval_mode():
    """model.fold_batchnorm: conv -> BN pairs of the ResNet family and of DenseNet become one
    convolution, the pre-activation nets' BN -> ReLU pairs one pass (ScaleShiftReLU); same eval-mode features
    up to fp32 re-association."""
    import torch.nn as nn
    from nwhead_amd.model import fold_batchnorm, load_model
    from tests.procedural import fill_procedural
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 64, 64, generator=g)
    for name in ("resnet18", "resnet50", "densenet121"):
        net = load_model(name)
        fill_procedural(net)
        net.train()
        with torch.no_grad():
            net(torch.randn(4, 3, 64, 64, generator=g))          # move the running statistics off their init
        net.eval()
        folded = fold_batchnorm(net)
        assert not any(isinstance(m, nn.BatchNorm2d) for m in folded.modules())
        with torch.no_grad():
            a, b = net(x), folded(x)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * a.abs().max().item()), (a - b).abs().max()
        assert any(isinstance(m, nn.BatchNorm2d) for m in net.modules())   # the original is untouched
    x32 = torch.randn(2, 3, 32, 32, generator=g)
    for name in ("CIFAR_ResNet18", "CIFAR_DenseNet121"):
        net = load_model(name)
        fill_procedural(net)
        net.train()
        with torch.no_grad():
            net(torch.randn(4, 3, 32, 32, generator=g))
        net.eval()
        folded = fold_batchnorm(net)
        assert not any(isinstance(m, nn.BatchNorm2d) for m in folded.modules())
        with torch.no_grad():
            a, b = net(x32), folded(x32)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * a.abs().max().item())


def test_g6b_well_conditioned_training_mode_on_host():
    """Fixture G6b (reference modules run by make_goldens.py; hash-procedural weights and inputs, 16 images: >= 64 samples per
    channel in every BatchNorm): our definitions' train-mode features and the last BatchNorm's running mean, on the host."""
    from conftest import load_golden
    from procedural import fill_procedural_hash, procedural_input
    from nwhead_amd.model import load_model
    g = load_golden("g6b_backbones_train.npz")
    for name in ("resnet18", "CIFAR_ResNet18", "densenet121"):
        net = fill_procedural_hash(load_model(name)).train()
        x = procedural_input(*(int(v) for v in g[f"{name}_shape"]), key=int(g[f"{name}_key"]))
        with torch.no_grad():
            out = net(x).numpy()
        scale = float(np.abs(g[f"{name}_train"]).max())
        np.testing.assert_allclose(out, g[f"{name}_train"], rtol=0, atol=1e-4 * scale)
        rm = net.state_dict()[str(g[f"{name}_rm_name"])].numpy()
        np.testing.assert_allclose(rm, g[f"{name}_rm_after"], rtol=1e-4, atol=1e-6)
