"""End-to-end training step on the MI355X: backbone (MIOpen) -> NW head (HIP fwd) ->
NLL loss -> HIP backward -> SGD, through NWNet.forward with explicit support_data (H7)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _data(n_way=10, n_shot=2, B=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    protos = torch.randn(n_way, 3, 32, 32, generator=g)
    sy = torch.arange(n_way).repeat_interleave(n_shot)
    sx = protos[sy] + 0.3 * torch.randn(len(sy), 3, 32, 32, generator=g)
    y = torch.randint(0, n_way, (B,), generator=g)
    x = protos[y] + 0.3 * torch.randn(B, 3, 32, 32, generator=g)
    return x, y, sx, sy


def test_gradients_match_cpu_autograd_through_oracle_head():
    from nwhead_amd.model import load_model
    from nwhead_amd.nwhead.nw import NWNet
    from oracle import nw_oracle as O
    torch.manual_seed(0)
    net = NWNet(load_model("CIFAR_ResNet10"), 10, device="cuda:0")
    ref_feat = load_model("CIFAR_ResNet10")
    ref_feat.load_state_dict(net.featurizer.state_dict())
    net = net.to("cuda:0").train()
    ref_feat.train()
    x, y, sx, sy = _data()
    out = net(x.cuda(), y.cuda(), support_data=(sx, sy, None))
    loss = F.nll_loss(out, y.cuda())
    loss.backward()
    # CPU: same backbone, joint pass on cat(x, sx) (shared BN statistics, nw.py:182-184), oracle head
    feats = ref_feat(torch.cat((x, sx), 0))
    out_ref = O.nw_head_f32(feats[:len(x)], feats[len(x):], sy, 10)
    loss_ref = F.nll_loss(out_ref, y)
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * max(1.0, abs(loss_ref.item()))
    g_gpu = net.featurizer.conv1.weight.grad.cpu()
    g_ref = ref_feat.conv1.weight.grad
    cos = F.cosine_similarity(g_gpu.flatten(), g_ref.flatten(), dim=0).item()
    assert cos > 0.999, cos                                 # MIOpen vs CPU conv rounding aside, same gradient
    np.testing.assert_allclose(g_gpu.norm().item(), g_ref.norm().item(), rtol=2e-2)


def test_sgd_steps_reduce_the_loss():
    from nwhead_amd.model import load_model
    from nwhead_amd.nwhead.nw import NWNet
    torch.manual_seed(1)
    net = NWNet(load_model("CIFAR_ResNet10"), 10, device="cuda:0", return_mask=True).to("cuda:0").train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-4)
    losses = []
    for step in range(8):
        x, y, sx, sy = _data(seed=step % 2)
        out, isin = net(x.cuda(), y.cuda(), support_data=(sx, sy, None))
        assert bool(isin.all())
        loss = F.nll_loss(out, y.cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0], losses
    assert all(np.isfinite(losses))
