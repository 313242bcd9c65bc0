"""Training-mode BatchNorm2d + ReLU through the HIP kernels (nw_bn_relu_train_fwd_f32 / _bwd_f32): one layer
against torch's batch_norm + relu and their autograd evaluated in fp64 on the host, and whole backbones with
the fused path on vs off, each against the fp32 CPU run of the same network."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,prefix", [((6, 12, 8, 8), 12), ((5, 20, 7, 7), 20), ((3, 9, 5, 3), 9),
                                          ((42, 64, 56, 56), 64), ((4, 40, 14, 14), 24), ((2, 8, 1, 1), 8),
                                          ((8, 600, 4, 4), 600)])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_relu_train_matches_torch(shape, prefix, relu):
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + prefix)
    slab = torch.randn(*shape, generator=g) * 1.7 + torch.randn(1, shape[1], 1, 1, generator=g) * 30   # offsets >> spread
    bn_a, bn_b = nn.BatchNorm2d(prefix).cuda().train(), nn.BatchNorm2d(prefix).double().train()
    with torch.no_grad():
        bn_a.weight.copy_(torch.randn(prefix, generator=g)); bn_a.bias.copy_(torch.randn(prefix, generator=g))
        bn_a.running_mean.normal_(); bn_a.running_var.uniform_(0.5, 2.0)
        bn_b.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in bn_a.state_dict().items()})
    xa = slab.cuda()[:, :prefix].detach().requires_grad_(True)     # a channel prefix of a wider slab when prefix < C
    xb = slab[:, :prefix].double().clone().requires_grad_(True)
    ya = ops.bn_relu_train(xa, bn_a, relu)
    pre = bn_b(xb)
    yb = F.relu(pre) if relu else pre
    torch.testing.assert_close(ya.cpu().double(), yb, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(bn_a.running_mean.cpu().double(), bn_b.running_mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn_a.running_var.cpu().double(), bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
    # an activation within rounding of zero may land on either side of the ReLU: it gets no upstream gradient here
    w = torch.randn(*ya.shape, generator=g).double()
    if relu:
        w = w * (pre.detach().abs() > 1e-4)
    (ya * w.float().cuda()).sum().backward()
    (yb * w).sum().backward()
    scale = max(float(xb.grad.abs().max()), 1e-3)
    if shape[0] * shape[2] * shape[3] > 2:      # with two values per channel dx is what is left of a total cancellation
        torch.testing.assert_close(xa.grad.cpu().double() / scale, xb.grad / scale, rtol=1e-4, atol=3e-5)
    for got, ref in ((bn_a.weight.grad, bn_b.weight.grad), (bn_a.bias.grad, bn_b.bias.grad)):
        torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


def test_bn_train_statistics_on_awkward_channels():
    """Channels the one-pass shifted sums could get wrong: constant, constant plus a rounding-sized ripple, a zero
    border around an offset interior (the sampled shift sees both), and a huge offset with a tiny spread."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(0)
    n, h, w = 8, 16, 16
    x = torch.zeros(n, 5, h, w)
    x[:, 0] = 50.0
    x[:, 1] = 50.0 + 1e-5 * torch.randn(n, h, w, generator=g)
    x[:, 2, 2:-2, 2:-2] = 40.0 + 0.5 * torch.randn(n, h - 4, w - 4, generator=g)
    x[:, 3] = 1000.0 + 0.01 * torch.randn(n, h, w, generator=g)
    x[:, 4] = torch.randn(n, h, w, generator=g)
    bn_a, bn_b = nn.BatchNorm2d(5).cuda().train(), nn.BatchNorm2d(5).double().train()
    ya = ops.bn_relu_train(x.cuda(), bn_a, False)
    yb = bn_b(x.double())
    torch.testing.assert_close(bn_a.running_mean.cpu().double(), bn_b.running_mean, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(bn_a.running_var.cpu().double(), bn_b.running_var, rtol=1e-4, atol=1e-7)
    # channel 3's inputs carry 6e-5 of representation error each (fp32 at 1000) against a spread of 0.01
    torch.testing.assert_close(ya.cpu().double()[:, [0, 1, 2, 4]], yb[:, [0, 1, 2, 4]], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ya.cpu().double()[:, 3], yb[:, 3], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("shape", [(6, 12, 8, 8), (5, 20, 7, 7), (16, 128, 28, 28)])
def test_bn_add_relu_train_matches_torch(shape):
    """relu(bn(x) + residual), the tail of a ResNet block: outputs and all four gradients against fp64 torch."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x0, r0 = torch.randn(*shape, generator=g) * 2 + 5, torch.randn(*shape, generator=g)
    C = shape[1]
    bn_a, bn_b = nn.BatchNorm2d(C).cuda().train(), nn.BatchNorm2d(C).double().train()
    with torch.no_grad():
        bn_a.weight.copy_(torch.randn(C, generator=g)); bn_a.bias.copy_(torch.randn(C, generator=g))
        bn_b.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in bn_a.state_dict().items()})
    xa, ra = x0.cuda().requires_grad_(True), r0.cuda().requires_grad_(True)
    xb, rb = x0.double().requires_grad_(True), r0.double().requires_grad_(True)
    ya = ops.bn_relu_train(xa, bn_a, True, ra)
    pre = bn_b(xb) + rb
    yb = F.relu(pre)
    torch.testing.assert_close(ya.cpu().double(), yb, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(bn_a.running_var.cpu().double(), bn_b.running_var, rtol=1e-5, atol=1e-6)
    w = torch.randn(*shape, generator=g).double() * (pre.detach().abs() > 1e-4)
    (ya * w.float().cuda()).sum().backward()
    (yb * w).sum().backward()
    for got, ref in ((xa.grad, xb.grad), (ra.grad, rb.grad), (bn_a.weight.grad, bn_b.weight.grad), (bn_a.bias.grad, bn_b.bias.grad)):
        scale = max(float(ref.abs().max()), 1e-3)
        torch.testing.assert_close(got.cpu().double() / scale, ref / scale, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("arch,size,batch", [("resnet18", 96, 8), ("resnet50", 96, 8), ("densenet121", 96, 8),
                                             ("CIFAR_ResNet18", 32, 16), ("CIFAR_DenseNet121", 32, 16)])
def test_backbone_train_step_fused_vs_torch(arch, size, batch):
    """One training forward + backward of a backbone on the device with the fused BatchNorm+ReLU path on and
    off, each compared with the fp32 CPU run of the same network (what fixture G6 pins to the reference): the
    fused path must be as close to it as torch's own device path is.  (Any two correct implementations differ by
    per-cents in some gradient here -- MIOpen's fp32 Winograd convolutions against the host's direct ones, through
    training-mode BatchNorm -- which is why the yardstick is torch's own device run, not a fixed tolerance.)"""
    from nwhead_amd.model import backbones, load_model
    from tests.procedural import fill_procedural
    g = torch.Generator().manual_seed(11)
    x = torch.randn(batch, 3, size, size, generator=g)

    def run(dev, fused):
        backbones.FUSED_BN_RELU_TRAINING = fused
        try:
            torch.manual_seed(3)                       # the reference's own initialisation, same draw every time
            net = load_model(arch).to(dev).train()
            f = net(x.to(dev))
            f.square().mean().backward()
            return (f.detach().cpu(), {k: v.detach().cpu() for k, v in net.state_dict().items() if "running" in k},
                    {k: p.grad.detach().cpu() for k, p in net.named_parameters()})
        finally:
            backbones.FUSED_BN_RELU_TRAINING = True

    ref, fused, plain = run("cpu", False), run("cuda:0", True), run("cuda:0", False)

    def err(a, b):
        # relative L2 per tensor: an activation within rounding of zero may fall on either side of a ReLU in two
        # correct implementations, which moves that channel's d(gamma) by a per-cent -- max-norm would be all flips
        rel = lambda u, v, floor: float((u - v).norm()) / max(float(v.norm()), floor)
        worst = rel(a[0], b[0], 1e-12)
        for part in (1, 2):
            top = max(float(v.norm()) for v in b[part].values())
            for k in b[part]:
                worst = max(worst, rel(a[part][k].float(), b[part][k].float(), 1e-2 * top))
        return worst
    e_fused, e_plain = err(fused, ref), err(plain, ref)
    print(f"{arch}: fused vs cpu {e_fused:.2e}   torch-device vs cpu {e_plain:.2e}")
    assert e_fused <= max(4 * e_plain, 1e-2), (e_fused, e_plain)


@pytest.mark.parametrize("shape,extra", [((6, 24, 8, 8), 8), ((5, 10, 7, 7), 3)])
def test_bn_relu_passthrough_accumulates_the_other_gradient(shape, extra):
    """bn_relu_train(..., passthrough=True): x goes through the node and on into a concatenation (a dense block's
    running concatenation); the gradient that comes back through the concatenation -- a channel-prefix slice, not
    contiguous -- is added to dx inside the backward kernel."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    n, c, h, w = shape
    x0, new0 = torch.randn(*shape, generator=g), torch.randn(n, extra, h, w, generator=g)
    bn_a, bn_b = nn.BatchNorm2d(c).cuda().train(), nn.BatchNorm2d(c).double().train()
    xa, xb = x0.cuda().requires_grad_(True), x0.double().requires_grad_(True)
    a, xpass = ops.bn_relu_train(xa, bn_a, True, None, passthrough=True)
    cat_a = torch.cat((xpass, new0.cuda()), 1)
    ref_pre = bn_b(xb)
    cat_b = torch.cat((xb, new0.double()), 1)
    w1 = torch.randn(*shape, generator=g).double() * (ref_pre.detach().abs() > 1e-4)
    w2 = torch.randn(n, c + extra, h, w, generator=g).double()
    ((a * w1.float().cuda()).sum() + (cat_a * w2.float().cuda()).sum()).backward()
    ((F.relu(ref_pre) * w1).sum() + (cat_b * w2).sum()).backward()
    scale = float(xb.grad.abs().max())
    torch.testing.assert_close(xa.grad.cpu().double() / scale, xb.grad / scale, rtol=1e-4, atol=3e-5)
    torch.testing.assert_close(bn_a.weight.grad.cpu().double(), bn_b.weight.grad, rtol=1e-4, atol=1e-4 * float(bn_b.weight.grad.abs().max()))
    # only the pass-through output used: the node is a plain identity for x
    xa2 = x0.cuda().requires_grad_(True)
    _, xp = ops.bn_relu_train(xa2, nn.BatchNorm2d(c).cuda().train(), True, None, passthrough=True)
    (xp * w2[:, :c].float().cuda()).sum().backward()
    torch.testing.assert_close(xa2.grad.cpu().double(), w2[:, :c], rtol=1e-6, atol=1e-6)


def test_bias_act_nhwc_and_the_folded_channels_last_resnets():
    """ops.bias_act_nhwc_ (folded BatchNorm bias + a block's identity + ReLU in one in-place pass over a channels_last
    activation) against torch, and the folded channels_last ResNets that use it (BasicBlock and Bottleneck forms)
    against the plain eval-mode networks."""
    from nwhead_amd import ops
    from nwhead_amd.model import fold_batchnorm, load_model
    from nwhead_amd.model.backbones import ConvBiasAct
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for n, c, h, w in ((3, 64, 7, 5), (2, 4, 1, 9), (5, 132, 6, 6)):
        x = torch.randn(n, c, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        r = torch.randn(n, c, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        b = torch.randn(c, generator=g).to(dev)
        for res in (None, r):
            for relu in (False, True):
                want = x + b.view(1, -1, 1, 1) + (0 if res is None else res)
                want = torch.relu(want) if relu else want
                got = ops.bias_act_nhwc_(x.clone(memory_format=torch.preserve_format), b, res, relu)
                assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.bias_act_nhwc_(torch.randn(2, 8, 4, 4, device=dev), torch.zeros(8, device=dev))     # NCHW memory
    for name in ("resnet18", "resnet50"):
        torch.manual_seed(0)
        net = load_model(name).to(dev)
        net.train()
        with torch.no_grad():
            net(torch.randn(4, 3, 96, 96, device=dev))
        net.eval()
        folded = fold_batchnorm(net).to(memory_format=torch.channels_last)
        assert sum(isinstance(m, ConvBiasAct) for m in folded.modules()) == (20 if name == "resnet18" else 53)
        x = torch.randn(8, 3, 96, 96, device=dev)
        with torch.no_grad():
            want = net(x)
            got = folded(x.contiguous(memory_format=torch.channels_last))
            got_nchw = folded(x)                      # NCHW input: the torch ops
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(got_nchw.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
