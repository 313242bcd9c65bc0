"""The channels-last convolution path on the fp16 matrix cores (csrc/conv_nhwc.hip, conv_wgrad.hip, bn_nhwc.hip; split-fp16
operands, fp32-grade accuracy) against torch in fp64: forward (3x3 PATCH mode, 1x1 / strided GATHER mode, the stems' ROWRUN
mode), weight gradient, data gradient, BatchNorm + ReLU forward / backward, the amax records, the batched weight split, and
the backbones that use them (folded channels_last ResNets: BASELINE config 2 end to end; DenseNet training step)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 3e-6          # relative to the largest output: K up to 4608 products of fp32-grade (2^-22) accuracy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad, bias, res, relu
    (2, 32, 8, 8, 32, 3, 1, 1, False, False, False),       # PATCH, 256-pixel tiles of 32 channels
    (3, 64, 9, 7, 64, 3, 1, 1, True, True, True),          # PATCH, tiles crossing rows and images, ragged tail
    (2, 64, 56, 56, 64, 3, 1, 1, True, True, True),        # ResNet layer1
    (4, 128, 28, 28, 128, 3, 1, 1, True, True, True),      # layer2, 128 x 128 tiles
    (16, 64, 56, 56, 64, 3, 1, 1, True, True, True),       # enough pixels for the 256 x 64 tiles (PATCH)
    (17, 96, 56, 56, 64, 1, 1, 0, True, False, True),      # ... and through GATHER, ragged last tile
    (9, 512, 7, 7, 512, 3, 1, 1, True, True, True),        # layer4: K = 4608
    (3, 128, 56, 56, 32, 3, 1, 1, False, False, False),    # DenseNet conv2
    (2, 64, 56, 56, 128, 1, 1, 0, True, False, True),      # GATHER: 1x1
    (3, 256, 28, 28, 128, 1, 1, 0, False, False, False),
    (2, 64, 56, 56, 128, 3, 2, 1, True, True, True),       # strided 3x3
    (2, 64, 56, 56, 128, 1, 2, 0, True, False, False),     # strided 1x1 projection
    (3, 992, 7, 7, 128, 1, 1, 0, False, False, False),     # DenseNet block 4 conv1
    (2, 96, 10, 12, 64, 5, 1, 2, True, True, True),        # another kernel size through GATHER
    (2, 32, 30, 30, 96, 3, 1, 0, True, True, True),        # 3x3 without padding
    (3, 3, 64, 64, 64, 7, 2, 3, True, False, True),        # ROWRUN: the 7x7 / 2 stem over RGB (4-channel padded input)
    (5, 3, 32, 32, 64, 3, 1, 1, True, False, True),        # CIFAR stem
    (2, 1, 28, 28, 32, 5, 1, 2, True, False, True),        # one input channel: the scalar ROWRUN path
    (42, 64, 14, 14, 128, 1, 1, 0, True, True, True),      # 64 pixels x 128 channels: 129 tiles instead of 258 of 64 x 64 (GATHER)
    (42, 32, 14, 14, 128, 3, 1, 1, True, True, True),      # ... and PATCH (the data gradient of a DenseNet conv2 at 14 x 14)
]


def _conv_case(dev, ops, case, seed=0):
    n, cin, h, w, cout, k, stride, pad, bias, res, relu = case
    g = torch.Generator().manual_seed(seed + 13 * cin + h)
    x = _cl((torch.randn(n, cin, h, w, generator=g) * 1.7 + 0.3).to(dev))
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev) if bias else None
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = _cl(torch.randn(n, cout, ho, wo, generator=g).to(dev)) if res else None
    y = ops.conv2d_nhwc(x, ops.SplitConvWeight(wt), b, r, relu, stride, pad)
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), stride, pad)
    if r is not None:
        ref = ref + r.double()
    if relu:
        ref = ref.relu()
    return y, ref


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_nhwc_against_torch_fp64(dev, case):
    from nwhead_amd import ops
    y, ref = _conv_case(dev, ops, case)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.shape == ref.shape
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < TOL, err
    # the amax record the store leaves: its maximum is max|y| exactly (per-workgroup maxima, no atomics)
    assert y.nw_amax.shape == (ops.AMAX_SLOTS,)
    assert float(y.nw_amax.max()) == float(y.abs().max())


def test_conv2d_nhwc_many_tiles_per_workgroup():
    """The persistent tile loop (loaders running ahead across tile boundaries) with 8 workgroups for up to 196 tiles:
    NW_CONV_MAX_WGS is read once per process, hence the subprocess."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
            "import torch\nfrom nwhead_amd import ops\nimport test_conv_nhwc_gpu as t\n"
            "dev = torch.device('cuda:0')\n"
            "for case in t.CONV_CASES:\n"
            "    y, ref = t._conv_case(dev, ops, case, seed=5)\n"
            "    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()\n"
            "    assert err < t.TOL, (case, err)\n"
            "for args in t.MOMENT_CASES:\n"          # a workgroup's tiles merged into one moments group each (Chan, in registers)
            "    t.test_conv_channel_windows_and_moments(dev, *args)\n"
            "for flag in (False, True, 'fused_norm1'):\n"            # ... and the backward sums of the data gradients' epilogues
            "    t.test_dense_block_slab_node_against_layer_by_layer(dev, flag)\n"
            "print('OK')\n" % (ROOT, ROOT))
    env = dict(os.environ, NW_CONV_MAX_WGS="8")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_conv_scale_follows_the_amax_record(dev):
    """One power of two per tensor from the amax record: a bound 1000x too large costs accuracy but stays finite and
    close; activations of magnitude 1e-20 and 1e+20 go through (the scale brings them into fp16's range)."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(3)
    wt = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    sw = ops.SplitConvWeight(wt)
    for mag in (1e-20, 1.0, 1e20):
        x = _cl((torch.randn(2, 64, 12, 12, generator=g) * mag).to(dev))
        y = ops.conv2d_nhwc(x, sw, None, None, False, 1, 1)
        ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
        assert (y.double() - ref).abs().max().item() / ref.abs().max().item() < TOL
    x = _cl(torch.randn(2, 64, 12, 12, generator=g).to(dev))
    loose = ops.absmax(x) * 1000.0
    y = ops.conv2d_nhwc(x, sw, None, None, False, 1, 1, amax=loose)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-4


def test_absmax_and_channel_padding(dev):
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 3, 37, 41, generator=g).to(dev)
    x[3, 1, 20, 7] = -9.5
    for t in (x, _cl(x)):
        a = ops.absmax(t)
        assert a.shape == (ops.AMAX_SLOTS,) and float(a.max()) == 9.5
        p = ops.to_nhwc_pad(t, 4)
        assert p.shape == (5, 4, 37, 41) and p.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(p[:, :3], x) and float(p[:, 3].abs().max()) == 0.0 and float(p.nw_amax.max()) == 9.5


WGRAD_CASES = [(2, 64, 8, 8, 128, 1), (3, 96, 14, 14, 128, 1), (2, 256, 28, 28, 128, 1), (2, 1024, 7, 7, 512, 1),
               (2, 128, 8, 8, 32, 3), (3, 128, 14, 14, 32, 3), (2, 128, 56, 56, 32, 3), (5, 128, 7, 7, 32, 3),
               (2, 64, 28, 28, 64, 3), (3, 40, 9, 11, 24, 3), (3, 40, 9, 11, 24, 1)]


@pytest.mark.parametrize("n,cin,h,w,cout,k", WGRAD_CASES)
def test_conv_weight_gradient_against_torch_fp64(dev, n, cin, h, w, cout, k):
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(n + cin + h)
    x = _cl((torch.randn(n, cin, h, w, generator=g) + 0.2).to(dev))
    gy = _cl(torch.randn(n, cout, h, w, generator=g).to(dev))
    pad = (k - 1) // 2
    dw = ops.conv2d_nhwc_wgrad(x, gy, (cout, cin, k, k), 1, pad)
    assert dw.shape == (cout, cin, k, k)
    ref = torch.ops.aten.convolution_backward(gy.double().contiguous(), x.double().contiguous(),
                                              torch.empty(cout, cin, k, k, dtype=torch.float64, device=dev), None, [1, 1],
                                              [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert (dw.double() - ref).abs().max().item() / ref.abs().max().item() < TOL
    dw2 = ops.conv2d_nhwc_wgrad(x, gy, (cout, cin, k, k), 1, pad)
    assert torch.equal(dw, dw2)                            # chunk sums added in a fixed order


STEM_WGRAD_CASES = [(3, 64, 64, 64, 7, 2, 3), (4, 37, 53, 32, 7, 2, 3), (5, 32, 32, 64, 3, 1, 1), (2, 28, 30, 16, 5, 1, 2),
                    (2, 40, 40, 64, 7, 2, 0), (42, 224, 224, 64, 7, 2, 3)]


@pytest.mark.parametrize("n,h,w,cout,k,stride,pad", STEM_WGRAD_CASES)
def test_stem_weight_gradient_against_torch_fp64(dev, n, h, w, cout, k, stride, pad):
    """Round 4: the weight gradient of the few-channel stems -- 7x7 / 2 over RGB (model/densenet.py:114-116,
    model/resnet.py:147; the last shape is K4's), CIFAR's 3x3 -- as one row-run job per kernel row (nw_wgrad_job.rowrun_stride)
    instead of MIOpen's kernel: against fp64, odd image sizes (ragged right / bottom borders), no padding, repeatable."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(n + h + k)
    x = (torch.randn(n, 3, h, w, generator=g) * 1.5 + 0.3).to(dev)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    gy = _cl(torch.randn(n, cout, ho, wo, generator=g).to(dev))
    dw = ops.conv2d_nhwc_wgrad(_cl(x), gy, (cout, 3, k, k), stride, pad)
    assert dw.shape == (cout, 3, k, k)
    ref = torch.ops.aten.convolution_backward(gy.double().contiguous(), x.double().contiguous(),
                                              torch.empty(cout, 3, k, k, dtype=torch.float64, device=dev), None, [stride, stride],
                                              [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert (dw.double() - ref).abs().max().item() / ref.abs().max().item() < TOL
    assert torch.equal(dw, ops.conv2d_nhwc_wgrad(x, gy, (cout, 3, k, k), stride, pad))      # (NCHW input: the padding pass takes either)


def test_batched_weight_gradients_equal_the_single_calls(dev):
    """nw_conv2d_nhwc_wgrad_batch_f16x2: 3x3 and 1x1 problems of different sizes in one call -- operands that are channel
    windows of wider tensors (ldx / ldg), both output layouts (out_oihw) -- against fp64; repeatable bit for bit; a batch
    with one unsupported job is refused as a whole and launches nothing."""
    import ctypes
    from nwhead_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    specs = [(3, 160, 14, 14, 128, 1, 32, 0, 0), (3, 128, 14, 14, 32, 3, 0, 96, 1), (5, 128, 7, 7, 32, 3, 0, 0, 0),
             (2, 64, 28, 28, 128, 1, 0, 0, 1), (2, 128, 28, 28, 32, 3, 64, 32, 1), (4, 224, 7, 7, 128, 1, 0, 0, 0)]
    keep, jobs, outs, refs = [], [], [], []
    for n, cin, h, w, cout, k, xw, gw, oihw in specs:
        wx = _cl((torch.randn(n, cin + xw, h, w, generator=g) + 0.2).to(dev))
        wg = _cl(torch.randn(n, cout + gw, h, w, generator=g).to(dev))
        x, gy = wx[:, xw:], wg[:, :cout]                       # windows: not at channel 0 / a prefix
        ax, ag = ops.absmax(wx), ops.absmax(wg)
        dw = torch.full((cout, cin, k, k) if oihw else (cout, k, k, cin), float("nan"), device=dev)
        pad = (k - 1) // 2
        jobs.append(_lib.WgradJob(x.data_ptr(), ax.data_ptr(), gy.data_ptr(), ag.data_ptr(), dw.data_ptr(), n, h, w, cin, cout, k, k,
                                  1, pad, cin + xw, cout + gw, oihw))
        keep += [wx, wg, ax, ag]
        outs.append(dw if oihw else dw.permute(0, 3, 1, 2))
        refs.append(torch.ops.aten.convolution_backward(gy.double().contiguous(), x.double().contiguous(),
                                                        torch.empty(cout, cin, k, k, dtype=torch.float64, device=dev), None, [1, 1],
                                                        [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1])
    arr = (_lib.WgradJob * len(jobs))(*jobs)
    wsb = lib.nw_conv2d_nhwc_wgrad_batch_workspace_bytes(arr, len(jobs))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    st = ops._stream(keep[0])
    assert lib.nw_conv2d_nhwc_wgrad_batch_f16x2(arr, len(jobs), ws.data_ptr(), wsb, st) == 0
    first = [o.clone() for o in outs]
    for o, r in zip(outs, refs):
        assert (o.double() - r).abs().max().item() / r.abs().max().item() < TOL
    assert lib.nw_conv2d_nhwc_wgrad_batch_f16x2(arr, len(jobs), ws.data_ptr(), wsb, st) == 0
    assert all(torch.equal(a, b) for a, b in zip(first, outs))
    # too small a workspace; then one job the kernels do not serve (stride 2): refused as a whole, outputs untouched
    assert lib.nw_conv2d_nhwc_wgrad_batch_f16x2(arr, len(jobs), ws.data_ptr(), max(wsb - 16, 0), st) != 0 or wsb == 0
    for o in outs:
        o.fill_(-3.0)
    bad = list(jobs)
    j = bad[2]
    bad[2] = _lib.WgradJob(j.x, j.amax_x, j.gy, j.amax_g, j.dw, j.n, j.H, j.W, j.Cin, j.Cout, j.KH, j.KW, 2, j.pad, j.ldx, j.ldg, 0)
    arr2 = (_lib.WgradJob * len(bad))(*bad)
    assert lib.nw_conv2d_nhwc_wgrad_batch_f16x2(arr2, len(bad), ws.data_ptr(), wsb, st) != 0
    torch.cuda.synchronize()
    assert all(bool((o == -3.0).all()) for o in outs)
    assert lib.nw_conv2d_nhwc_wgrad_batch_f16x2(arr, 0, None, 0, st) == 0          # an empty batch is a no-op


@pytest.mark.parametrize("n,cin,h,w,cout,k,stride,pad", [(3, 64, 14, 14, 128, 1, 1, 0), (2, 128, 28, 28, 32, 3, 1, 1),
                                                         (2, 160, 14, 14, 128, 1, 1, 0), (2, 64, 28, 28, 128, 3, 2, 1),
                                                         (2, 3, 32, 32, 64, 7, 2, 3),
                                                         # the ResNets' strided layers on the own kernels (round 4): 3x3 / 2, 1x1 / 2, odd maps
                                                         (3, 64, 56, 56, 128, 3, 2, 1), (3, 64, 56, 56, 128, 1, 2, 0),
                                                         (2, 128, 15, 13, 256, 3, 2, 1), (2, 256, 7, 9, 512, 1, 2, 0),
                                                         (6, 512, 3, 3, 512, 3, 1, 1), (6, 256, 6, 6, 512, 3, 2, 1)])   # ResNet-18 layer4 at 96 x 96 inputs
def test_conv_autograd_node(dev, n, cin, h, w, cout, k, stride, pad):
    """ops.conv2d_nhwc_train: forward, data gradient (the same kernel on the flipped, transposed weight; for the strided
    many-channel shapes over gy with zeros between its pixels) and weight gradient (strided: one 1x1 problem per tap) against
    fp64 autograd, with and without a ConvWeightBank."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(cin + k)
    x0 = _cl(torch.randn(n, cin, h, w, generator=g).to(dev))
    w0 = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    x64, w64 = x0.double().requires_grad_(cin != 3), w0.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, stride, pad)
    t = torch.randn(y64.shape, generator=g).to(dev)
    (y64 * t.double()).sum().backward()
    wp = torch.nn.Parameter(w0.clone())
    for bank in (None, ops.ConvWeightBank([(wp, cin != 3)])):
        wp.grad = None
        x = x0.clone().requires_grad_(cin != 3)
        if bank is not None:
            bank.refresh()
        y = ops.conv2d_nhwc_train(x, wp, stride, pad, operands=None if bank is None else bank.operands(wp))
        (y * _cl(t)).sum().backward()
        assert (y.double() - y64).abs().max().item() / y64.abs().max().item() < TOL
        assert (wp.grad.double() - w64.grad).abs().max().item() / w64.grad.abs().max().item() < 2e-5
        if cin != 3:
            assert (x.grad.double() - x64.grad).abs().max().item() / x64.grad.abs().max().item() < 2e-5


def test_weight_bank_operands_equal_the_per_weight_split(dev):
    """nw_split_conv_weights_f16x2 (one launch for all weights) writes the very bytes SplitConvWeight does, forward and
    data-gradient layouts, and follows in-place updates."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(9)
    ws = [torch.nn.Parameter((torch.randn(s, generator=g) * 10 ** float(e)).to(dev))
          for s, e in (((64, 3, 7, 7), 0), ((128, 64, 1, 1), -3), ((32, 128, 3, 3), 2), ((96, 160, 1, 1), 0))]
    bank = ops.ConvWeightBank([(w, i > 0) for i, w in enumerate(ws)])
    for rnd in range(2):
        bank.refresh()
        for i, w in enumerate(ws):
            fw, dg = bank.operands(w)
            one = ops.SplitConvWeight(w)
            assert fw.shape == one.shape and torch.equal(fw.split.view(-1), one.split.view(-1)) and torch.equal(fw.scale, one.scale)
            if i > 0 and w.shape[0] % 32 == 0:
                t = ops.SplitConvWeight(w.detach().flip(2, 3).transpose(0, 1))
                assert dg.shape == t.shape and torch.equal(dg.split.view(-1), t.split.view(-1)) and torch.equal(dg.scale, t.scale)
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5).add_(0.01)


@pytest.mark.parametrize("n,c,h,w,prefix,relu", [(4, 64, 12, 12, 0, True), (3, 160, 7, 7, 32, True), (2, 32, 30, 17, 0, False),
                                                (6, 256, 14, 14, 0, True), (2, 1024, 7, 7, 0, True),
                                                (3, 1664, 7, 7, 0, True), (2, 2592, 4, 4, 0, True)])   # DenseNet-169's last norm; beyond the kernels' 2560: torch
def test_bn_relu_nhwc_train_against_fp64(dev, n, c, h, w, prefix, relu):
    """nw_bn_relu_nhwc_train_fwd/bwd: outputs, running statistics, dx, dgamma, dbeta against fp64 torch, also on a channel
    prefix of a wider channels-last tensor (row stride > c) and with a pass-through gradient."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(c + h)
    full = _cl((torch.randn(n, c + prefix, h, w, generator=g) * 2.0 + 0.7).to(dev))
    x0 = full[:, :c]
    bn = torch.nn.BatchNorm2d(c).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    ref = torch.nn.BatchNorm2d(c).to(dev).double().train()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    x = x0.detach().clone().requires_grad_(True) if prefix == 0 else x0.detach().requires_grad_(True)
    y = ops.bn_relu_train_nhwc(x, bn, relu)
    t = torch.randn(y.shape, generator=g).to(dev)
    (y * t).sum().backward()
    x64 = x0.detach().double().requires_grad_(True)
    y64 = ref(x64)
    y64 = y64.relu() if relu else y64
    (y64 * t.double()).sum().backward()
    sc = lambda a: max(float(a.detach().abs().max()), 1e-12)
    assert (y.double() - y64).abs().max().item() < 2e-6 * sc(y64)
    if c <= ops.BN_NHWC_MAX_C:
        assert float(y.nw_amax.max()) == float(y.detach().abs().max())
    assert (x.grad.double() - x64.grad).abs().max().item() < 1e-5 * sc(x64.grad)
    assert (bn.weight.grad.double() - ref.weight.grad).abs().max().item() < 1e-5 * sc(ref.weight.grad)
    assert (bn.bias.grad.double() - ref.bias.grad).abs().max().item() < 1e-5 * sc(ref.bias.grad)
    assert (bn.running_mean.double() - ref.running_mean).abs().max().item() < 1e-6 * sc(ref.running_mean)
    assert (bn.running_var.double() - ref.running_var).abs().max().item() < 1e-5 * sc(ref.running_var)
    assert int(bn.num_batches_tracked) == 1


FUSED_NORM1_CASES = [(2, 14, 14, 96, 160, 128), (3, 7, 7, 992, 1024, 128), (1, 5, 3, 32, 32, 32), (2, 28, 28, 288, 320, 128),
                     (4, 9, 11, 64, 64, 64), (2, 12, 12, 544, 576, 96), (42, 14, 14, 256, 288, 128)]


@pytest.mark.parametrize("n,h,w,c,ctot,mid", FUSED_NORM1_CASES)
def test_fused_norm1_backward_against_fp64(dev, n, h, w, c, ctot, mid):
    """nw_bn_dgrad1x1_bwd_f16x2 (round 4; backward of model/densenet.py:36-40 norm1 -> relu1 -> conv1): the gradient slab's
    prefix, dgamma, dbeta and the amax record against fp64, and against the two-step path (data-gradient convolution +
    nw_bn_relu_nhwc_train_bwd_f32) it replaces -- channel prefix of a wider slab, channels with an offset, negative gamma,
    channel groups of unequal size (c = 288: 192 + 96), fewer than 16 pixels, a ragged last strip."""
    from nwhead_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(c + mid + h)
    rows = n * h * w
    slab = (torch.randn(rows, ctot, generator=g) * (0.3 + 1.5 * torch.rand(1, ctot, generator=g)) + 2.0 * torch.randn(1, ctot, generator=g)).to(dev)
    gamma = ((torch.rand(c, generator=g) + 0.5) * torch.where(torch.rand(c, generator=g) < 0.2, -1.0, 1.0)).to(dev)
    beta = (torch.randn(c, generator=g) * 0.5).to(dev)
    wt = (torch.randn(mid, c, generator=g) / c ** 0.5).to(dev)                 # conv1's weight (Cout = mid, Cin = c)
    du = (torch.randn(rows, mid, generator=g) * 0.7).to(dev)
    G0 = torch.randn(rows, ctot, generator=g).to(dev)
    x = slab[:, :c]
    mean = x.double().mean(0)
    var = x.double().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    tab = torch.cat([mean, gamma.double() * invstd, beta.double()]).float().contiguous()
    inv32 = invstd.float().contiguous()
    m32 = mean.float().contiguous()
    # fp64 reference from the fp32 table (what the forward used)
    a64, b64, m64, i64 = tab[c:2 * c].double(), tab[2 * c:].double(), m32.double(), inv32.double()
    g64 = du.double() @ wt.double()
    xm = x.double() - m64
    gd = torch.where(xm * a64 + b64 > 0, g64, torch.zeros_like(g64))
    dbeta, dgamma = gd.sum(0), (gd * xm * i64).sum(0)
    dx = a64 * (gd - dbeta / rows - xm * i64 * dgamma / rows)
    want = G0.double().clone()
    want[:, :c] += dx
    # the fused call
    d1 = ops.SplitConvWeight(wt.t().reshape(c, mid, 1, 1).contiguous())
    am_du = ops.absmax(du)
    G = G0.clone()
    am_g = torch.empty(ops.AMAX_SLOTS, device=dev)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    fb = lib.nw_bn_dgrad1x1_workspace_bytes(rows, c)
    ws = torch.empty(fb // 4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    _lib.check(lib.nw_bn_dgrad1x1_bwd_f16x2(P(du), P(am_du), P(d1.split), P(d1.scale), P(slab), ctot, P(tab), c, P(inv32), P(G), ctot,
                                            P(am_g), P(dg), P(db), P(ws), fb, rows, c, mid, st), "nw_bn_dgrad1x1_bwd_f16x2")
    sc = lambda t: max(float(t.abs().max()), 1e-12)
    assert (G[:, :c].double() - want[:, :c]).abs().max().item() < 2e-5 * sc(want[:, :c])
    assert torch.equal(G[:, c:], G0[:, c:])                                  # nothing outside the prefix is touched
    assert (dg.double() - dgamma).abs().max().item() < 2e-5 * sc(dgamma)
    assert (db.double() - dbeta).abs().max().item() < 2e-5 * sc(dbeta)
    assert float(am_g.max()) == float(G[:, :c].abs().max())
    # the two-step path on the same operands
    dt1 = torch.empty(rows, c, device=dev)
    am_t = torch.empty(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.nw_conv2d_nhwc_f16x2(P(du), P(am_du), P(d1.split), P(d1.scale), None, None, 0, P(dt1), P(am_t), n, h, w, mid, c, 1, 1,
                                        1, 0, 0, 0, None, st), "nw_conv2d_nhwc_f16x2")
    G2 = G0.clone()
    dg2, db2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    bnb = lib.nw_bn_nhwc_workspace_bytes(rows, c)
    ws2 = torch.empty(bnb // 4 + 4, device=dev)
    am2 = torch.empty(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.nw_bn_relu_nhwc_train_bwd_f32(P(slab), ctot, P(dt1), P(gamma), P(beta), P(m32), P(inv32), P(G2), P(dg2), P(db2), P(G2),
                                                 ctot, ctot, P(am2), P(ws2), bnb, rows, c, 1, st), "nw_bn_relu_nhwc_train_bwd_f32")
    assert (G[:, :c] - G2[:, :c]).abs().max().item() < 2e-5 * sc(want[:, :c])
    assert (dg - dg2).abs().max().item() < 2e-5 * sc(dgamma) and (db - db2).abs().max().item() < 2e-5 * sc(dbeta)


@pytest.mark.parametrize("shape", [(3, 64, 7, 5), (2, 128, 28, 28), (1, 4, 1, 1), (5, 36, 9, 3)])
def test_add_relu_nhwc_against_torch(dev, shape):
    """ops.add_relu_nhwc (nw_add_relu_f32 / nw_relu_bwd_f32: the end of a residual block, model/resnet.py:60-66): values bit-equal
    to torch's relu(a + b) incl. a NaN, the amax records of the result and of the gradient, one masked gradient for both summands."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    a = _cl(torch.randn(*shape, generator=g).to(dev)).requires_grad_(True)
    b = _cl(torch.randn(*shape, generator=g).to(dev)).requires_grad_(True)
    t = _cl(torch.randn(*shape, generator=g).to(dev))
    out = ops.add_relu_nhwc(a, b)
    ref = torch.relu(a.detach() + b.detach())
    assert torch.equal(out.detach(), ref) and out.is_contiguous(memory_format=torch.channels_last)
    assert float(out.nw_amax.max()) == float(ref.abs().max()) and out.nw_amax.shape == (ops.AMAX_SLOTS,)
    (out * t).sum().backward()
    want = torch.where(ref > 0, t, torch.zeros_like(t))
    assert torch.equal(a.grad, want) and torch.equal(b.grad, want)
    with torch.no_grad():
        a2 = a.detach().clone()
        a2[0, 0, 0, 0] = float("nan")
        o2 = ops.add_relu_nhwc(a2, b.detach())
        assert torch.isnan(o2[0, 0, 0, 0]) and torch.equal(o2.reshape(-1)[1:], ref.reshape(-1)[1:])


@pytest.mark.parametrize("c", [64, 2592])
def test_bn_relu_nhwc_cumulative_average(dev, c):
    """BatchNorm2d(momentum=None): running statistics are the cumulative average (factor 1 / num_batches_tracked, counted
    first) on the kernels' path and on the wider-than-2560-channels fallback (it used factor 0)."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(c)
    bn = torch.nn.BatchNorm2d(c, momentum=None).to(dev).train()
    ref = torch.nn.BatchNorm2d(c, momentum=None).to(dev).double().train()
    for step in range(3):
        x = _cl((torch.randn(3, c, 5, 5, generator=g) * (1.0 + step) + 0.5 * step).to(dev))
        ops.bn_relu_train_nhwc(x, bn, True)
        ref(x.double())
    assert int(bn.num_batches_tracked) == 3
    assert (bn.running_mean.double() - ref.running_mean).abs().max().item() < 1e-5
    assert (bn.running_var.double() - ref.running_var).abs().max().item() < 1e-5 * float(ref.running_var.max())


@pytest.mark.parametrize("n,c,h,w,prefix", [(2, 8, 9, 11, 0), (3, 64, 56, 56, 0), (2, 128, 28, 28, 32), (1, 4, 2, 2, 0),
                                            (2, 12, 7, 1, 0), (42, 64, 112, 112, 0)])
def test_pools_nhwc_against_torch(dev, n, c, h, w, prefix):
    """nw_avgpool2x2_nhwc / nw_maxpool3x3s2_nhwc forward and backward against torch's pools on the same channels-last input:
    forward bit-equal (same window scan, same tie rule -- quantised values make ties frequent; -inf and NaN included),
    backward bit-equal for the average pool and to rounding of <= 4-term sums for the max pool; odd sizes, strided rows."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(n * 131 + c + h)
    full = _cl((torch.randn(n, c + prefix, h, w, generator=g) * 2).round().div(2).to(dev))      # many equal values
    if full.numel() > 64:
        full[0, 0, 0, 0] = float("-inf")
        full[-1, 1, 1, 0] = float("nan")
    for name, mine, theirs in (("max", ops.maxpool3s2_nhwc, lambda t: F.max_pool2d(t, 3, 2, 1)),
                               ("avg", ops.avgpool2_nhwc, lambda t: F.avg_pool2d(t, 2, 2))):
        if name == "avg" and (h < 2 or w < 2):
            continue
        x = full[:, :c].detach().requires_grad_(True)
        xr = full[:, :c].detach().clone().requires_grad_(True)
        y, yr = mine(x), theirs(xr)
        assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(torch.nan_to_num(y, nan=123.0), torch.nan_to_num(yr, nan=123.0)), name
        assert torch.equal(y.isnan(), yr.isnan())
        with torch.no_grad():                                      # the inference path (the max pool: no tap record)
            yi = mine(full[:, :c])
        assert torch.equal(torch.nan_to_num(yi, nan=123.0), torch.nan_to_num(yr.detach(), nan=123.0)), name
        t = _cl(torch.randn(y.shape, generator=g).to(dev))
        y.backward(t); yr.backward(t)
        if name == "avg":
            assert torch.equal(x.grad, xr.grad)
        else:
            assert (x.grad - xr.grad).abs().max().item() <= 1e-6 * float(t.abs().max()) * 4, name


def test_bn_nhwc_statistics_on_awkward_channels(dev):
    """Constant channels, an offset 1e3 with spread 1e-2, a rounding-sized ripple: the chunk moments (shifted sums merged
    with Chan's formula) keep the variance where E[x^2] - E[x]^2 loses it."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 8, 24, 24, generator=g)
    x[:, 0] = 3.25
    x[:, 1] = 1000.0 + 0.01 * torch.randn(8, 24, 24, generator=g)
    x[:, 2] = 7.0 + 1e-6 * torch.randn(8, 24, 24, generator=g)
    x[:, 3] = 1e-3 * torch.randn(8, 24, 24, generator=g) - 50.0
    xd = _cl(x.to(dev))
    bn = torch.nn.BatchNorm2d(8).to(dev).train()
    y = ops.bn_relu_train_nhwc(xd, bn, False)
    ref = F.batch_norm(x.double(), None, None, bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu(), True, 0.1, bn.eps)
    # the 1e3-offset channel: an fp32 mean near 1000 is known to 3e-5, i.e. 3e-3 of the channel's spread of 1e-2
    assert (y.detach().cpu().double() - ref).abs().max().item() < 6e-3
    # (channels 2 and 3: spreads of 1e-6 / 1e-3 under eps = 1e-5, so 1 / sqrt(var + eps) ~ 300 multiplies the half-ulp of an
    #  fp32 mean near 7 / 50)
    assert (y.detach().cpu().double()[:, [2, 3]] - ref[:, [2, 3]]).abs().max().item() < 1e-3
    assert (y.detach().cpu().double()[:, [0, 4, 5, 6, 7]] - ref[:, [0, 4, 5, 6, 7]]).abs().max().item() < 2e-5
    var = x.double().var((0, 2, 3), unbiased=True)
    assert ((bn.running_var.cpu().double() - (0.9 + 0.1 * var)).abs() / (0.9 + 0.1 * var)).max().item() < 1e-4


def test_densenet_training_step_nhwc_against_fp64(dev):
    """DenseNet-121's training forward + backward on the channels-last path (own convolutions: forward, data and weight
    gradient; own BatchNorm + ReLU) against the same network in fp64 (torch, NCHW) -- and it is at least as close to it
    as the NCHW path (MIOpen convolutions) is.  96x96 inputs, batch 6: 3x3 maps at the end, well-conditioned enough."""
    import copy
    import nwhead_amd.model.backbones as BB
    from nwhead_amd.model import load_model
    torch.manual_seed(0)
    net = load_model("densenet121").to(dev).train()
    x = torch.randn(6, 3, 96, 96, device=dev)
    t = torch.randn(6, net.num_features, device=dev)

    def run(model, xx, tt, nhwc):
        old = BB.NHWC_TRAINING
        BB.NHWC_TRAINING = nhwc
        try:
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.reset_running_stats()
            model.zero_grad(set_to_none=True)
            out = model(xx)
            (out * tt).sum().backward()
        finally:
            BB.NHWC_TRAINING = old
        return out.detach().double(), torch.cat([p.grad.detach().double().flatten() for p in model.parameters()])

    net64 = copy.deepcopy(net).double()
    old = BB.FUSED_BN_RELU_TRAINING
    BB.FUSED_BN_RELU_TRAINING = False
    try:
        o64, g64 = run(net64, x.double(), t.double(), False)
    finally:
        BB.FUSED_BN_RELU_TRAINING = old
    o0, g0 = run(net, x, t, False)
    o1, g1 = run(net, x, t, True)
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    e0 = ((o0 - o64).abs().max() / o64.abs().max()).item()
    e1 = ((o1 - o64).abs().max() / o64.abs().max()).item()
    assert e1 < 1e-4 and e1 < 3 * e0 + 1e-6, (e0, e1)
    c0, c1 = cos(g0, g64), cos(g1, g64)
    assert c1 > 0.9999 and (1 - c1) < 3 * (1 - c0) + 1e-7, (c0, c1)


@pytest.mark.parametrize("arch,size,batch", [("resnet18", 96, 6), ("resnet50", 96, 6), ("CIFAR_ResNet18", 32, 16), ("CIFAR_DenseNet121", 32, 8)])
def test_resnet_training_step_nhwc_against_fp64(dev, arch, size, batch):
    """The ResNets' (ImageNet-style and the CIFAR pre-activation one) training forward + backward on the channels-last path (round 4: own convolutions incl. the
    strided 3x3 / 2 and 1x1 / 2 data and weight gradients and the 7x7 / 2 stem, own NHWC BatchNorm, own max pool;
    model/resnet.py:31-108, :136-207) against the same network in fp64 -- and at least as close to it as the NCHW path
    (MIOpen convolutions + bnrelu.hip) is.  The parameter gradients include every BatchNorm's and every projection's."""
    import copy
    import nwhead_amd.model.backbones as BB
    from nwhead_amd.model import load_model
    from tests.procedural import fill_procedural_hash
    torch.manual_seed(0)
    net = load_model(arch)
    fill_procedural_hash(net)
    net = net.to(dev).train()
    x = torch.randn(batch, 3, size, size, device=dev)
    out_dim = net(x[:2]).shape[1]
    t = torch.randn(batch, out_dim, device=dev)

    def run(model, xx, tt, nhwc):
        old = BB.RESNET_NHWC_TRAINING, BB.CIFAR_DENSENET_NHWC_TRAINING
        BB.RESNET_NHWC_TRAINING = BB.CIFAR_DENSENET_NHWC_TRAINING = nhwc
        try:
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.reset_running_stats()
            model.zero_grad(set_to_none=True)
            out = model(xx)
            (out * tt).sum().backward()
        finally:
            BB.RESNET_NHWC_TRAINING, BB.CIFAR_DENSENET_NHWC_TRAINING = old
        stats = torch.cat([b.detach().double().flatten() for k, b in model.named_buffers() if "running" in k])
        return out.detach().double(), torch.cat([p.grad.detach().double().flatten() for p in model.parameters()]), stats

    net64 = copy.deepcopy(net).double()
    old = BB.FUSED_BN_RELU_TRAINING
    BB.FUSED_BN_RELU_TRAINING = False
    try:
        o64, g64, s64 = run(net64, x.double(), t.double(), False)
    finally:
        BB.FUSED_BN_RELU_TRAINING = old
    o0, g0, s0 = run(net, x, t, False)
    o1, g1, s1 = run(net, x, t, True)
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    e0 = ((o0 - o64).abs().max() / o64.abs().max()).item()
    e1 = ((o1 - o64).abs().max() / o64.abs().max()).item()
    assert e1 < 1e-4 and e1 < 3 * e0 + 1e-6, (e0, e1)
    # Gradients: a SINGLE ReLU whose pre-activation is zero to fp32 rounding and falls the other way than in fp64 moves every
    # gradient upstream of it by ~2e-3 of its norm (measured on this input: one flip among layer4.1.bn1's 27 648 activations,
    # tools/bn_bwd_diag.py; every kernel of the path agrees with fp64 of its OWN inputs to 1e-6, tools/dgrad_diag.py) -- which
    # path catches such a flip is chance, so the bar is absolute, not relative to the NCHW path.
    c0, c1 = cos(g0, g64), cos(g1, g64)
    assert c1 > 0.9999, (c0, c1)
    m0, m1 = ((g0 - g64).abs().max() / g64.abs().max()).item(), ((g1 - g64).abs().max() / g64.abs().max()).item()
    assert m1 < max(1e-2, 5 * m0), (m0, m1)
    assert ((s1 - s64).abs().max() / s64.abs().max()).item() < 1e-4


def test_training_forward_sees_a_fused_optimizer_step(dev):
    """torch's fused optimizers update parameters without advancing their version counters: the channels-last training
    forward must rebuild its weight operands anyway (ConvWeightBank.refresh(force=True)).  After a large fused SGD step the
    channels-last forward agrees with the NCHW / MIOpen forward of the same, updated weights -- and not with the old ones."""
    import nwhead_amd.model.backbones as BB
    from nwhead_amd.model import load_model
    torch.manual_seed(1)
    net = load_model("densenet121").to(dev).train()
    x = torch.randn(4, 3, 64, 64, device=dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.5, fused=True)

    def fwd(nhwc):
        old = BB.NHWC_TRAINING
        BB.NHWC_TRAINING = nhwc
        try:
            return net(x)
        finally:
            BB.NHWC_TRAINING = old

    before = fwd(True)
    before.square().mean().backward()
    versions = [p._version for p in net.parameters()]
    opt.step()
    if [p._version for p in net.parameters()] != versions:
        pytest.skip("this torch build's fused SGD advances the version counters")
    after, ref = fwd(True).detach(), fwd(False).detach()
    sc = float(ref.abs().max())
    assert (after - ref).abs().max().item() < 1e-4 * sc
    assert (before.detach() - ref).abs().max().item() > 1e-2 * sc     # the step did move the features


MOMENT_CASES = [(3, 64, 14, 14, 128, 1, 96, 0), (2, 128, 28, 28, 32, 3, 0, 96), (2, 32, 14, 14, 128, 3, 160, 0),
                (5, 128, 7, 7, 32, 3, 0, 864), (2, 128, 56, 56, 128, 1, 0, 0), (9, 96, 56, 56, 128, 1, 0, 0),
                (3, 64, 28, 28, 256, 1, 0, 0)]      # (the last: two output-channel tiles, moments per tile)


@pytest.mark.parametrize("n,cin,h,w,cout,k,xw,yw", MOMENT_CASES)
def test_conv_channel_windows_and_moments(dev, n, cin, h, w, cout, k, xw, yw):
    """nw_conv2d_nhwc_f16x2 with ldx / ldy: the input is a channel window of a wider channels-last tensor, the output is
    written into a window of another (the rest of which must stay untouched); `moments`: the groups it leaves merge
    (nw_bn_nhwc_moments_from_partials_f32) to the batch mean / variance of y, against fp64."""
    from nwhead_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout + xw)
    pad = k // 2
    wide_x = _cl((torch.randn(n, cin + xw, h, w, generator=g) * 1.3 + 0.2).to(dev))
    x = wide_x[:, 32:32 + cin] if xw else wide_x
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    sw = ops.SplitConvWeight(wt)
    wide_y = _cl(torch.full((n, cout + yw, h, w), 7.5, device=dev))
    c_off = 64 if yw else 0
    am = ops.absmax(wide_x)
    G = lib.nw_conv2d_nhwc_moments_groups(n, h, w, cin, cout, k, k, 1, pad)
    assert G > 0
    part = torch.empty(5 * G * cout, dtype=torch.float32, device=dev)     # count, mean, M2, minimum, maximum per group
    am_out = torch.empty(ops.AMAX_SLOTS, dtype=torch.float32, device=dev)
    st = ops._stream(wide_x)
    _lib.check(lib.nw_conv2d_nhwc_f16x2(x.data_ptr(), am.data_ptr(), sw.split.data_ptr(), sw.scale.data_ptr(), None, None, 0,
                                        wide_y.data_ptr() + 4 * c_off, am_out.data_ptr(), n, h, w, cin, cout, k, k, 1, pad,
                                        cin + xw, cout + yw, part.data_ptr(), st), "conv")
    stats = torch.empty(3 * cout, dtype=torch.float32, device=dev)
    _lib.check(lib.nw_bn_nhwc_moments_from_partials_f32(part.data_ptr(), G, cout, 1e-5, stats.data_ptr(),
                                                        stats.data_ptr() + 4 * cout, stats.data_ptr() + 8 * cout, st), "merge")
    ref = F.conv2d(x.double(), wt.double(), None, 1, pad)
    y = wide_y[:, c_off:c_off + cout]
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() < TOL
    if yw:
        assert float(wide_y[:, :c_off].min()) == 7.5 and float(wide_y[:, c_off + cout:].max()) == 7.5
    assert abs(float(am_out.max()) - float(y.abs().max())) <= 1e-6 * float(y.abs().max())
    mean, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
    sd = var.sqrt()
    assert ((stats[:cout].double() - mean).abs() / sd).max().item() < 1e-5
    assert ((stats[2 * cout:].double() - var).abs() / var).max().item() < 1e-5
    assert ((stats[cout:2 * cout].double() - (var + 1e-5).rsqrt()).abs() * sd).max().item() < 1e-5
    # round 4: the same groups with their minima and maxima (the values the kernel itself stored), statistics-only merge
    st5 = torch.empty(5 * cout, dtype=torch.float32, device=dev)
    _lib.check(lib.nw_bn_nhwc_prep_from_partials_f32(part.data_ptr(), G, cout, 1e-5, *(st5.data_ptr() + 4 * cout * j for j in range(5)),
                                                     None, None, None, None, None, 0.0, 1, None, None, st), "merge5")
    np.testing.assert_allclose(st5[:3 * cout].cpu().numpy(), stats.cpu().numpy(), rtol=2e-5, atol=1e-7)
    assert torch.equal(st5[3 * cout:4 * cout], y.amin((0, 2, 3))) and torch.equal(st5[4 * cout:], y.amax((0, 2, 3)))


BNRELU_CASES = [
    # n, cin, h, w, cout, k, wide (channels of the tensor x is a prefix of; 0: dense), moments
    (2, 64, 12, 12, 128, 1, 96, True),       # GATHER, 64 x 64 tiles, a channel prefix
    (3, 160, 28, 28, 128, 1, 0, True),       # GATHER, 128-pixel tiles
    (9, 96, 56, 56, 128, 1, 0, True),        # GATHER, 128 x 128 tiles, moments merged per workgroup
    (3, 992, 7, 7, 128, 1, 1024, False),     # DenseNet block 4 conv1: 31 chunks, ragged tile
    (16, 64, 56, 56, 32, 1, 0, False),       # 256-pixel tiles of 32 channels (the shallower ring)
    (2, 128, 14, 14, 32, 3, 0, True),        # PATCH: padding must stay ZERO behind the BatchNorm
    (3, 128, 9, 7, 32, 3, 0, True),          # PATCH, tiles crossing rows and images
    (3, 128, 56, 56, 32, 3, 0, False),       # DenseNet conv2 at 56 x 56 (256-pixel tiles)
    (5, 128, 7, 7, 32, 3, 0, True),
    (42, 96, 14, 14, 128, 1, 128, True),     # 64 pixels x 128 channels (one round of 129 tiles), moments merged per workgroup
    (42, 32, 14, 14, 128, 3, 0, True),       # ... in PATCH mode
]


@pytest.mark.parametrize("n,cin,h,w,cout,k,wide,moments", BNRELU_CASES)
def test_conv_with_batchnorm_relu_in_the_loaders(dev, n, cin, h, w, cout, k, wide, moments):
    """nw_conv2d_nhwc_bnrelu_f16x2 (round 4, model/densenet.py:36-45 without t1 / t2): y = conv(relu((x - mean) a + beta)),
    the table and the exact bound from nw_bn_nhwc_moments_minmax_f32 + nw_bn_nhwc_prep_f32, against fp64 -- channels with
    an offset, negative gamma, a constant channel; zero padding of the TRANSFORMED map in 3x3 mode; the moments it leaves;
    and nw_conv2d_nhwc_wgrad_batch_f16x2 with the same table on its x operand."""
    from nwhead_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(7 * cin + cout + k)
    pad = k // 2
    ctot = wide if wide else cin
    xw = torch.randn(n, ctot, h, w, generator=g) * (0.2 + 2.0 * torch.rand(1, ctot, 1, 1, generator=g)) + 3.0 * torch.randn(1, ctot, 1, 1, generator=g)
    xw[:, 1] = 0.75                                                   # a constant channel (variance 0)
    xw = _cl(xw.to(dev))
    x = xw[:, :cin]
    gamma = (torch.rand(cin, generator=g) + 0.5) * torch.where(torch.rand(cin, generator=g) < 0.2, -1.0, 1.0)
    beta = torch.randn(cin, generator=g) * 0.5
    gamma, beta = gamma.to(dev), beta.to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    sw = ops.SplitConvWeight(wt)
    rows, eps, st = n * h * w, 1e-5, ops._stream(xw)
    f32 = dict(dtype=torch.float32, device=dev)
    stats = torch.empty(5 * cin, **f32)
    sp = [stats.data_ptr() + 4 * cin * j for j in range(5)]
    wsb = lib.nw_bn_nhwc_minmax_workspace_bytes(rows, cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.nw_bn_nhwc_moments_minmax_f32(x.data_ptr(), ctot, rows, cin, eps, *sp, ws.data_ptr(), wsb, st), "moments")
    xd = x.double()
    mean, var = xd.mean((0, 2, 3)), xd.var((0, 2, 3), unbiased=False)
    assert torch.equal(stats[3 * cin:4 * cin], x.amin((0, 2, 3))) and torch.equal(stats[4 * cin:], x.amax((0, 2, 3)))
    rm, rv = torch.zeros(cin, **f32), torch.ones(cin, **f32)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    tab, am = torch.empty(3 * cin, **f32), torch.empty(ops.AMAX_SLOTS, **f32)
    _lib.check(lib.nw_bn_nhwc_prep_f32(*sp, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1,
                                       rows, cin, tab.data_ptr(), am.data_ptr(), st), "prep")
    t_ref = torch.relu((xd - mean[None, :, None, None]) * (gamma.double() * (var + eps).rsqrt())[None, :, None, None]
                       + beta.double()[None, :, None, None])
    # the bound is the maximum of the transformed tensor as the loaders compute it (fp32): never below it, and tight
    t32 = torch.relu((x - stats[:cin][None, :, None, None]) * tab[cin:2 * cin][None, :, None, None] + tab[2 * cin:][None, :, None, None])
    assert float(t32.max()) * (1 - 1e-6) <= float(am.max()) <= float(t32.max()) * (1 + 1e-5) + 1e-30
    assert int(nbt) == 1
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * mean.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 + 0.1 * (var * rows / (rows - 1)).float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    y = _cl(torch.empty(n, cout, h, w, **f32))
    G = lib.nw_conv2d_nhwc_moments_groups(n, h, w, cin, cout, k, k, 1, pad)
    part = torch.empty(5 * G * cout, **f32) if moments else None
    am_out = torch.empty(ops.AMAX_SLOTS, **f32)
    _lib.check(lib.nw_conv2d_nhwc_bnrelu_f16x2(x.data_ptr(), tab.data_ptr(), am.data_ptr(), 0, sw.split.data_ptr(), sw.scale.data_ptr(),
                                               None, 0, y.data_ptr(), am_out.data_ptr(), n, h, w, cin, cout, k, k, 1, pad, ctot, 0,
                                               None if part is None else part.data_ptr(), st), "conv bnrelu")
    ref = F.conv2d(t_ref, wt.double(), None, 1, pad)
    sc = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() / sc < 2 * TOL
    assert abs(float(am_out.max()) - float(y.abs().max())) <= 1e-6 * float(y.abs().max())
    if moments:
        st5 = torch.empty(5 * cout, **f32)
        tab2, am2 = torch.empty(3 * cout, **f32), torch.empty(ops.AMAX_SLOTS, **f32)
        g2, b2 = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
        _lib.check(lib.nw_bn_nhwc_prep_from_partials_f32(part.data_ptr(), G, cout, eps, *(st5.data_ptr() + 4 * cout * j for j in range(5)),
                                                         g2.data_ptr(), b2.data_ptr(), None, None, None, 0.0, 1, tab2.data_ptr(),
                                                         am2.data_ptr(), st), "merge + prep")
        m_y, v_y = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
        assert ((st5[:cout].double() - m_y).abs() / v_y.sqrt()).max().item() < 2e-5
        assert ((st5[2 * cout:3 * cout].double() - v_y).abs() / v_y).max().item() < 2e-5
        assert torch.equal(st5[3 * cout:4 * cout], y.amin((0, 2, 3))) and torch.equal(st5[4 * cout:], y.amax((0, 2, 3)))
        t2 = torch.relu((y - st5[:cout][None, :, None, None]) * tab2[cout:2 * cout][None, :, None, None] + tab2[2 * cout:][None, :, None, None])
        assert float(t2.max()) * (1 - 1e-6) <= float(am2.max()) <= float(t2.max()) * (1 + 1e-5) + 1e-30
        np.testing.assert_allclose(tab2[cout:2 * cout].cpu().numpy(), (g2.double() * (v_y + eps).rsqrt()).float().cpu().numpy(), rtol=1e-4)
    # the weight gradient with the same table on its x operand: dW = sum_pixels relu(bn(x)) (x) gy
    if lib.nw_conv2d_nhwc_wgrad_supported(n, h, w, cin, cout, k, k, 1, pad):
        gy = _cl((torch.randn(n, cout, h, w, generator=g) * 0.7).to(dev))
        am_g = ops.absmax(gy)
        dw = torch.empty((cout, k, k, cin), **f32)
        job = _lib.WgradJob(x.data_ptr(), am.data_ptr(), gy.data_ptr(), am_g.data_ptr(), dw.data_ptr(), n, h, w, cin, cout, k, k, 1, pad,
                            ctot, 0, 0, tab.data_ptr())
        jobs = (_lib.WgradJob * 1)(job)
        wb = lib.nw_conv2d_nhwc_wgrad_batch_workspace_bytes(jobs, 1)
        wsw = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.nw_conv2d_nhwc_wgrad_batch_f16x2(jobs, 1, wsw.data_ptr(), wb, st), "wgrad")
        dref = torch.nn.grad.conv2d_weight(t_ref, wt.shape, gy.double(), 1, pad).permute(0, 2, 3, 1)
        assert (dw.double() - dref).abs().max().item() / dref.abs().max().item() < 2 * TOL


def test_bn_forward_in_phases_equals_the_fused_call(dev):
    """nw_bn_nhwc_moments_f32 + nw_bn_relu_nhwc_apply_f32 against nw_bn_relu_nhwc_train_fwd_f32 on a channel prefix: same
    outputs, saved statistics, running statistics and step counter."""
    from nwhead_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    n, c, h, w, wide = 3, 96, 9, 11, 160
    full = _cl((torch.randn(n, wide, h, w, generator=g) * 1.7 - 0.4).to(dev))
    x = full[:, :c]
    rows = n * h * w
    bn_a, bn_b = torch.nn.BatchNorm2d(c).to(dev).train(), torch.nn.BatchNorm2d(c).to(dev).train()
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn_a.bias.copy_(torch.randn(c, generator=g) * 0.3)
    bn_b.load_state_dict(bn_a.state_dict())
    y_a = ops.bn_relu_train_nhwc(x, bn_a, True)
    st = ops._stream(full)
    stats = torch.empty(3 * c, dtype=torch.float32, device=dev)
    wsb = lib.nw_bn_nhwc_workspace_bytes(rows, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.nw_bn_nhwc_moments_f32(x.data_ptr(), wide, rows, c, float(bn_b.eps), stats.data_ptr(), stats.data_ptr() + 4 * c,
                                          stats.data_ptr() + 8 * c, ws.data_ptr(), wsb, st), "moments")
    y_b = torch.empty(rows, c, dtype=torch.float32, device=dev)
    am = torch.empty(ops.AMAX_SLOTS, dtype=torch.float32, device=dev)
    _lib.check(lib.nw_bn_relu_nhwc_apply_f32(x.data_ptr(), wide, stats.data_ptr(), stats.data_ptr() + 4 * c, stats.data_ptr() + 8 * c,
                                             bn_b.weight.data_ptr(), bn_b.bias.data_ptr(), bn_b.running_mean.data_ptr(),
                                             bn_b.running_var.data_ptr(), bn_b.num_batches_tracked.data_ptr(), float(bn_b.momentum),
                                             y_b.data_ptr(), am.data_ptr(), rows, c, 1, st), "apply")
    assert torch.equal(y_b.view(n, h, w, c).permute(0, 3, 1, 2), y_a.detach())
    assert float(am.max()) == float(y_a.detach().abs().max())
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-6, atol=1e-7)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-6, atol=1e-7)
    assert int(bn_b.num_batches_tracked) == 1


@pytest.mark.parametrize("bwd_stats_in_dgrad", [False, True, "fused_norm1"])
def test_dense_block_slab_node_against_layer_by_layer(dev, bwd_stats_in_dgrad):
    """ops.dense_block_nhwc_train (one autograd node over one slab, statistics shared between the layers, moments from the
    convolutions' epilogues, the gradient slab accumulated in place) against the layer-by-layer channels-last path and
    fp64 torch: outputs, input gradient, every parameter gradient, running statistics."""
    import copy
    import nwhead_amd.model.backbones as BB
    from nwhead_amd import ops
    torch.manual_seed(3)
    block = BB._DenseBlock(4, 64, 4, 32, 0.0).to(dev).train()
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    x0 = _cl(torch.randn(5, 64, 14, 14, device=dev) * 1.5 + 0.3)
    t = _cl(torch.randn(5, 64 + 4 * 32, 14, 14, device=dev))
    convs = [(m.weight, True) for m in block.modules() if isinstance(m, torch.nn.Conv2d)]

    def run(slab):
        blk = copy.deepcopy(block)
        bank = ops.ConvWeightBank([(m.weight, True) for m in blk.modules() if isinstance(m, torch.nn.Conv2d)])
        bank.refresh()
        old, old_f, old_n = BB.DENSE_SLAB, ops.DENSE_BWD_STATS_IN_DGRAD, ops.DENSE_FUSED_NORM1_BWD
        BB.DENSE_SLAB = slab
        ops.DENSE_BWD_STATS_IN_DGRAD = bwd_stats_in_dgrad is True      # (BatchNorm's backward sums from the data gradients' epilogues)
        ops.DENSE_FUSED_NORM1_BWD = bwd_stats_in_dgrad == "fused_norm1"   # (norm1 -> relu1 -> conv1 backward in nw_bn_dgrad1x1_bwd_f16x2)
        try:
            x = x0.clone().requires_grad_(True)
            if slab:
                assert ops.dense_block_nhwc_supported(x, list(blk.children()), bank)
            y = blk.forward_nhwc_train(x, bank)
            (y * t).sum().backward()
        finally:
            BB.DENSE_SLAB, ops.DENSE_BWD_STATS_IN_DGRAD, ops.DENSE_FUSED_NORM1_BWD = old, old_f, old_n
        return y.detach().double(), x.grad.double(), {k: p.grad.double() for k, p in blk.named_parameters()}, \
            {k: b.double() for k, b in blk.named_buffers() if "running" in k}

    ya, gxa, gpa, rsa = run(False)
    yb, gxb, gpb, rsb = run(True)
    b64 = copy.deepcopy(block).double()
    x64 = x0.double().contiguous().requires_grad_(True)
    feats = [x64]
    for layer in b64.children():
        inp = torch.cat(feats, 1)
        feats.append(layer.conv2(F.relu(layer.norm2(layer.conv1(F.relu(layer.norm1(inp)))))))
    y64 = torch.cat(feats, 1)
    (y64 * t.double()).sum().backward()
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert rel(yb, y64) < 2e-5 and rel(yb, y64) < 3 * rel(ya, y64) + 1e-6
    assert rel(gxb, x64.grad) < 2e-4 and rel(gxb, x64.grad) < 3 * rel(gxa, x64.grad) + 1e-5
    g64 = {k: p.grad for k, p in b64.named_parameters()}
    for k in g64:
        assert rel(gpb[k], g64[k]) < 3e-4 and rel(gpb[k], g64[k]) < 3 * rel(gpa[k], g64[k]) + 2e-5, k
    for k in rsa:
        assert rel(rsb[k], rsa[k]) < 1e-5, k


@pytest.mark.parametrize("n,c0,h,w,L", [(3, 96, 9, 11, 3), (1, 32, 5, 7, 2), (2, 160, 30, 17, 2), (7, 64, 3, 3, 5)])
def test_dense_block_slab_node_on_ragged_shapes(dev, n, c0, h, w, L):
    """The slab node on shapes whose pixel counts are no multiple of any tile (ragged last tiles and groups, one image,
    tiny maps): input gradient and parameter gradients against the layer-by-layer channels-last path."""
    import copy
    import nwhead_amd.model.backbones as BB
    from nwhead_amd import ops
    torch.manual_seed(n + c0 + h)
    block = BB._DenseBlock(L, c0, 4, 32, 0.0).to(dev).train()
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    x0 = _cl(torch.randn(n, c0, h, w, device=dev) + 0.2)
    t = _cl(torch.randn(n, c0 + L * 32, h, w, device=dev))
    outs = []
    for slab in (False, True):
        blk = copy.deepcopy(block)
        bank = ops.ConvWeightBank([(m.weight, True) for m in blk.modules() if isinstance(m, torch.nn.Conv2d)])
        bank.refresh()
        old = BB.DENSE_SLAB
        BB.DENSE_SLAB = slab
        try:
            x = x0.clone().requires_grad_(True)
            if slab and not ops.dense_block_nhwc_supported(x, list(blk.children()), bank):
                pytest.skip("shape not served by the slab node")
            y = blk.forward_nhwc_train(x, bank)
            (y * t).sum().backward()
        finally:
            BB.DENSE_SLAB = old
        outs.append((y.detach().double(), x.grad.double(), torch.cat([p.grad.double().flatten() for p in blk.parameters()])))
    (ya, gxa, gpa), (yb, gxb, gpb) = outs
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert rel(yb, ya) < 2e-5
    assert rel(gxb, gxa) < 5e-4
    assert float((gpb * gpa).sum() / (gpb.norm() * gpa.norm())) > 0.99999


@pytest.mark.parametrize("folding", [False, True])
def test_k2_resnet18_plus_head_end_to_end(dev, folding):
    """BASELINE configs[1] end to end: load_model('resnet18') @224, 64 queries, a bank of N = 1000
    supports in C = 200 classes through NWNet.precompute() + predict('full') -- plain, and with enable_bn_folding (the
    channels-last copy whose every convolution runs in nw_conv2d_nhwc_f16x2) -- against the same network on the host with
    the oracle head (reference call order: train.py:290-297, nw.py:118-160).  Tolerance: an fp64 run of the network sits
    ~2e-6 (relative, features) from the host's fp32 run; the log-probabilities see that through distances of ~30."""
    from oracle import nw_oracle as O
    from nwhead_amd.model import load_model
    from nwhead_amd.nwhead.nw import NWNet
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(2)
    n_sup, C = 1000, 200
    sup_x = torch.randn(n_sup, 3, 64, 64, generator=g)       # the bank's images (64x64: 1000 images @224 would be 600 MB)
    sup_y = torch.arange(n_sup) % C

    class DS(torch.utils.data.Dataset):
        targets = sup_y.tolist()

        def __len__(self):
            return n_sup

        def __getitem__(self, i):
            return sup_x[i], int(sup_y[i])

    net = NWNet(load_model("resnet18"), C, support_dataset=DS(), feat_dim=512, n_shot_full=5, device="cuda:0").to(dev).eval()
    if folding:
        net.enable_bn_folding(True)
    host = load_model("resnet18").eval()
    host.load_state_dict(net.featurizer.state_dict())
    xq = torch.randn(64, 3, 224, 224, generator=g)
    with torch.no_grad():
        net.precompute()
        out = net.predict(xq.to(dev), mode="full")
        torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
        fq = host(xq)
        order = np.argsort(np.asarray(sup_y), kind="stable")   # the balanced, class-sorted bank of precompute()
        fs = host(sup_x[order])
        ref = O.nw_head_f64(fq, fs, sup_y[order], C)
    assert out.shape == (64, C) and net.full_feat.shape == (n_sup, 512)
    assert torch.equal(net.full_y.cpu(), sup_y[order])
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-3, err
    assert (out.argmax(-1).cpu() == ref.argmax(-1)).float().mean().item() > 0.98


def test_densenet121_inference_on_the_channels_last_kernels(dev):
    """DenseNet.forward in eval mode on the device (round 4: stem, pools, two launches per dense layer with the BatchNorms
    inside the convolutions, transitions) against the fp64 network on the host, with trained-looking BatchNorm statistics
    (channel scales over two decades, offsets); the folded copy NWNet builds is the same path; a weight update rebuilds the plan."""
    from nwhead_amd.model import load_model, fold_batchnorm
    torch.manual_seed(3)
    net = load_model("densenet121").eval()
    g = torch.Generator().manual_seed(5)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            c = m.num_features
            m.running_mean.copy_(torch.randn(c, generator=g) * 0.3)
            m.running_var.copy_(torch.exp(torch.randn(c, generator=g) * 1.0))
            m.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(c, generator=g) * 0.2)
    x = torch.randn(5, 3, 96, 96, generator=g)
    with torch.no_grad():
        ref = net.double()(x.double())
    net = net.float().to(dev)
    with torch.no_grad():
        y = net(x.to(dev))
        assert getattr(net, "_nw_infer_plan", None) is not None          # the channels-last path ran
        yf = fold_batchnorm(net)(x.to(dev))
        sc = ref.abs().max().item()
        assert (y.cpu().double() - ref).abs().max().item() < 2e-5 * sc
        assert torch.equal(y, yf)
        net.features.denseblock2.denselayer3.conv1.weight.mul_(0.5)       # in place: the plan follows
        y2 = net(x.to(dev))
        ref2 = net.cpu().double()(x.double())
    assert (y2.cpu().double() - ref2).abs().max().item() < 2e-5 * ref2.abs().max().item()
    assert not torch.allclose(y2, y)


@pytest.mark.parametrize("arch,batch", [("CIFAR_ResNet18", 16), ("CIFAR_ResNet10", 5)])
def test_cifar_resnet_inference_on_the_channels_last_kernels(dev, arch, batch):
    """CIFAR_ResNet._forward_nhwc_infer (round 4; model/resnet.py:111-134, :209-239): the eval-mode pre-activation ResNet on the split-fp16
    kernels -- stem with bn1 folded, relu(bn1(.)) per block in one pass, conv1 with bn2 folded, conv2 with the shortcut in its
    store -- against the fp32 CPU network with the same parameters and running statistics; no torch convolution is called."""
    import copy
    from nwhead_amd.model import load_model
    from tests.procedural import fill_procedural_hash
    net = load_model(arch)
    fill_procedural_hash(net)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        net.train()
        net(torch.randn(8, 3, 32, 32, generator=g))            # running statistics off their init
        net.eval()
        x = torch.randn(batch, 3, 32, 32, generator=g)
        want = net(x)
        dnet = copy.deepcopy(net).to(dev).eval()
        calls = []
        orig = F.conv2d
        F.conv2d = lambda *a, **k: calls.append(1) or orig(*a, **k)
        try:
            got = dnet(x.to(dev))
        finally:
            F.conv2d = orig
    assert getattr(dnet, "_nw_infer_plan", None) is not None and not calls
    assert (got.cpu() - want).abs().max().item() < 1e-4 * want.abs().max().item()


@pytest.mark.parametrize("n,h,w,room", [(3, 64, 64, 0), (2, 224, 224, 0), (2, 37, 53, 0), (1, 9, 7, 0), (5, 96, 80, 32), (1, 8, 201, 0)])
def test_fused_stem_against_convolution_plus_pool(dev, n, h, w, room):
    """nw_stem7x7s2_relu_maxpool_f16x2 (round 4; model/resnet.py:147, :200-203, model/densenet.py:114-120 with the BatchNorm folded in):
    conv7x7/2 + bias + ReLU + maxpool3x3/2 in one kernel against the two kernels it replaces (same arithmetic: equal to fp32
    rounding) and against fp64 torch; odd sizes, maps smaller than a tile, a result written into a wider slab, a NaN pixel."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(n + h + w)
    x = (torch.randn(n, 3, h, w, generator=g) * 1.3 + 0.2).to(dev)
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).to(dev)
    b = (torch.randn(64, generator=g) * 0.3).to(dev)
    sw = ops.SplitConvWeight(wt)
    got = ops.stem_conv_relu_maxpool_nhwc(x, sw, b, room)
    two = ops.maxpool3s2_nhwc(ops.conv2d_nhwc(x, sw, b, None, True, 2, 3), room)
    ref = F.max_pool2d(F.relu(F.conv2d(x.double(), wt.double(), b.double(), 2, 3)), 3, 2, 1)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last) == two.is_contiguous(memory_format=torch.channels_last)
    assert (got.double() - ref).abs().max().item() < TOL * ref.abs().max().item()
    assert (got - two).abs().max().item() <= 1e-6 * float(two.abs().max())
    assert float(got.nw_amax.max()) == float(got.abs().max())
    if room:
        assert got.nw_slab.shape[1] == 64 + room
    x2 = x.clone()
    x2[0, 1, h // 2, w // 2] = float("nan")
    a, c = ops.stem_conv_relu_maxpool_nhwc(x2, sw, b), ops.maxpool3s2_nhwc(ops.conv2d_nhwc(x2, sw, b, None, True, 2, 3))
    # the NaN reaches exactly the pooled pixels whose windows hold a convolution output that read the pixel (kx = 7 of the padded
    # 8-pixel run does not exist: 0 x NaN there must not count; torch's own kernels spread a NaN further)
    hit = F.conv2d(torch.isnan(x2).float(), torch.ones(1, 3, 7, 7, device=dev), None, 2, 3) > 0
    want = (F.max_pool2d(hit.float(), 3, 2, 1) > 0).expand(-1, 64, -1, -1)
    assert torch.equal(torch.isnan(a), want) and torch.equal(torch.isnan(c), want)


def test_fused_stem_on_random_shapes(dev):
    """The fused stem against convolution + pool on twenty random image sizes (7 .. 150 pixels a side, 1 .. 4 images): tiles cut
    by the right and bottom edges, maps smaller than one tile, odd convolution and pool sizes."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(11)
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).to(dev)
    b = (torch.randn(64, generator=g) * 0.3).to(dev)
    sw = ops.SplitConvWeight(wt)
    for _ in range(20):
        n = int(torch.randint(1, 5, (1,), generator=g))
        h, w = (int(v) for v in torch.randint(7, 151, (2,), generator=g))
        x = (torch.randn(n, 3, h, w, generator=g) * 2.0).to(dev)
        got = ops.stem_conv_relu_maxpool_nhwc(x, sw, b)
        two = ops.maxpool3s2_nhwc(ops.conv2d_nhwc(x, sw, b, None, True, 2, 3))
        assert got.shape == two.shape, (n, h, w)
        assert (got - two).abs().max().item() <= 1e-6 * max(float(two.abs().max()), 1e-30), (n, h, w)
        assert float(got.nw_amax.max()) == float(got.abs().max())


@pytest.mark.parametrize("n,c,h,w,wide", [(2, 64, 8, 8, 0), (3, 256, 28, 28, 0), (2, 96, 9, 7, 128), (1, 4, 2, 3, 0)])
def test_bn_relu_avgpool2_nhwc_against_torch(dev, n, c, h, w, wide):
    """ops.bn_relu_avgpool2_nhwc (nw_bn_relu_avgpool2x2_nhwc_f32: an eval-mode transition's norm -> relu with the 2 x 2 average pulled
    in front of the 1 x 1 convolution, model/densenet.py:83-91) against torch, odd sizes (the last row / column dropped), a channel
    prefix of a wider tensor, the amax record."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(c + h)
    full = _cl((torch.randn(n, max(wide, c), h, w, generator=g) * 1.5 + 0.4).to(dev))
    x = full[:, :c]
    bn = torch.nn.BatchNorm2d(c).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.5); bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        got = ops.bn_relu_avgpool2_nhwc(x, ops.bn_table(bn))
        ref = F.avg_pool2d(F.relu(bn(x)), 2, 2)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got - ref).abs().max().item() < 2e-6 * float(ref.abs().max())
    assert float(got.nw_amax.max()) == float(got.abs().max())


def test_strided_gradients_on_random_shapes(dev):
    """ops.conv2d_nhwc_train with stride 2 on fifteen random shapes (3x3 / pad 1 and 1x1 / pad 0, 32 .. 160 channels, maps of 5 .. 40
    pixels a side, odd sizes): data gradient (zero-dilated gy) and weight gradient (one 1x1 problem per tap) against fp64 autograd."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(23)
    for _ in range(15):
        n = int(torch.randint(1, 4, (1,), generator=g))
        cin, cout = (32 * int(v) for v in torch.randint(1, 6, (2,), generator=g))
        h, w = (int(v) for v in torch.randint(5, 41, (2,), generator=g))
        k = 3 if float(torch.rand(1, generator=g)) < 0.6 else 1
        pad = k // 2
        x0 = _cl(torch.randn(n, cin, h, w, generator=g).to(dev))
        w0 = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        x64, w64 = x0.double().requires_grad_(True), w0.double().requires_grad_(True)
        y64 = F.conv2d(x64, w64, None, 2, pad)
        t = torch.randn(y64.shape, generator=g).to(dev)
        (y64 * t.double()).sum().backward()
        x, wp = x0.clone().requires_grad_(True), torch.nn.Parameter(w0.clone())
        y = ops.conv2d_nhwc_train(x, wp, 2, pad)
        (y * _cl(t)).sum().backward()
        tag = (n, cin, cout, h, w, k)
        assert (y.double() - y64).abs().max().item() < TOL * y64.abs().max().item(), tag
        assert (x.grad.double() - x64.grad).abs().max().item() < 2e-5 * x64.grad.abs().max().item(), tag
        assert (wp.grad.double() - w64.grad).abs().max().item() < 2e-5 * w64.grad.abs().max().item(), tag


def test_fused_norm1_backward_on_random_shapes(dev):
    """nw_bn_dgrad1x1_bwd_f16x2 on twelve random shapes (rows 1 .. 3000, c = 32 .. 800, k = 32 .. 128, slab wider than c) against the
    two-step path (data-gradient convolution + nw_bn_relu_nhwc_train_bwd_f32)."""
    from nwhead_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(29)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    for _ in range(12):
        rows = int(torch.randint(2, 3001, (1,), generator=g))
        c = 32 * int(torch.randint(1, 26, (1,), generator=g))
        mid = 32 * int(torch.randint(1, 5, (1,), generator=g))
        ctot = c + 32 * int(torch.randint(0, 3, (1,), generator=g))
        slab = (torch.randn(rows, ctot, generator=g) * 1.2 + 0.3).to(dev)
        gamma = ((torch.rand(c, generator=g) + 0.5) * torch.where(torch.rand(c, generator=g) < 0.2, -1.0, 1.0)).to(dev)
        beta = (torch.randn(c, generator=g) * 0.5).to(dev)
        wt = (torch.randn(mid, c, generator=g) / c ** 0.5).to(dev)
        du = torch.randn(rows, mid, generator=g).to(dev)
        G0 = torch.randn(rows, ctot, generator=g).to(dev)
        x = slab[:, :c]
        mean = x.mean(0).contiguous()
        inv = (1.0 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)).contiguous()
        tab = torch.cat([mean, gamma * inv, beta]).contiguous()
        d1 = ops.SplitConvWeight(wt.t().reshape(c, mid, 1, 1).contiguous())
        am_du = ops.absmax(du)
        G, am_g = G0.clone(), torch.empty(ops.AMAX_SLOTS, device=dev)
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        fb = lib.nw_bn_dgrad1x1_workspace_bytes(rows, c)
        ws = torch.empty(fb // 4, device=dev)
        _lib.check(lib.nw_bn_dgrad1x1_bwd_f16x2(P(du), P(am_du), P(d1.split), P(d1.scale), P(slab), ctot, P(tab), c, P(inv), P(G), ctot,
                                                P(am_g), P(dg), P(db), P(ws), fb, rows, c, mid, st), "nw_bn_dgrad1x1_bwd_f16x2")
        dt1, am_t = torch.empty(rows, c, device=dev), torch.empty(ops.AMAX_SLOTS, device=dev)
        _lib.check(lib.nw_conv2d_nhwc_f16x2(P(du), P(am_du), P(d1.split), P(d1.scale), None, None, 0, P(dt1), P(am_t), 1, rows, 1, mid, c, 1,
                                            1, 1, 0, 0, 0, None, st), "nw_conv2d_nhwc_f16x2")
        G2 = G0.clone()
        dg2, db2, am2 = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(ops.AMAX_SLOTS, device=dev)
        bnb = lib.nw_bn_nhwc_workspace_bytes(rows, c)
        ws2 = torch.empty(bnb // 4 + 4, device=dev)
        _lib.check(lib.nw_bn_relu_nhwc_train_bwd_f32(P(slab), ctot, P(dt1), P(gamma), P(beta), P(mean), P(inv), P(G2), P(dg2), P(db2), P(G2),
                                                     ctot, ctot, P(am2), P(ws2), bnb, rows, c, 1, st), "nw_bn_relu_nhwc_train_bwd_f32")
        tag = (rows, c, mid, ctot)
        sc = lambda t_: max(float(t_.abs().max()), 1e-12)
        assert (G[:, :c] - G2[:, :c]).abs().max().item() < 3e-5 * sc(G2[:, :c]), tag
        assert torch.equal(G[:, c:], G0[:, c:]), tag
        assert (dg - dg2).abs().max().item() < 3e-5 * sc(dg2) and (db - db2).abs().max().item() < 3e-5 * sc(db2), tag
        assert float(am_g.max()) == float(G[:, :c].abs().max()), tag
