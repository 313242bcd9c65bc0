"""The implicit-GEMM 3x3 convolution of the folded inference backbones (csrc/conv3x3.hip: stride 1, padding 1, fp32 matrix
cores, bias / residual / ReLU fused, output into a channel window of a slab) against torch's conv2d in fp64 on the host."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,cin,cout,h,w,slab_in,bias,res,relu,into_slab", [
    (4, 128, 32, 56, 56, 64, False, False, False, True),     # DenseNet conv2 on 56x56: 32 x 256 tiles, 16-byte DMAs, into a slab
    (3, 128, 32, 28, 28, 0, False, False, False, True),      # 28x28: ragged last tile (784 = 3 * 256 + 16)
    (2, 20, 32, 14, 14, 5, True, False, True, False),        # cin % 8 != 0, W % 4 != 0 (4-byte DMAs), one partly filled tile
    (2, 64, 64, 56, 56, 0, True, True, True, False),         # BasicBlock tail: bias + identity + ReLU, 64 x 128 tiles
    (2, 40, 128, 12, 12, 0, True, False, True, False),       # two M tiles of 64, one full + one ragged N tile
    (5, 24, 256, 7, 7, 0, True, True, True, False),          # 7x7 planes, 128 x 64 tiles, W % 4 != 0
    (1, 3, 32, 5, 9, 0, False, False, False, False),         # tiny: every column on a border somewhere
    # (few images: the cases above run 32 x 64 tiles with the K range shared by the workgroup's waves, like the small
    #  planes of a real batch; the ones below have enough tiles for the large-tile forms)
    (16, 16, 32, 56, 56, 8, False, False, False, True),      # 32 x 256 tiles (208 workgroups), into a slab
    (28, 16, 64, 28, 28, 0, True, True, True, False),        # 64 x 128 tiles (196 workgroups), ragged last tile
    (96, 8, 256, 7, 7, 0, True, False, True, False),         # 128 x 64 tiles (192 workgroups), 4-byte DMAs
    (64, 128, 32, 14, 14, 32, False, False, False, True),    # DenseNet block 3 at batch 64: K-split tiles, W % 4 != 0, slab
    (20, 72, 32, 16, 16, 0, True, False, True, False),       # K-split tiles with 16-byte DMAs, nine stages
    (64, 520, 32, 7, 7, 24, False, False, False, True),      # DenseNet block 4 at batch 64: K also split over 4 workgroups, slab
    (8, 250, 32, 7, 7, 0, True, True, True, False),          # ... ragged last K chunk, bias + residual + ReLU in the reduce kernel
    (64, 128, 32, 7, 7, 0, False, False, False, True),       # DenseNet's own 7x7 shape: four chunks of four stages
])
def test_conv3x3_against_torch_fp64(n, cin, cout, h, w, slab_in, bias, res, relu, into_slab):
    from nwhead_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1000 * n + cin + h)
    full = torch.randn(n, cin + slab_in, h, w, generator=g)
    x = full[:, :cin]
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    r = torch.randn(n, cout, h, w, generator=g) if res else None
    ref = F.conv2d(x.double(), wgt.double(), None if b is None else b.double(), padding=1)
    if res:
        ref = ref + r.double()
    if relu:
        ref = F.relu(ref)
    mv = lambda t: None if t is None else t.to(dev)
    wt = ops.conv3x3_weight(wgt.to(dev))
    assert wt.shape == ((cin + 7) // 8, 9, 8, cout)
    if into_slab:
        slab = torch.full((n, cout + 24, h, w), 7.0, device=dev)
        out = ops.conv3x3(full.to(dev)[:, :cin], wt, cin, mv(b), mv(r), relu, out=slab[:, 8:8 + cout])
        assert out.data_ptr() == slab[:, 8:8 + cout].data_ptr()
        assert float(slab[:, :8].min()) == 7.0 and float(slab[:, 8 + cout:].max()) == 7.0     # neighbours untouched
    else:
        out = ops.conv3x3(full.to(dev)[:, :cin], wt, cin, mv(b), mv(r), relu)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6 * max(scale, 1.0))


def test_folded_densenets_with_fused_3x3():
    import nwhead_amd.model.backbones as bb
    from nwhead_amd.model import fold_batchnorm, load_model
    dev = torch.device("cuda:0")
    for name, side, batch in (("densenet121", 224, 8), ("CIFAR_DenseNet121", 32, 64)):
        torch.manual_seed(0)
        net = load_model(name).to(dev)
        net.train()
        with torch.no_grad():
            net(torch.randn(4, 3, side, side, device=dev))
        net.eval()
        x = torch.randn(batch, 3, side, side, device=dev)
        was, was_nhwc = bb.FUSED_CONV3X3, bb.NHWC_INFERENCE
        bb.FUSED_CONV3X3 = True                      # (the default; NW_OWN_CONV3X3=0 in the environment turns it off)
        bb.NHWC_INFERENCE = False                    # (round 4: DenseNet-121 itself runs the channels-last path; see test_conv1x1_gpu)
        try:
            folded = fold_batchnorm(net)
            assert sum(isinstance(m, bb.Conv3x3Fused) for m in folded.modules()) == 58
            with torch.no_grad():
                want, got = net(x), folded(x)
        finally:
            bb.FUSED_CONV3X3, bb.NHWC_INFERENCE = was, was_nhwc
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)


def test_folded_resnet_with_fused_3x3_blocks():
    """backbones.FUSED_RESNET_CONV3X3 (off by default: MIOpen's channels_last kernels are ahead on these shapes): the
    stride-1 3x3 convolutions of the BasicBlocks with bias, identity and ReLU in the kernel's store."""
    import nwhead_amd.model.backbones as bb
    from nwhead_amd.model import fold_batchnorm, load_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = load_model("resnet18").to(dev)
    net.train()
    with torch.no_grad():
        net(torch.randn(4, 3, 96, 96, device=dev))
    net.eval()
    x = torch.randn(16, 3, 96, 96, device=dev)
    was = bb.FUSED_RESNET_CONV3X3
    bb.FUSED_RESNET_CONV3X3 = True
    try:
        folded = fold_batchnorm(net)
    finally:
        bb.FUSED_RESNET_CONV3X3 = was
    assert sum(isinstance(m, bb.Conv3x3Fused) for m in folded.modules()) == 13     # 16 3x3 convolutions, 3 of them strided
    with torch.no_grad():
        want, got = net(x), folded(x)
    scale = float(want.abs().max())
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
