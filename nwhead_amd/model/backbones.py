"""Headless CNN featurizers that feed the NW head: image batch (n,3,H,W) -> feature rows (n,feat).

Own definitions of the architectures the reference trains (model/resnet.py, model/densenet.py,
model/densenet3.py), with the same parameter/buffer names so that reference checkpoints load
(`featurizer.*` keys) and the same initialisation schemes.  The convolutions run on
MIOpen (MFMA) through PyTorch-ROCm; what is specific here:

  * DenseNet blocks run concat-free under `torch.no_grad()` (precompute()/predict(), the bulk of the
    backbone work in 'full' inference): one channel slab per block is allocated up front and every
    layer writes its `growth_rate` new channels into it instead of re-concatenating all previous
    features (the reference's `torch.cat` per layer, densenet.py:70-75, copies O(L^2) channels);
  * the factories the reference ships broken (`densenet121`, `CIFAR_DenseNet121` pass arguments their
    constructors do not take: SURVEY section 2 rows 6-7) work here and return the intended networks.

Feature widths: resnet18 / CIFAR_ResNet18 512, densenet121 1024, CIFAR_DenseNet121 1024.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)


def _no_pretrained(pretrained):
    if pretrained:
        raise RuntimeError("pretrained weights need a download; load a state_dict instead")


# ----------------------------------------------------------------------------- ImageNet-style ResNet
def _conv_nhwc_train(conv, x, bank):
    """A bias-free nn.Conv2d on the channels-last training kernels (ops.conv2d_nhwc_train; model/resnet.py:15-28)."""
    from .. import ops
    return ops.conv2d_nhwc_train(x, conv.weight, conv.stride[0], conv.padding[0], operands=bank.operands(conv.weight))


def _fold_conv_bn_split(conv, bn):
    """conv -> eval-mode BatchNorm as ONE split-row weight and a bias (what ops.conv2d_nhwc takes)."""
    from .. import ops
    a = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    wf = conv.weight.detach().float() * a.view(-1, 1, 1, 1)
    b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(a)
    return ops.SplitConvWeight(wf), (bn.bias.detach().float() + (b0 - bn.running_mean.detach().float()) * a).contiguous()


def _bn_apply_args(bn):
    """(mean, invstd, gamma, beta) of an eval-mode BatchNorm2d for ops.bn_relu_nhwc_apply."""
    return (bn.running_mean.detach().float().contiguous(), torch.rsqrt(bn.running_var.detach().float() + bn.eps).contiguous(),
            bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous())


def _add_relu_nhwc(z, identity):
    """relu(z + identity) at the end of a residual block (model/resnet.py:60-66, :100-108), with the amax record the next
    convolution scales its operand by."""
    from .. import ops
    return ops.add_relu_nhwc(z, identity)


def _convs_nhwc_servable(model, stem):
    """Do csrc/conv_nhwc.hip / conv_wgrad.hip serve every convolution (and bn_nhwc.hip every BatchNorm) of the network?"""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            cout, cin, kh, kw = m.weight.shape
            ok = (m.bias is None and m.groups == 1 and m.dilation == (1, 1) and kh == kw and m.stride[0] == m.stride[1]
                  and m.padding[0] == m.padding[1] and cout % 32 == 0)
            if m is stem:
                ok = ok and cin == 3 and 4 * kw <= 32
            else:
                ok = ok and cin % 32 == 0 and kh in (1, 3) and m.padding[0] == (kh - 1) // 2 and m.stride[0] in (1, 2)
            if not ok:
                return False
        elif isinstance(m, nn.BatchNorm2d) and not (m.affine and m.num_features % 4 == 0):
            return False
    return True


class BasicBlock(nn.Module):
    """conv3x3-bn-relu-conv3x3-bn (+ identity or 1x1 projection) - relu; model/resnet.py:31-66."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv(planes, planes, 3, 1, 1), nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if isinstance(self.conv2, Conv3x3Fused):     # folded inference copy with FUSED_RESNET_CONV3X3: bias, identity and
            y = self.conv1(x) if isinstance(self.conv1, Conv3x3Fused) else F.relu(self.conv1(x))   # ReLU in the kernels
            return self.conv2(y, residual=x if self.downsample is None else self.downsample(x))
        if isinstance(self.conv2, ConvBiasAct):      # folded inference copy: bias, identity and ReLU behind each convolution
            return self.conv2(self.conv1(x, relu=True), residual=x if self.downsample is None else self.downsample(x), relu=True)
        y = self.conv2(_bn_relu(self.bn1, self.conv1(x)))
        return _bn_relu(self.bn2, y, x if self.downsample is None else self.downsample(x))

    def forward_nhwc_train(self, x, bank):
        """The block in channels-last layout on the own kernels (ResNet._forward_nhwc_train)."""
        from .. import ops
        y = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv1, x, bank), self.bn1)
        z = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv2, y, bank), self.bn2, relu=False)
        idt = x if self.downsample is None else ops.bn_relu_train_nhwc(_conv_nhwc_train(self.downsample[0], x, bank),
                                                                         self.downsample[1], relu=False)
        return _add_relu_nhwc(z, idt)


class Bottleneck(nn.Module):
    """1x1-3x3-1x1 bottleneck with expansion 4; model/resnet.py:69-108."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if isinstance(self.conv3, ConvBiasAct):      # folded inference copy
            y = self.conv2(self.conv1(x, relu=True), relu=True)
            return self.conv3(y, residual=x if self.downsample is None else self.downsample(x), relu=True)
        y = _bn_relu(self.bn1, self.conv1(x))
        y = _bn_relu(self.bn2, self.conv2(y))
        return _bn_relu(self.bn3, self.conv3(y), x if self.downsample is None else self.downsample(x))

    def forward_nhwc_train(self, x, bank):
        from .. import ops
        y = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv1, x, bank), self.bn1)
        y = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv2, y, bank), self.bn2)
        z = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv3, y, bank), self.bn3, relu=False)
        idt = x if self.downsample is None else ops.bn_relu_train_nhwc(_conv_nhwc_train(self.downsample[0], x, bank),
                                                                         self.downsample[1], relu=False)
        return _add_relu_nhwc(z, idt)


class ResNet(nn.Module):
    """7x7/2 stem, max-pool, four stages, global average pool; no fc (model/resnet.py:136-207)."""

    def __init__(self, block, layers, zero_init_residual=False):
        super().__init__()
        self.inplanes = 64
        self.conv1, self.bn1 = _conv(3, 64, 7, 2, 3), nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for idx, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), start=1):
            setattr(self, f"layer{idx}", self._stage(block, planes, n, stride))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)
                elif isinstance(m, BasicBlock):
                    nn.init.zeros_(m.bn2.weight)

    def _stage(self, block, planes, n, stride):
        out = planes * block.expansion
        down = None
        if stride != 1 or self.inplanes != out:
            down = nn.Sequential(_conv(self.inplanes, out, 1, stride), nn.BatchNorm2d(out))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = out
        blocks += [block(out, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def _nhwc_train_servable(self):
        ok = getattr(self, "_nw_nhwc_ok", None)
        if ok is None:
            ok = self._nw_nhwc_ok = (_convs_nhwc_servable(self, self.conv1) and _is_pool(self.maxpool, nn.MaxPool2d, 3, 2, 1)
                                     and all(type(d) is nn.Sequential and len(d) == 2 for d in
                                             (b.downsample for st in (self.layer1, self.layer2, self.layer3, self.layer4) for b in st)
                                             if d is not None))
        return ok

    def _forward_nhwc_train(self, x):
        """The training forward in channels-last layout on the MI355X (round 4; same module sequence as
        model/resnet.py:192-207): every convolution -- forward, data and weight gradient, the strided ones included -- in
        csrc/conv_nhwc.hip / conv_wgrad.hip, BatchNorm (+ ReLU) in csrc/bn_nhwc.hip, the max pool in pool_nhwc.hip."""
        from .. import ops
        bank = getattr(self, "_nw_bank", None)
        if bank is None or bank.weights[0] is not self.conv1.weight:
            convs = [(m.weight, m is not self.conv1) for m in self.modules() if isinstance(m, nn.Conv2d)]
            bank = self._nw_bank = ops.ConvWeightBank(convs)
        bank.refresh(force=True)        # one launch; a fused optimizer's step leaves no trace in the version counters
        y = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv1, x, bank), self.bn1)
        y = ops.maxpool3s2_nhwc(y)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                y = blk.forward_nhwc_train(y, bank)
        return torch.flatten(self.avgpool(y), 1)

    def forward(self, x):
        if (NHWC_TRAINING and RESNET_NHWC_TRAINING and self.training and x.is_cuda and x.dtype == torch.float32
                and torch.is_grad_enabled() and x.dim() == 4 and x.shape[1] == 3 and not isinstance(self.conv1, ConvBiasAct)
                and self._nhwc_train_servable()):
            return self._forward_nhwc_train(x)
        if isinstance(self.conv1, ConvBiasAct):      # folded inference copy
            c1 = self.conv1
            if (FUSED_CONV_NHWC and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and not torch.is_grad_enabled()
                    and c1.kernel_size == (7, 7) and c1.stride == (2, 2) and c1.padding == (3, 3) and c1.out_channels == 64
                    and c1.groups == 1 and c1.dilation == (1, 1) and c1.bias is not None and _is_pool(self.maxpool, nn.MaxPool2d, 3, 2, 1)):
                from .. import ops       # stem convolution + ReLU + max pool in one kernel: the 112 x 112 map stays on the CU
                y = ops.stem_conv_relu_maxpool_nhwc(x, c1._split_weight(), c1.bias)
                y = self.layer4(self.layer3(self.layer2(self.layer1(y))))
                return torch.flatten(self.avgpool(y), 1)
            y = self.conv1(x, relu=True)
            if (y.is_cuda and y.dtype == torch.float32 and y.shape[1] % 4 == 0 and _is_pool(self.maxpool, nn.MaxPool2d, 3, 2, 1)
                    and y.is_contiguous(memory_format=torch.channels_last) and not torch.is_grad_enabled()):
                from .. import ops
                x = ops.maxpool3s2_nhwc(y)          # (keeps y's amax bound: the next convolution needs it)
            else:
                x = self.maxpool(y)
                if hasattr(y, "nw_amax"):           # a bound on max|.| survives the pooling
                    x.nw_amax = y.nw_amax
        else:
            x = self.maxpool(_bn_relu(self.bn1, self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return torch.flatten(self.avgpool(x), 1)


# ----------------------------------------------------------------------------- CIFAR pre-act ResNet
DENSE_PASSTHROUGH = True        # the running concatenation passes through norm1's autograd node (see _DenseBlock.forward)
DENSE_INCREMENTAL_CAT = True    # dense blocks extend one running concatenation (see _DenseBlock.forward)
# Folded inference copies: the dense layers' 3x3 convolutions as the implicit-GEMM kernel (Conv3x3Fused,
# csrc/conv3x3.hip), which writes straight into the dense-block slab.  On DenseNet-121's shape (128 -> 32 channels,
# batch 64) it draws level with MIOpen's Winograd kernels on the 56x56 and 28x28 planes (165 vs 156 us, 52 vs 51) and
# is 2-3x ahead on 14x14 and 7x7 (23 vs 47 us, 16 vs 47: 32 x 64 tiles, K range shared inside / split over workgroups);
# DenseNet-121 over 64 images 5.29 ms with it, 5.72 without (tools/fold_time.py, same box).  NW_OWN_CONV3X3=0: off.
import os as _os
FUSED_CONV3X3 = _os.environ.get("NW_OWN_CONV3X3", "1") != "0"
# ResNet's BasicBlocks in the folded copies: stride-1 3x3 convolutions through the same kernel with the folded BatchNorm
# bias, the identity and the ReLU in its store (NCHW; the stem, the strided convolutions and the 1x1 projections stay on
# MIOpen).  Measured against the channels_last MIOpen path (tools/fold_time.py): see DESIGN.md 4.7e; NW_RESNET_OWN_CONV3X3.
FUSED_RESNET_CONV3X3 = _os.environ.get("NW_RESNET_OWN_CONV3X3", "0") == "1"
# Folded channels_last copies (the ResNets): every convolution through ops.conv2d_nhwc (csrc/conv_nhwc.hip: implicit GEMM on
# the fp16 matrix cores with split-fp16 operands, bias / identity / ReLU in its store).  NW_CONV_NHWC=0: MIOpen + the bias pass.
FUSED_CONV_NHWC = _os.environ.get("NW_CONV_NHWC", "1") != "0"
FUSED_CONV1X1 = True            # folded inference copies: 1x1 convolutions with their BatchNorm / ReLU neighbours as one kernel (Conv1x1Fused)
FUSED_BN_RELU_TRAINING = True   # training-mode BatchNorm2d + ReLU through ops.bn_relu_train on the MI355X
# Training on the MI355X in channels-last layout: every convolution (forward, data and weight gradient) on the fp16 matrix
# cores with split-fp16 operands (ops.conv2d_nhwc_train, csrc/conv_nhwc.hip + conv_wgrad.hip) and BatchNorm + ReLU through
# csrc/bn_nhwc.hip; the DenseNets take this path.  NW_NHWC_TRAINING=0: the NCHW path (MIOpen convolutions).
NHWC_TRAINING = _os.environ.get("NW_NHWC_TRAINING", "1") != "0"
# ... and the ImageNet-style ResNets (round 4): NW_RESNET_NHWC_TRAINING=0 keeps them on the NCHW path (MIOpen convolutions + bnrelu.hip)
RESNET_NHWC_TRAINING = _os.environ.get("NW_RESNET_NHWC_TRAINING", "1") != "0"
# ... and the CIFAR DenseNets (densenet3.py), layer by layer with torch's concatenations (their new-channels-first order does not fit
# the slab node): vendor-free, but 22.1 vs 19.8 ms per 128-image step against the NCHW / MIOpen path -- OFF by default
CIFAR_DENSENET_NHWC_TRAINING = _os.environ.get("NW_CIFAR_DENSENET_NHWC_TRAINING", "0") == "1"
# a dense block of the channels-last training path as one autograd node over one slab (ops._DenseBlockNhwcFn)
DENSE_SLAB = _os.environ.get("NW_DENSE_SLAB", "1") != "0"
# a transition of that path as BatchNorm-ReLU -> 2x2 average pool -> 1x1 convolution (pool and convolution commute)
TRANSITION_POOL_FIRST = _os.environ.get("NW_TRANSITION_POOL_FIRST", "1") != "0"


def _fused_training(bn, x):
    return (FUSED_BN_RELU_TRAINING and isinstance(bn, nn.BatchNorm2d) and bn.training and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 4 and bn.affine and torch.is_grad_enabled()
            and (x.stride(3) == 1 and x.stride(2) == x.shape[3] and x.stride(1) == x.shape[2] * x.shape[3]))


def _bn_relu(bn, x, residual=None):
    """relu(bn(x)), or relu(bn(x) + residual) at the end of a ResNet block.  One pass in a folded inference copy
    (ScaleShiftReLU); one HIP kernel each way in training on the device (ops.bn_relu_train); otherwise the torch
    ops of the reference."""
    if isinstance(bn, ScaleShiftReLU):
        return bn(x)
    if _fused_training(bn, x):
        from .. import ops
        return ops.bn_relu_train(x, bn, True, residual)
    return F.relu(bn(x) if residual is None else bn(x) + residual)


class PreActBlock(nn.Module):
    """bn-relu-conv3x3-bn-relu-conv3x3 with the shortcut taken after the first activation
    (model/resnet.py:111-134)."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.bn1, self.conv1 = nn.BatchNorm2d(in_planes), _conv(in_planes, planes, 3, stride, 1)
        self.bn2, self.conv2 = nn.BatchNorm2d(planes), _conv(planes, planes, 3, 1, 1)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(_conv(in_planes, planes, 1, stride))

    def forward(self, x):
        a = _bn_relu(self.bn1, x)
        y = self.conv2(_bn_relu(self.bn2, self.conv1(a)))
        return y + self.shortcut(a)

    def forward_nhwc_train(self, x, bank):
        """The block in channels-last layout on the own kernels (CIFAR_ResNet._forward_nhwc_train)."""
        from .. import ops
        a = ops.bn_relu_train_nhwc(x, self.bn1)
        y = _conv_nhwc_train(self.conv2, ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv1, a, bank), self.bn2), bank)
        return y + (a if len(self.shortcut) == 0 else _conv_nhwc_train(self.shortcut[0], a, bank))


class CIFAR_ResNet(nn.Module):
    """3x3 stem, four stages, 4x4 average pool: 32x32 inputs only (model/resnet.py:209-239)."""

    def __init__(self, block, num_blocks):
        super().__init__()
        self.in_planes = 64
        self.conv1, self.bn1 = _conv(3, 64, 3, 1, 1), nn.BatchNorm2d(64)
        for idx, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), num_blocks, (1, 2, 2, 2)), start=1):
            blocks = []
            for s in [stride] + [1] * (n - 1):
                blocks.append(block(self.in_planes, planes, s))
                self.in_planes = planes * block.expansion
            setattr(self, f"layer{idx}", nn.Sequential(*blocks))

    def _nhwc_train_servable(self):
        ok = getattr(self, "_nw_nhwc_ok", None)
        if ok is None:
            ok = self._nw_nhwc_ok = (_convs_nhwc_servable(self, self.conv1)
                                     and all(isinstance(b, PreActBlock) and len(b.shortcut) <= 1
                                             for st in (self.layer1, self.layer2, self.layer3, self.layer4) for b in st))
        return ok

    def _nhwc_infer_servable(self):
        """Plain modules with running statistics (a copy fold_batchnorm has rewritten runs its own modules)."""
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d) and (type(m) is not nn.BatchNorm2d or m.running_mean is None or not m.affine):
                return False
        return all(type(b.bn1) is nn.BatchNorm2d and type(b.bn2) is nn.BatchNorm2d and isinstance(b.conv1, nn.Conv2d)
                   and isinstance(b.conv2, nn.Conv2d) for st in (self.layer1, self.layer2, self.layer3, self.layer4) for b in st)

    def _forward_nhwc_train(self, x):
        """The training forward in channels-last layout on the MI355X (round 4; model/resnet.py:209-239, the reference's default
        CIFAR backbone): every convolution incl. the strided ones and the 1x1 shortcuts in csrc/conv_nhwc.hip / conv_wgrad.hip,
        BatchNorm + ReLU in csrc/bn_nhwc.hip."""
        from .. import ops
        bank = getattr(self, "_nw_bank", None)
        if bank is None or bank.weights[0] is not self.conv1.weight:
            convs = [(m.weight, m is not self.conv1) for m in self.modules() if isinstance(m, nn.Conv2d)]
            bank = self._nw_bank = ops.ConvWeightBank(convs)
        bank.refresh(force=True)
        y = ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv1, x, bank), self.bn1)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                y = blk.forward_nhwc_train(y, bank)
        return torch.flatten(F.avg_pool2d(y, 4), 1)

    def _infer_plan(self, dev):
        """What the inference kernels read: split-row weights (conv -> BatchNorm pairs folded), biases, BatchNorm factors.
        Rebuilt when a parameter or buffer has changed (address or in-place version)."""
        from .. import ops
        ts = list(self.parameters()) + list(self.buffers())
        sig = (str(dev), len(ts), sum(0 if t.is_inference() else t._version for t in ts), sum(t.data_ptr() & 0xffffff for t in ts))
        plan = getattr(self, "_nw_infer_plan", None)
        if plan is not None and plan["sig"] == sig:
            return plan
        plan = {"sig": sig, "stem": _fold_conv_bn_split(self.conv1, self.bn1), "blocks": []}
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for b in stage:
                sc = ops.SplitConvWeight(b.shortcut[0].weight.detach().float()) if len(b.shortcut) else None
                plan["blocks"].append((_bn_apply_args(b.bn1), _fold_conv_bn_split(b.conv1, b.bn2), b.conv1.stride[0],
                                       ops.SplitConvWeight(b.conv2.weight.detach().float()), sc))
        object.__setattr__(self, "_nw_infer_plan", plan)
        return plan

    @torch.no_grad()
    def _forward_nhwc_infer(self, x):
        """Eval-mode forward on the channels-last split-fp16 kernels (round 4; model/resnet.py:111-134, :209-239): the stem with bn1
        folded in and ReLU in its store; per block relu(bn1(x)) in one pass (nw_bn_relu_nhwc_apply_f32: the shortcut needs it too),
        conv1 with bn2 folded in + ReLU, conv2 with the shortcut added in its store."""
        from .. import ops
        plan = self._infer_plan(x.device)
        w0, b0 = plan["stem"]
        y = ops.conv2d_nhwc(x, w0, b0, None, True, 1, 1)
        for bn1, (w1, b1), stride, w2, wsc in plan["blocks"]:
            a = ops.bn_relu_nhwc_apply(y, *bn1)
            sc = a if wsc is None else ops.conv2d_nhwc(a, wsc, None, None, False, stride, 0, want_amax=False)
            t = ops.conv2d_nhwc(a, w1, b1, None, True, stride, 1)
            y = ops.conv2d_nhwc(t, w2, None, sc, False, 1, 1, want_amax=False)
        return torch.flatten(F.avg_pool2d(y, 4), 1)

    def forward(self, x, lin=0, lout=5):
        if (NHWC_TRAINING and RESNET_NHWC_TRAINING and self.training and x.is_cuda and x.dtype == torch.float32
                and torch.is_grad_enabled() and x.dim() == 4 and x.shape[1] == 3 and isinstance(self.conv1, nn.Conv2d)
                and self._nhwc_train_servable()):
            return self._forward_nhwc_train(x)
        if (NHWC_INFERENCE and not self.training and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and x.dim() == 4 and x.shape[1] == 3 and isinstance(self.conv1, nn.Conv2d) and type(self.bn1) is nn.BatchNorm2d
                and self._nhwc_train_servable() and self._nhwc_infer_servable()):
            return self._forward_nhwc_infer(x)
        x = _bn_relu(self.bn1, self.conv1(x))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return torch.flatten(F.avg_pool2d(x, 4), 1)


# ----------------------------------------------------------------------------- DenseNet-BC (ImageNet)
class _DenseLayer(nn.Sequential):
    """norm1-relu1-conv1(1x1, bn_size*k) - norm2-relu2-conv2(3x3, k); densenet.py:33-60."""

    def __init__(self, cin, growth_rate, bn_size, drop_rate):
        super().__init__(OrderedDict([
            ("norm1", nn.BatchNorm2d(cin)), ("relu1", nn.ReLU(inplace=True)),
            ("conv1", _conv(cin, bn_size * growth_rate, 1)),
            ("norm2", nn.BatchNorm2d(bn_size * growth_rate)), ("relu2", nn.ReLU(inplace=True)),
            ("conv2", _conv(bn_size * growth_rate, growth_rate, 3, 1, 1))]))
        self.drop_rate = drop_rate

    def forward(self, x):
        # relu1 must not run in place on a slab/concat that later layers re-read
        if isinstance(self.conv1, Conv1x1Fused):         # folded inference copy: norm1-relu1-conv1-norm2-relu2 in one kernel
            return self.conv2(self.conv1(x))              #   (conv2: Conv3x3Fused when FUSED_CONV3X3)
        if isinstance(self.norm1, ScaleShiftReLU):       # folded inference copy without the fused 1x1 (FUSED_CONV1X1 off)
            return self.conv2(self.relu2(self.conv1(self.norm1(x))))
        return self.after_norm1(_bn_relu(self.norm1, x))

    def after_norm1(self, a):
        y = self.conv2(_bn_relu(self.norm2, self.conv1(a)))
        return F.dropout(y, self.drop_rate, self.training) if self.drop_rate > 0 else y


class _DenseBlock(nn.Module):
    def __init__(self, num_layers, cin, bn_size, growth_rate, drop_rate):
        super().__init__()
        self.cin, self.growth_rate = cin, growth_rate
        for i in range(num_layers):
            self.add_module(f"denselayer{i + 1}", _DenseLayer(cin + i * growth_rate, growth_rate, bn_size, drop_rate))

    def slab_room(self):
        """Channels the block's layers append to its input."""
        return sum(layer.conv2.weight.shape[0] for layer in self.children())

    def forward_nhwc_train(self, x, bank=None):
        """Channels-last training forward of the block (DenseNet._forward_nhwc_train): the running concatenation passes
        through norm1's autograd node like in forward() below."""
        from .. import ops
        layers = list(self.children())
        if DENSE_SLAB and ops.dense_block_nhwc_supported(x, layers, bank):
            return ops.dense_block_nhwc_train(x, layers, bank)      # one autograd node over one slab: no concatenations
        cur = x
        for layer in layers:
            a, cur = ops.bn_relu_train_nhwc(cur, layer.norm1, True, passthrough=True)
            o1 = bank.operands(layer.conv1.weight) if bank is not None else None
            o2 = bank.operands(layer.conv2.weight) if bank is not None else None
            t = ops.bn_relu_train_nhwc(ops.conv2d_nhwc_train(a, layer.conv1.weight, 1, 0, operands=o1), layer.norm2)
            new = ops.conv2d_nhwc_train(t, layer.conv2.weight, 1, 1, operands=o2)
            if layer.drop_rate > 0:
                new = F.dropout(new, layer.drop_rate, self.training)
            cur = torch.cat((cur, new), 1)
        return cur

    def forward(self, x):
        layers = list(self.children())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # The running concatenation is extended by one layer at a time (same values and copy volume as
            # concatenating the list of features at every layer, densenet.py:62-80): its backward is then ONE
            # gradient accumulation per layer instead of one per (feature, later layer) pair -- 58 instead of
            # 540 small strided adds in a DenseNet-121 step.
            if not DENSE_INCREMENTAL_CAT:
                feats = [x]
                for layer in layers:
                    feats.append(layer(torch.cat(feats, 1)))
                return torch.cat(feats, 1)
            cur = x
            for layer in layers:
                if DENSE_PASSTHROUGH and _fused_training(layer.norm1, cur):
                    # cur feeds norm1 AND the next concatenation: it goes through the BatchNorm node and comes out
                    # again, so the concatenation's gradient is added to dx inside the backward kernel
                    from .. import ops
                    a, cur = ops.bn_relu_train(cur, layer.norm1, True, None, passthrough=True)
                    cur = torch.cat((cur, layer.after_norm1(a)), 1)
                else:
                    cur = torch.cat((cur, layer(cur)), 1)
            return cur
        # inference: one slab, each layer appends its channels (no per-layer re-concatenation)
        n, c, h, w = x.shape
        slab = x.new_empty(n, c + len(layers) * self.growth_rate, h, w)
        slab[:, :c] = x
        for layer in layers:
            dst = slab[:, c:c + self.growth_rate]
            if isinstance(layer.conv2, Conv3x3Fused) and isinstance(layer.conv1, Conv1x1Fused) and layer.drop_rate == 0:
                layer.conv2(layer.conv1(slab[:, :c]), out=dst)     # the 3x3 kernel writes its channels into the slab
            else:
                slab[:, c:c + self.growth_rate] = layer(slab[:, :c])
            c += self.growth_rate
        return slab


def _is_pool(mod, kind, k, s, p):
    """mod is exactly kind(k, s, p) with the defaults the reference uses (no ceil_mode, dilation 1, padded values counted)."""
    def same(v, want):
        return v == want or v == (want, want)
    return (type(mod) is kind and same(mod.kernel_size, k) and same(mod.stride, s) and same(mod.padding, p)
            and not mod.ceil_mode and same(getattr(mod, "dilation", 1), 1) and not getattr(mod, "return_indices", False)
            and getattr(mod, "count_include_pad", True) and getattr(mod, "divisor_override", None) is None)


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(OrderedDict([("norm", nn.BatchNorm2d(cin)), ("relu", nn.ReLU(inplace=True)),
                                      ("conv", _conv(cin, cout, 1)), ("pool", nn.AvgPool2d(2, 2))]))

    def forward(self, x):
        if isinstance(self.conv, Conv1x1Fused):          # folded inference copy: norm-relu-pool, then the 1x1 convolution
            return self.pool(self.conv(x))                #   (the pool module is the identity when it ran first)
        return self.pool(self.conv(_bn_relu(self.norm, x)))


class DenseNet(nn.Module):
    """densenet.py:93-163, classifier removed: relu + global average pool of `features`."""

    def __init__(self, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4,
                 drop_rate=0, memory_efficient=False, bias=True):
        super().__init__()
        self.features = nn.Sequential(OrderedDict([
            ("conv0", _conv(3, num_init_features, 7, 2, 3)), ("norm0", nn.BatchNorm2d(num_init_features)),
            ("relu0", nn.ReLU(inplace=True)), ("pool0", nn.MaxPool2d(3, 2, 1))]))
        c = num_init_features
        for i, n in enumerate(block_config, start=1):
            self.features.add_module(f"denseblock{i}", _DenseBlock(n, c, bn_size, growth_rate, drop_rate))
            c += n * growth_rate
            if i != len(block_config):
                self.features.add_module(f"transition{i}", _Transition(c, c // 2))
                c //= 2
        self.features.add_module("norm5", nn.BatchNorm2d(c))
        self.num_features = c
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        if (NHWC_TRAINING and self.training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and x.dim() == 4 and x.shape[1] == 3 and self._nhwc_servable()):
            return self._forward_nhwc_train(x)
        if (NHWC_INFERENCE and not self.training and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and x.dim() == 4 and x.shape[1] == 3 and self._nhwc_servable() and self._nhwc_infer_servable(x)):
            return self._forward_nhwc_infer(x)
        return torch.flatten(F.adaptive_avg_pool2d(F.relu(self.features(x)), (1, 1)), 1)

    # ------------------------------------------------------------------ inference on the channels-last kernels (round 4)
    def _nhwc_infer_servable(self, x):
        """The module sequence the inference plan covers: the reference's (densenet.py:114-145), plain BatchNorm2d / Conv2d
        modules with running statistics (a copy fold_batchnorm has rewritten runs its own modules), the reference's pools,
        no dropout, maps that halve cleanly."""
        f = self.features
        mods = list(f.children())
        if not (isinstance(f.conv0, nn.Conv2d) and type(f.norm0) is nn.BatchNorm2d and type(f.norm5) is nn.BatchNorm2d
                and _is_pool(f.pool0, nn.MaxPool2d, 3, 2, 1) and f.conv0.stride == (2, 2) and f.conv0.padding == (3, 3)):
            return False
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d) and (m.running_mean is None or not m.affine):
                return False
            if isinstance(m, _DenseLayer) and (m.drop_rate > 0 or type(m.norm1) is not nn.BatchNorm2d or type(m.norm2) is not nn.BatchNorm2d
                                               or not isinstance(m.conv1, nn.Conv2d) or not isinstance(m.conv2, nn.Conv2d)
                                               or m.conv2.kernel_size != (3, 3)):
                return False
            if isinstance(m, _Transition) and (type(m.norm) is not nn.BatchNorm2d or not isinstance(m.conv, nn.Conv2d)
                                               or not _is_pool(m.pool, nn.AvgPool2d, 2, 2, 0)):
                return False
        hw = ((x.shape[2] + 6 - 7) // 2 + 1, (x.shape[3] + 6 - 7) // 2 + 1)
        hw = ((hw[0] - 1) // 2 + 1, (hw[1] - 1) // 2 + 1)
        for mod in mods:
            if isinstance(mod, _Transition):
                if hw[0] % 2 or hw[1] % 2:
                    return False
                hw = (hw[0] // 2, hw[1] // 2)
        return min(hw) >= 1

    def _infer_plan(self, dev):
        """Per module what the inference kernels read: folded split-row weights, biases, BatchNorm tables.  Rebuilt when a
        parameter or buffer has changed (address or in-place version)."""
        from .. import ops
        ts = list(self.parameters()) + list(self.buffers())
        sig = (str(dev), len(ts), sum(0 if t.is_inference() else t._version for t in ts), sum(t.data_ptr() & 0xffffff for t in ts))
        plan = getattr(self, "_nw_infer_plan", None)
        if plan is not None and plan["sig"] == sig:
            return plan
        f = self.features

        def fold(conv, bn):         # conv -> BatchNorm as one weight and bias
            a = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
            wf = conv.weight.detach().float() * a.view(-1, 1, 1, 1)
            b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(a)
            return ops.SplitConvWeight(wf), (bn.bias.detach().float() + (b0 - bn.running_mean.detach().float()) * a).contiguous()
        plan = {"sig": sig, "stem": fold(f.conv0, f.norm0), "mods": []}
        for mod in f.children():
            if isinstance(mod, _DenseBlock):
                layers = []
                for layer in mod.children():
                    w1, b1 = fold(layer.conv1, layer.norm2)
                    layers.append((ops.bn_table(layer.norm1), w1, b1, ops.SplitConvWeight(layer.conv2.weight.detach().float())))
                plan["mods"].append(("block", layers))
            elif isinstance(mod, _Transition):
                plan["mods"].append(("transition", (ops.bn_table(mod.norm), ops.SplitConvWeight(mod.conv.weight.detach().float()))))
        n5 = f.norm5
        plan["norm5"] = (n5.running_mean.detach().float().contiguous(),
                         torch.rsqrt(n5.running_var.detach().float() + n5.eps).contiguous(),
                         n5.weight.detach().float().contiguous(), n5.bias.detach().float().contiguous())
        object.__setattr__(self, "_nw_infer_plan", plan)
        return plan

    @torch.no_grad()
    def _forward_nhwc_infer(self, x):
        """Eval-mode forward on the channels-last split-fp16 kernels (the copy of round 2 ran NCHW fp32-MFMA
        kernels and MIOpen's stem): folded stem convolution + ReLU, max pool into the first block's slab, dense blocks as two
        launches per layer with the BatchNorms in the convolutions (ops.dense_block_nhwc_infer), transitions in the
        reference's order norm-relu-conv-pool with norm + relu in the convolution's loaders."""
        from .. import ops
        f = self.features
        plan = self._infer_plan(x.device)
        mods = [m for m in f.children() if isinstance(m, (_DenseBlock, _Transition))]
        rooms = [m.slab_room() if isinstance(m, _DenseBlock) else 0 for m in mods] + [0]
        w0, b0 = plan["stem"]
        y = ops.stem_conv_relu_maxpool_nhwc(x, w0, b0, rooms[0])        # (one kernel where it serves the shape)
        for i, (mod, (kind, pl)) in enumerate(zip(mods, plan["mods"])):
            if kind == "block":
                y = ops.dense_block_nhwc_infer(y, pl)
            else:
                tab, wt = pl
                if TRANSITION_POOL_FIRST and y.shape[2] >= 2 and y.shape[3] >= 2:
                    # BatchNorm + ReLU + the 2 x 2 average in one pass, then the (bias-free: it commutes with the pool) 1 x 1
                    # convolution on a quarter of the pixels, written straight into the next block's slab
                    y = ops.conv2d_nhwc(ops.bn_relu_avgpool2_nhwc(y, tab), wt, None, None, False, 1, 0, room=rooms[i + 1])
                else:
                    y = ops.avgpool2_nhwc(ops.conv1x1_bnrelu_nhwc_infer(y, tab, wt), rooms[i + 1])
        y = ops.bn_relu_nhwc_apply(y, *plan["norm5"])           # norm5 + the relu of DenseNet.forward (densenet.py:139, :160)
        return torch.flatten(F.adaptive_avg_pool2d(y, (1, 1)), 1)

    def _nhwc_servable(self):
        """Do the channels-last kernels serve every layer?  (convolutions: input and output channels in multiples of 32 --
        DenseNet-161's growth of 48 is not -- apart from the RGB stem; BatchNorms: channels in multiples of 4.)  Otherwise the
        NCHW path (MIOpen convolutions + the NCHW BatchNorm kernels) runs, as for every other module layout."""
        ok = getattr(self, "_nw_nhwc_ok", None)
        if ok is None:
            f = self.features
            ok = True
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    cout, cin = m.weight.shape[:2]
                    if m is f.conv0:
                        ok &= cin == 3 and cout % 32 == 0 and m.bias is None
                    else:
                        ok &= cin % 32 == 0 and cout % 32 == 0 and m.bias is None and m.stride == (1, 1) and m.groups == 1
                elif isinstance(m, nn.BatchNorm2d):
                    ok &= m.num_features % 4 == 0 and m.affine
            self._nw_nhwc_ok = ok = bool(ok)
        return ok

    def _forward_nhwc_train(self, x):
        """The training forward in channels-last layout on the MI355X (same values as the reference's module sequence,
        densenet.py:93-163): activations fp32 NHWC, BatchNorm + ReLU pairs through ops.bn_relu_train_nhwc (they leave the
        amax record the following convolution scales its operand by), convolutions through ops.conv2d_nhwc_train."""
        from .. import ops
        f = self.features
        bank = getattr(self, "_nw_bank", None)
        if bank is None or bank.weights[0] is not f.conv0.weight:
            # every convolution weight's split-row operands (forward + data gradient), rebuilt by one launch per step
            convs = [(m.weight, m is not f.conv0) for m in self.modules() if isinstance(m, nn.Conv2d)]
            bank = self._nw_bank = ops.ConvWeightBank(convs)
        bank.refresh(force=True)        # one launch (80 us); a fused optimizer's step leaves no trace in the version counters
        y = ops.bn_relu_train_nhwc(ops.conv2d_nhwc_train(x, f.conv0.weight, 2, 3, operands=bank.operands(f.conv0.weight)), f.norm0)
        mods = list(f.children())

        def room(i):        # the pool in front of a dense block writes into the block's slab-to-be (ops._with_room)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            return nxt.slab_room() if isinstance(nxt, _DenseBlock) and DENSE_SLAB else 0

        i0 = mods.index(f.pool0)
        y = ops.maxpool3s2_nhwc(y, room(i0)) if _is_pool(f.pool0, nn.MaxPool2d, 3, 2, 1) else f.pool0(y)
        for i, mod in enumerate(mods):
            if isinstance(mod, _DenseBlock):
                y = mod.forward_nhwc_train(y, bank)
            elif isinstance(mod, _Transition):
                t = ops.bn_relu_train_nhwc(y, mod.norm)
                if TRANSITION_POOL_FIRST and _is_pool(mod.pool, nn.AvgPool2d, 2, 2, 0) and mod.conv.bias is None \
                        and t.shape[2] % 2 == 0 and t.shape[3] % 2 == 0:
                    # the 2 x 2 average and the bias-free 1 x 1 convolution commute (both linear, per pixel / per channel):
                    # pooled first, the convolution and both of its gradients work on a quarter of the pixels
                    # (densenet.py:83-91 runs conv -> pool; the values differ by fp32 rounding only)
                    y = ops.conv2d_nhwc_train(ops.avgpool2_nhwc(t), mod.conv.weight, 1, 0, operands=bank.operands(mod.conv.weight),
                                              room=room(i))
                else:
                    z = ops.conv2d_nhwc_train(t, mod.conv.weight, 1, 0, operands=bank.operands(mod.conv.weight))
                    y = ops.avgpool2_nhwc(z, room(i)) if _is_pool(mod.pool, nn.AvgPool2d, 2, 2, 0) else mod.pool(z)
        y = ops.bn_relu_train_nhwc(y, f.norm5)          # (norm5 + the relu of DenseNet.forward)
        return torch.flatten(F.adaptive_avg_pool2d(y, (1, 1)), 1)


NHWC_INFERENCE = _os.environ.get("NW_NHWC_INFERENCE", "1") != "0"   # DenseNet.forward in eval mode on the device: the channels-last inference path above


class ScaleShiftReLU(nn.Module):
    """Eval-mode BatchNorm2d followed by ReLU as one pass, y = max(a_c x + b_c, 0): on the MI355X the HIP
    kernel nw_scale_shift_relu_f32 (ops.scale_shift_relu), on CPU tensors the same two torch ops the
    reference backbone runs.  Built by fold_batchnorm for the BN -> ReLU -> conv layers that cannot fold."""

    def __init__(self, bn, relu=True):
        super().__init__()
        a = bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)
        self.register_buffer("scale", a.clone())
        self.register_buffer("shift", (bn.bias.detach() - bn.running_mean.detach() * a).clone())
        self.relu = relu

    def forward(self, x):
        if x.is_cuda:
            from .. import ops
            return ops.scale_shift_relu(x, self.scale, self.shift, self.relu)
        y = x * self.scale.view(1, -1, 1, 1) + self.shift.view(1, -1, 1, 1)
        return F.relu(y) if self.relu else y


class Conv1x1Fused(nn.Module):
    """[eval-mode BatchNorm -> ReLU ->] 1x1 convolution [-> folded BatchNorm bias -> ReLU] as ONE kernel on the
    MI355X (ops.conv1x1, csrc/conv1x1.hip: fp32 matrix cores); on CPU tensors the same torch ops the reference
    backbone runs.  Built by fold_batchnorm for DenseNet's dense layers (norm1-relu1-conv1-norm2-relu2) and
    transitions (norm-relu-conv) and their CIFAR counterparts.  Inference only."""

    def __init__(self, conv, pre_bn=None, post_bn=None, post_relu=False, pool_first=False):
        super().__init__()
        # pool_first: the module is followed by a 2x2 average pool and has no bias, BatchNorm or ReLU behind the
        # convolution (a DenseNet transition): pool and convolution commute, forward() returns avg_pool2d(conv(..), 2)
        # computed as conv(avg_pool2d(..)) -- a quarter of the pixels in the matrix product
        assert not pool_first or (post_bn is None and not post_relu and conv.bias is None and pre_bn is not None)
        self.pool_first = pool_first
        assert conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
        w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels)
        b = conv.bias.detach() if conv.bias is not None else None
        if post_bn is not None:                      # BatchNorm behind the convolution: into weight and bias
            a = post_bn.weight.detach() * torch.rsqrt(post_bn.running_var.detach() + post_bn.eps)
            w = w * a[:, None]
            b = post_bn.bias.detach() + ((b if b is not None else 0) - post_bn.running_mean.detach()) * a
        self.register_buffer("weight", w.contiguous().clone())            # (cout, cin): the CPU path
        wt = w.t().contiguous()
        pad = (-wt.shape[0]) % 16                    # the kernel's operand: (cin rounded up to 16, cout), zero rows behind
        self.register_buffer("weight_t", torch.cat((wt, wt.new_zeros(pad, wt.shape[1]))).contiguous() if pad else wt.clone())
        self.register_buffer("bias", None if b is None else b.clone())
        if pre_bn is not None:
            a = pre_bn.weight.detach() * torch.rsqrt(pre_bn.running_var.detach() + pre_bn.eps)
            self.register_buffer("pre_scale", a.clone())
            self.register_buffer("pre_shift", (pre_bn.bias.detach() - pre_bn.running_mean.detach() * a).clone())
        else:
            self.pre_scale = self.pre_shift = None
        self.pre_relu, self.post_relu = pre_bn is not None, post_relu

    def forward(self, x):
        if x.is_cuda and self.weight_t.shape[1] % 4 == 0:
            from .. import ops
            if self.pool_first:
                return ops.conv1x1(ops.scale_shift_relu_avgpool2(x, self.pre_scale, self.pre_shift), self.weight_t,
                                   cin=self.weight.shape[1])
            return ops.conv1x1(x, self.weight_t, self.bias, self.pre_scale, self.pre_shift, self.pre_relu, self.post_relu,
                               cin=self.weight.shape[1])
        if self.pre_scale is not None:
            x = F.relu(x * self.pre_scale.view(1, -1, 1, 1) + self.pre_shift.view(1, -1, 1, 1))
        y = F.conv2d(x, self.weight[:, :, None, None], self.bias)
        y = F.relu(y) if self.post_relu else y
        return F.avg_pool2d(y, 2) if self.pool_first else y


class Conv3x3Fused(nn.Module):
    """3x3 convolution (stride 1, padding 1) [-> folded BatchNorm bias] [+ residual] [-> ReLU] as ONE kernel on the
    MI355X (ops.conv3x3, csrc/conv3x3.hip: implicit GEMM on the fp32 matrix cores), written straight into `out` when
    given (a channel window of a dense block's slab); on CPU tensors -- and for shapes that would leave most of the
    chip idle (fewer than 192 workgroups: 7x7 and 14x14 planes of narrow layers) -- the same torch ops the reference
    backbone runs.  Built by fold_batchnorm.  Inference only."""

    def __init__(self, conv, post_bn=None, post_relu=False):
        super().__init__()
        assert conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == 1 \
            and conv.dilation == (1, 1)
        w = conv.weight.detach()
        b = conv.bias.detach() if conv.bias is not None else None
        if post_bn is not None:
            a = post_bn.weight.detach() * torch.rsqrt(post_bn.running_var.detach() + post_bn.eps)
            w = w * a[:, None, None, None]
            b = post_bn.bias.detach() + ((b if b is not None else 0) - post_bn.running_mean.detach()) * a
        from ..ops import conv3x3_weight
        self.cin, self.cout = w.shape[1], w.shape[0]
        self.register_buffer("weight", w.contiguous().clone())           # (cout, cin, 3, 3): the torch path
        self.register_buffer("weight_t", conv3x3_weight(w))               # (ceil(cin / 8), 9, 8, cout): the kernel's operand
        self.register_buffer("bias", None if b is None else b.clone())
        self.post_relu = post_relu

    def _use_kernel(self, x):
        if not x.is_cuda or self.cout % 32 != 0 or x.dtype != torch.float32:
            return False
        n, _, h, w = x.shape
        from .. import _lib
        # enough workgroups for the chip (small planes run 32 x 64 tiles with the K range shared inside the workgroup
        # and, with many input channels, split over workgroups)
        return h * w >= 4 and _lib.load().nw_conv3x3_workgroups(n, self.cin, self.cout, h, w) >= 192

    def forward(self, x, out=None, residual=None):
        if self._use_kernel(x):
            from .. import ops, _lib
            try:
                return ops.conv3x3(x, self.weight_t, self.cin, self.bias, residual, self.post_relu, out)
            except _lib.NWHipError as e:      # a shape the kernel refuses (wide images, misaligned views): the torch ops
                if "status -2" not in str(e) and "status -1" not in str(e):
                    raise
        y = F.conv2d(x, self.weight, self.bias, padding=1)
        if residual is not None:
            y = y + residual
        if self.post_relu:
            y = F.relu(y)
        if out is not None:
            out.copy_(y)
            return out
        return y


# ----------------------------------------------------------------------------- CIFAR DenseNet
class CifarBottleneck(nn.Module):
    """bn1-relu-conv1(1x1,4k)-bn2-relu-conv2(3x3,k), output = cat(new, x); densenet3.py:10-22."""

    def __init__(self, in_planes, growth_rate):
        super().__init__()
        self.bn1, self.conv1 = nn.BatchNorm2d(in_planes), _conv(in_planes, 4 * growth_rate, 1)
        self.bn2, self.conv2 = nn.BatchNorm2d(4 * growth_rate), _conv(4 * growth_rate, growth_rate, 3, 1, 1)

    def forward(self, x):
        if isinstance(self.conv1, Conv1x1Fused):         # folded inference copy: bn1-relu-conv1-bn2-relu in one kernel
            return torch.cat([self.conv2(self.conv1(x)), x], 1)
        y = self.conv2(_bn_relu(self.bn2, self.conv1(_bn_relu(self.bn1, x))))
        return torch.cat([y, x], 1)

    def forward_nhwc_train(self, x, bank):
        """The layer in channels-last layout on the own kernels (CIFAR_DenseNet._forward_nhwc_train); the concatenation keeps the
        reference's order -- new channels FIRST (densenet3.py:20-21) -- so it stays a copy (the slab node of the ImageNet DenseNets
        appends)."""
        from .. import ops
        y = _conv_nhwc_train(self.conv2, ops.bn_relu_train_nhwc(_conv_nhwc_train(self.conv1, ops.bn_relu_train_nhwc(x, self.bn1), bank),
                                                              self.bn2), bank)
        return torch.cat([y, x], 1).contiguous(memory_format=torch.channels_last)


class CifarTransition(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.bn, self.conv = nn.BatchNorm2d(in_planes), _conv(in_planes, out_planes, 1)

    def forward(self, x):
        if isinstance(self.conv, Conv1x1Fused):          # folded inference copy: bn-relu-pool, then the 1x1 convolution
            y = self.conv(x)
            return y if self.conv.pool_first else F.avg_pool2d(y, 2)
        return F.avg_pool2d(self.conv(_bn_relu(self.bn, x)), 2)

    def forward_nhwc_train(self, x, bank):
        from .. import ops
        t = ops.bn_relu_train_nhwc(x, self.bn)
        if TRANSITION_POOL_FIRST and t.shape[2] % 2 == 0 and t.shape[3] % 2 == 0:     # (pool and the bias-free convolution commute)
            return _conv_nhwc_train(self.conv, ops.avgpool2_nhwc(t), bank)
        return ops.avgpool2_nhwc(_conv_nhwc_train(self.conv, t, bank))


class CIFAR_DenseNet(nn.Module):
    """densenet3.py:37-83 (32x32 inputs)."""

    def __init__(self, block, nblocks, growth_rate=12, reduction=0.5):
        super().__init__()
        self.growth_rate = growth_rate
        c = 2 * growth_rate
        self.conv1 = _conv(3, c, 3, 1, 1)
        for i, n in enumerate(nblocks, start=1):
            layers = []
            for _ in range(n):
                layers.append(block(c, growth_rate))
                c += growth_rate
            setattr(self, f"dense{i}", nn.Sequential(*layers))
            if i != len(nblocks):
                out = int(math.floor(c * reduction))
                setattr(self, f"trans{i}", CifarTransition(c, out))
                c = out
        self.bn = nn.BatchNorm2d(c)
        self.num_features = c

    def _nhwc_train_servable(self):
        ok = getattr(self, "_nw_nhwc_ok", None)
        if ok is None:
            ok = self._nw_nhwc_ok = (isinstance(self.conv1, nn.Conv2d) and _convs_nhwc_servable(self, self.conv1)
                                     and all(isinstance(m, (CifarBottleneck, CifarTransition))
                                             for k in (1, 2, 3, 4) for m in getattr(self, f"dense{k}"))
                                     and all(isinstance(getattr(self, f"trans{k}"), CifarTransition) for k in (1, 2, 3)))
        return ok

    def _forward_nhwc_train(self, x):
        """The training forward in channels-last layout on the MI355X (round 4; densenet3.py:37-83): every convolution in
        csrc/conv_nhwc.hip / conv_wgrad.hip, BatchNorm + ReLU in csrc/bn_nhwc.hip, the transitions' pools in pool_nhwc.hip; the
        concatenations are torch copies."""
        from .. import ops
        bank = getattr(self, "_nw_bank", None)
        if bank is None or bank.weights[0] is not self.conv1.weight:
            convs = [(m.weight, m is not self.conv1) for m in self.modules() if isinstance(m, nn.Conv2d)]
            bank = self._nw_bank = ops.ConvWeightBank(convs)
        bank.refresh(force=True)
        y = _conv_nhwc_train(self.conv1, x, bank)
        for k in (1, 2, 3, 4):
            for layer in getattr(self, f"dense{k}"):
                y = layer.forward_nhwc_train(y, bank)
            if k < 4:
                y = getattr(self, f"trans{k}").forward_nhwc_train(y, bank)
        return torch.flatten(F.avg_pool2d(ops.bn_relu_train_nhwc(y, self.bn), 4), 1)

    def forward(self, x):
        if (NHWC_TRAINING and CIFAR_DENSENET_NHWC_TRAINING and self.training and x.is_cuda and x.dtype == torch.float32
                and torch.is_grad_enabled() and x.dim() == 4 and x.shape[1] == 3 and self._nhwc_train_servable()):
            return self._forward_nhwc_train(x)
        x = self.conv1(x)
        x = self.trans1(self.dense1(x))
        x = self.trans2(self.dense2(x))
        x = self.trans3(self.dense3(x))
        x = self.dense4(x)
        return torch.flatten(F.avg_pool2d(_bn_relu(self.bn, x), 4), 1)


# ----------------------------------------------------------------------------- factories (load_model names)
def resnet10(pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [1, 1, 1, 1], **kw)


def resnet18(pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 4, 6, 3], **kw)


def resnet101(pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 4, 23, 3], **kw)


def resnet152(pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 8, 36, 3], **kw)


def CIFAR_ResNet10(pretrained=False, **kw):
    return CIFAR_ResNet(PreActBlock, [1, 1, 1, 1], **kw)


def CIFAR_ResNet18(pretrained=False, **kw):
    return CIFAR_ResNet(PreActBlock, [2, 2, 2, 2], **kw)


def CIFAR_ResNet34(pretrained=False, **kw):
    return CIFAR_ResNet(PreActBlock, [3, 4, 6, 3], **kw)


def _densenet(growth_rate, block_config, num_init_features, pretrained, num_classes=None, **kw):
    _no_pretrained(pretrained)          # num_classes is accepted and ignored: the net is headless
    return DenseNet(growth_rate, block_config, num_init_features, **kw)


def densenet121(pretrained=False, progress=True, **kw):
    return _densenet(32, (6, 12, 24, 16), 64, pretrained, **kw)


def densenet161(pretrained=False, progress=True, **kw):
    return _densenet(48, (6, 12, 36, 24), 96, pretrained, **kw)


def densenet169(pretrained=False, progress=True, **kw):
    return _densenet(32, (6, 12, 32, 32), 64, pretrained, **kw)


def densenet201(pretrained=False, progress=True, **kw):
    return _densenet(32, (6, 12, 48, 32), 64, pretrained, **kw)


def CIFAR_DenseNet121(pretrained=False, num_classes=10, bias=True, **kw):
    return CIFAR_DenseNet(CifarBottleneck, [6, 12, 24, 16], growth_rate=32)


# ---------------------------------------------------------------------------------------------
# Eval-mode BatchNorm folding.  In ResNet / Bottleneck blocks and the ResNet stem a
# BatchNorm directly follows a bias-free convolution, so at inference (running statistics) the pair is
# one convolution: w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps).
# The pre-activation designs (DenseNet, CIFAR_DenseNet, CIFAR_ResNet's PreActBlock) apply BN -> ReLU -> conv
# -> BN -> ReLU -> conv: the inner conv -> BN pair (and the stems') folds; the BatchNorms in front of a ReLU
# (norm1 / bn1, the transitions', the last one) become ScaleShiftReLU (one pass instead of torch's batch-norm
# + relu kernels).
# ---------------------------------------------------------------------------------------------
class ConvBiasAct(nn.Conv2d):
    """A convolution with a folded BatchNorm in it (fold_batchnorm): conv(x) + bias [+ residual] [-> ReLU].  On the
    MI355X with channels_last fp32 tensors (what NWNet.enable_bn_folding gives the ResNets) the bias, the block's identity
    and the ReLU are ONE in-place pass behind a bias-free convolution (ops.bias_act_nhwc_) instead of MIOpen's bias kernel
    + add + relu; everywhere else the torch ops of the reference backbone.  Inference only."""

    def _split_weight(self):
        w = self.weight
        from .. import ops
        key = (w.data_ptr(), ops._ver(w), str(w.device))   # (a copy folded under inference_mode has no version counter)
        if getattr(self, "_nw_split_key", None) != key:
            self._nw_split, self._nw_split_key = ops.SplitConvWeight(w), key
        return self._nw_split

    def forward(self, x, residual=None, relu=False):
        c = self.out_channels
        if (FUSED_CONV_NHWC and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_grad_enabled()
                and self.groups == 1 and self.dilation == (1, 1) and self.padding[0] == self.padding[1]
                and self.stride[0] == self.stride[1] and isinstance(self.padding[0], int)
                and (x.is_contiguous(memory_format=torch.channels_last) or x.shape[1] == 3)):
            from .. import ops
            if ops.conv2d_nhwc_supported(x.shape, self.weight.shape, self.stride[0], self.padding[0]):
                # the whole conv -> bias -> (+ identity) -> ReLU on the fp16 matrix cores at fp32-grade accuracy
                return ops.conv2d_nhwc(x, self._split_weight(), self.bias, residual, relu, self.stride[0], self.padding[0])
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and c % 4 == 0 and self.bias is not None
                and not torch.is_grad_enabled() and x.is_contiguous(memory_format=torch.channels_last)
                and not x.is_contiguous()):
            y = F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
            if y.is_contiguous(memory_format=torch.channels_last) and (
                    residual is None or (residual.shape == y.shape and residual.dtype == torch.float32
                                         and residual.is_contiguous(memory_format=torch.channels_last))):
                from .. import ops
                return ops.bias_act_nhwc_(y, self.bias, residual, relu)
            y = y + self.bias.view(1, -1, 1, 1)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


def _fold_pair(conv, bn):
    scale = bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)
    fused = ConvBiasAct(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                        conv.dilation, conv.groups, bias=True).to(conv.weight.device, conv.weight.dtype)
    fused.weight.data.copy_(conv.weight.detach() * scale.view(-1, 1, 1, 1))
    bias = conv.bias.detach() if conv.bias is not None else torch.zeros_like(scale)
    fused.bias.data.copy_(bn.bias.detach() + (bias - bn.running_mean.detach()) * scale)
    return fused


def fold_batchnorm(model):
    """A deep copy of `model` (eval mode) in which every conv -> BatchNorm pair of the ResNet family and of
    DenseNet is one convolution and DenseNet's BatchNorm -> ReLU pairs are one pass; other architectures come
    back unchanged.  For inference only: the copy shares nothing with the original and must be re-made after
    the weights change."""
    import copy
    m = copy.deepcopy(model).eval()
    # A DenseNet on the device that the channels-last inference path serves (DenseNet._forward_nhwc_infer: it folds for itself,
    # from the plain modules) stays as it is; its eval forward on other tensors is the reference's module sequence.
    keep = set()
    for mod in m.modules():
        if isinstance(mod, DenseNet) and NHWC_INFERENCE and mod._nhwc_servable() and next(mod.parameters()).is_cuda:
            keep.update(id(x) for x in mod.modules())
        if (isinstance(mod, CIFAR_ResNet) and NHWC_INFERENCE and next(mod.parameters()).is_cuda and mod._nhwc_train_servable()
                and mod._nhwc_infer_servable()):
            keep.update(id(x) for x in mod.modules())      # (likewise: CIFAR_ResNet._forward_nhwc_infer)
    for mod in list(m.modules()):                # (the featurizer may sit inside a Sequential: proj_dim > 0)
        if id(mod) in keep:
            continue
        if isinstance(mod, DenseNet):
            f = mod.features
            f.conv0, f.norm0 = _fold_pair(f.conv0, f.norm0), nn.Identity()
            f.norm5 = ScaleShiftReLU(f.norm5)   # DenseNet.forward's F.relu on top is then the identity
        elif isinstance(mod, _DenseLayer) and FUSED_CONV1X1:
            mod.conv1 = Conv1x1Fused(mod.conv1, pre_bn=mod.norm1, post_bn=mod.norm2, post_relu=True)
            mod.norm1 = mod.relu1 = mod.norm2 = mod.relu2 = nn.Identity()
            if FUSED_CONV3X3:
                mod.conv2 = Conv3x3Fused(mod.conv2)
        elif isinstance(mod, _DenseLayer):
            mod.conv1, mod.norm2 = _fold_pair(mod.conv1, mod.norm2), nn.Identity()
            mod.norm1, mod.relu1 = ScaleShiftReLU(mod.norm1), nn.Identity()
        elif isinstance(mod, _Transition) and FUSED_CONV1X1:
            pool2 = isinstance(mod.pool, nn.AvgPool2d) and mod.pool.kernel_size in (2, (2, 2)) and mod.pool.stride in (2, (2, 2)) \
                and mod.pool.padding in (0, (0, 0)) and not mod.pool.ceil_mode
            mod.conv = Conv1x1Fused(mod.conv, pre_bn=mod.norm, pool_first=pool2 and mod.conv.bias is None)
            mod.norm = mod.relu = nn.Identity()
            if mod.conv.pool_first:
                mod.pool = nn.Identity()
        elif isinstance(mod, _Transition):
            mod.norm, mod.relu = ScaleShiftReLU(mod.norm), nn.Identity()
        elif isinstance(mod, CifarBottleneck) and FUSED_CONV1X1:
            mod.conv1 = Conv1x1Fused(mod.conv1, pre_bn=mod.bn1, post_bn=mod.bn2, post_relu=True)
            mod.bn1 = mod.bn2 = nn.Identity()
            if FUSED_CONV3X3:
                mod.conv2 = Conv3x3Fused(mod.conv2)
        elif isinstance(mod, (PreActBlock, CifarBottleneck)):
            mod.conv1, mod.bn2 = _fold_pair(mod.conv1, mod.bn2), nn.Identity()
            mod.bn1 = ScaleShiftReLU(mod.bn1)
        elif isinstance(mod, CifarTransition) and FUSED_CONV1X1:
            mod.conv, mod.bn = Conv1x1Fused(mod.conv, pre_bn=mod.bn, pool_first=mod.conv.bias is None), nn.Identity()
        elif isinstance(mod, (CifarTransition, CIFAR_DenseNet)):
            mod.bn = ScaleShiftReLU(mod.bn)
    for mod in m.modules():
        if id(mod) in keep:
            continue
        if isinstance(mod, BasicBlock) and FUSED_RESNET_CONV3X3 and isinstance(mod.conv2, nn.Conv2d) and mod.conv2.out_channels % 32 == 0:
            if mod.conv1.stride == (1, 1):
                mod.conv1, mod.bn1 = Conv3x3Fused(mod.conv1, mod.bn1, post_relu=True), nn.Identity()
            mod.conv2, mod.bn2 = Conv3x3Fused(mod.conv2, mod.bn2, post_relu=True), nn.Identity()
        if isinstance(mod, (BasicBlock, Bottleneck, ResNet)) or (isinstance(mod, CIFAR_ResNet) and hasattr(mod, "bn1")):
            k = 1
            while hasattr(mod, f"conv{k}") and hasattr(mod, f"bn{k}"):
                conv, bn = getattr(mod, f"conv{k}"), getattr(mod, f"bn{k}")
                if isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d):
                    setattr(mod, f"conv{k}", _fold_pair(conv, bn))
                    setattr(mod, f"bn{k}", nn.Identity())
                k += 1
        ds = getattr(mod, "downsample", None)
        if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d) and isinstance(ds[1], nn.BatchNorm2d):
            mod.downsample = nn.Sequential(_fold_pair(ds[0], ds[1]), nn.Identity())
    return m
