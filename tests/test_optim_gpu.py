"""nwhead_amd.optim.SGD (nw_sgd_step_f32) against torch.optim.SGD, the optimizer of the reference's train.py:243-247."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 7, 7), (128,), (128,), (1,), (3,), (4097,), (32, 128, 3, 3), (128, 992, 1, 1), (5, 7), (1000, 1024)]
    shapes += [(64 + 8 * k,) for k in range(120)]            # more tensors than one launch carries (96)
    return [torch.randn(s, generator=g).to(dev) for s in shapes]


@pytest.mark.parametrize("momentum,nesterov,wd", [(0.9, True, 1e-4), (0.9, False, 0.0), (0.0, False, 5e-4), (0.5, True, 0.0)])
def test_sgd_matches_torch_over_several_steps(dev, momentum, nesterov, wd):
    from nwhead_amd.optim import SGD
    mine = [torch.nn.Parameter(t.clone()) for t in _params(dev, 3)]
    ref = [torch.nn.Parameter(t.clone()) for t in _params(dev, 3)]
    o1 = SGD(mine, lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    o2 = torch.optim.SGD(ref, lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    g = torch.Generator().manual_seed(11)
    for step in range(4):
        for a, b in zip(mine, ref):
            if step == 1 and a.numel() == 3:                  # a parameter without a gradient in one step: skipped by both
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).to(dev)
            if a.dim() == 4 and step == 2:                    # a gradient in another memory layout
                gr = gr.contiguous(memory_format=torch.channels_last)
            a.grad, b.grad = gr.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
        o1.step(); o2.step()
        if step == 1:
            for grp in o1.param_groups + o2.param_groups:     # a scheduler's step
                grp["lr"] = 0.01
    for a, b in zip(mine, ref):
        assert (a - b).abs().max().item() <= 2e-6 * max(float(b.abs().max()), 1.0)
        if momentum:
            ba, bb = o1.state[a]["momentum_buffer"], o2.state[b]["momentum_buffer"]
            assert (ba - bb).abs().max().item() <= 2e-6 * max(float(bb.abs().max()), 1.0)


def test_sgd_state_dict_is_torchs(dev):
    """A run continues in torch.optim.SGD from this optimizer's state_dict, and the other way round."""
    from nwhead_amd.optim import SGD
    g = torch.Generator().manual_seed(5)
    base = [torch.randn(257, generator=g).to(dev), torch.randn(16, 8, generator=g).to(dev)]
    grads = [[torch.randn(t.shape, generator=g).to(dev) for t in base] for _ in range(4)]

    def run(first, second):
        ps = [torch.nn.Parameter(t.clone()) for t in base]
        kw = dict(lr=0.1, momentum=0.9, weight_decay=1e-3, nesterov=True)
        opt = first(ps, **kw)
        for k in range(2):
            for p, gr in zip(ps, grads[k]):
                p.grad = gr.clone()
            opt.step()
        opt2 = second(ps, **kw)
        opt2.load_state_dict(opt.state_dict())
        for k in range(2, 4):
            for p, gr in zip(ps, grads[k]):
                p.grad = gr.clone()
            opt2.step()
        return [p.detach().clone() for p in ps]

    a, b, c = run(SGD, torch.optim.SGD), run(torch.optim.SGD, SGD), run(torch.optim.SGD, torch.optim.SGD)
    for x, y, z in zip(a, b, c):
        assert (x - z).abs().max().item() < 2e-6 and (y - z).abs().max().item() < 2e-6


def test_sgd_refuses_what_it_does_not_serve(dev):
    from nwhead_amd.optim import SGD
    from nwhead_amd.ops import NWHipError
    p = torch.nn.Parameter(torch.randn(8))
    p.grad = torch.randn(8)
    with pytest.raises(NWHipError):
        SGD([p], lr=0.1).step()
    with pytest.raises(ValueError):
        SGD([torch.nn.Parameter(torch.randn(4, device=dev))], lr=0.1, nesterov=True)
