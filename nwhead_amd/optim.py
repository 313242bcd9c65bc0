"""The harness's optimizer on the MI355X: torch.optim.SGD's update (reference train.py:243-247: momentum 0.9, weight decay,
Nesterov) for all parameter tensors of a network in a few launches of nw_sgd_step_f32 (csrc/sgd.hip) instead of torch's
multi-tensor kernels -- DenseNet-121 has 364 parameter tensors, two thirds of them BatchNorm vectors; 0.34 -> 0.05 ms per
step.  Same hyper-parameters, same state ('momentum_buffer' per parameter: state_dict()s are interchangeable with
torch.optim.SGD's), same arithmetic (dampening 0, no maximize).  HIP fp32 parameters only: there is no CPU path."""
import ctypes

import torch

from . import _lib
from .ops import NWHipError, _OnDevice


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("SGD: lr, momentum and weight_decay must be non-negative")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        # (torch.optim.SGD's group keys, so that state_dict()s load either way; the values this class does not serve are refused)
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=nesterov,
                                      maximize=False, foreach=None, differentiable=False, fused=None))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            mu = float(group["momentum"])
            if group.get("dampening", 0) != 0 or group.get("maximize", False):
                raise NWHipError("nwhead_amd.optim.SGD: dampening and maximize are not served (torch.optim.SGD is)")
            fresh, old = [], []                    # parameters taking their first step (buffer := gradient) / all later ones
            keep = []
            touched = []                           # tensors the launches below write through raw pointers
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse:
                    raise NWHipError("nwhead_amd.optim.SGD updates dense fp32 parameters on the HIP device only "
                                     "(torch.optim.SGD serves everything else)")
                if not p.is_contiguous():
                    raise NWHipError("nwhead_amd.optim.SGD needs contiguous parameters")
                if not g.is_contiguous():
                    g = g.contiguous()             # (a gradient in another layout: its elements in the parameter's order)
                    keep.append(g)
                buf_ptr, first = None, False
                if mu != 0.0:
                    st = self.state[p]
                    buf = st.get("momentum_buffer")
                    if buf is None:
                        buf = st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                        first = True
                    buf_ptr = buf.data_ptr()
                (fresh if first else old).append((p.device, p.data_ptr(), g.data_ptr(), buf_ptr, p.numel()))
                touched.append(p)
                if mu != 0.0:
                    touched.append(buf)
            for jobs, init in ((fresh, 1), (old, 0)):
                by_dev = {}
                for dev, *rest in jobs:
                    by_dev.setdefault(dev, []).append(rest)
                for dev, rows in by_dev.items():
                    arr = (_lib.SgdParam * len(rows))(*[_lib.SgdParam(a, b, c, n) for a, b, c, n in rows])
                    with _OnDevice(dev):
                        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                        _lib.check(lib.nw_sgd_step_f32(arr, len(rows), float(group["lr"]), mu, float(group["weight_decay"]),
                                                       int(bool(group["nesterov"])), init, stream), "nw_sgd_step_f32")
            del keep
            # The kernel writes through data_ptr(): tell autograd.  Everything that caches on `_version` (NWNet's folded
            # inference copy, ConvBiasAct's split weights, ConvWeightBank.refresh) and autograd's own saved-tensor check
            # (step() between forward and backward) relies on it; host-only, no launch.
            for t in touched:
                if not t.is_inference():
                    torch.autograd.graph.increment_version(t)
        return loss
