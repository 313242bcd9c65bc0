"""K4 training step (DenseNet-121 + NW head, 32 queries + 10 supports @224) as a captured HIP graph vs eager."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.model import load_model
from nwhead_amd.nwhead.kernel import get_kernel
from nwhead_amd.nwhead.nw import NWHead
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
dn = load_model("densenet121").to(dev).train()
opt = torch.optim.SGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4,
                              fused=os.environ.get("NW_SGD_FOREACH", "0") != "1")
xq = torch.randn(32, 3, 224, 224, generator=g).to(dev); yq = torch.randint(0, 10, (32,), generator=g).to(dev)
xs = torch.randn(10, 3, 224, 224, generator=g).to(dev); ys = torch.arange(10).to(dev)
head = NWHead(get_kernel("euclidean"), 10)
xin = torch.cat((xq, xs))

def fwd_bwd():
    feats = dn(xin)
    loss = F.nll_loss(head(feats[:32], feats[32:], ys), yq)
    loss.backward()
    return loss

def timeit(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def eager():
    opt.zero_grad(set_to_none=True)
    fwd_bwd()
    opt.step()

print(f"eager step: {timeit(eager):.2f} ms", flush=True)
# capture: grads must exist (static tensors), the optimizer step inside the graph
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        fwd_bwd()
        opt.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=False)
with torch.cuda.graph(graph):
    for p in dn.parameters():
        p.grad.zero_()
    loss = fwd_bwd()
    opt.step()
graph.replay(); torch.cuda.synchronize()
print(f"graph step: {timeit(graph.replay):.2f} ms, loss {float(loss):.4f}", flush=True)
