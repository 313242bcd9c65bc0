"""Host-side logic of the NWNet mirror that needs no GPU: bank ordering, sampler draws, env handling."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden


class _DS(torch.utils.data.Dataset):
    def __init__(self, data, targets):
        self.data, self.targets = data, list(targets)

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return self.data[i], self.targets[i]


def test_separated_indices_and_full_dataset_order():
    from nwhead_amd.nwhead.utils import FullDataset, get_separated_indices
    assert get_separated_indices([0, 1, 1, 2, 3]) == [[0], [1, 2], [3], [4]]
    assert get_separated_indices(torch.tensor([7, 3, 7, 5])) == [[1], [3], [0, 2]]     # ranked by value
    g = load_golden("g5_nwnet_plumbing.npz")
    ds = _DS(T(g["ds_data"]), g["ds_targets"].tolist())
    full = FullDataset(ds, 7)
    ys = [full[i][1] for i in range(len(full))]
    np.testing.assert_array_equal(np.array(ys), g["full_y"])           # class-sorted, balanced


def test_sampler_draws_match_reference_for_a_seed():
    from nwhead_amd.nwhead.support import SupportSetTrain
    g = load_golden("g5_nwnet_plumbing.npz")
    ds = _DS(T(g["ds_data"]), g["ds_targets"].tolist())
    st = SupportSetTrain(ds, 10, "random", n_shot=2, n_way=6)
    np.random.seed(99)
    sx, sy, sm = st.get_support(T(g["yq"]))
    np.testing.assert_array_equal(sy.numpy(), g["train_sy"])
    np.testing.assert_array_equal(sx.numpy(), g["train_sx"])
    assert sm.dtype == torch.float64 and float(sm.abs().sum()) == 0.0
    with pytest.raises(AssertionError):
        st.get_support(torch.arange(7))                     # more query classes than n_way


def test_eval_modes_error_contract():
    from nwhead_amd.nwhead.support import SupportSetEval
    g = load_golden("g5_nwnet_plumbing.npz")
    ds = _DS(T(g["ds_data"]), g["ds_targets"].tolist())
    ev = SupportSetEval(ds, 10, 1, 7)
    with pytest.raises(NotImplementedError):
        ev.get_support("bogus")
    with pytest.raises(AttributeError, match="precompute"):
        ev.get_support("full")
    assert len(ev.support_loaders) == 1 and len(ev.full_datasets[0]) == 70


def test_env_array_split():
    from nwhead_amd.nwhead.support import SupportSetEval
    data = torch.zeros(12, 2)
    ds = _DS(data, [0, 1, 2] * 4)
    ev = SupportSetEval(ds, 3, 1, 2, env_array=np.array([0] * 6 + [5] * 6))
    assert len(ev.env_datasets) == 2 and ev.env_map == {0: 0, 5: 1}
    assert [len(d) for d in ev.full_datasets] == [6, 6]


def test_shard_bounds_cover_the_bank():
    from nwhead_amd.sharded import shard_bounds
    for n, w in [(50000, 8), (10, 3), (7, 8), (0, 4)]:
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_kernel_factory_contract():
    from nwhead_amd.nwhead.kernel import get_kernel
    for name in ("euclidean", "hypersphere_euclidean", "cosine", "dotproduct", "clip"):
        assert get_kernel(name).kind == name
    assert abs(float(get_kernel("clip").logit_scale) - np.log(1 / 0.07)) < 1e-6
    with pytest.raises(NotImplementedError):
        get_kernel("relationnet")


def test_dropin_import_paths():
    """The reference's import lines (train.py:13-19, README usage) resolve to this implementation when
    dropin/ is ahead of everything else on the path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from nwhead.nw import NWNet, NWHead; from nwhead.kernel import get_kernel; from model import load_model; "
            "from util.metric import support_influence, Metric, ECELoss; from util import metric; "
            "from util.utils import parse_bool, ParseKwargs, summary, save_checkpoint, initialize_wandb; "   # train.py:16, verbatim
            "from util.utils import load_checkpoint; "
            "import nwhead_amd.nwhead.nw as a; assert NWNet is a.NWNet; "
            "m = load_model('resnet18'); print(type(m).__name__, type(get_kernel('euclidean')).__name__)")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "dropin"), root]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert r.stdout.split()[0] == "ResNet"


def test_parse_kwargs_action():
    import argparse
    from nwhead_amd.util.utils import ParseKwargs, initialize_wandb
    p = argparse.ArgumentParser()
    p.add_argument("--kw", nargs="*", action=ParseKwargs)
    ns = p.parse_args(["--kw", "a=1", "b=-0.5", "c=true", "d=False", "e=run-7x"])
    assert ns.kw == {"a": 1, "b": -0.5, "c": True, "d": False, "e": "run-7x"}
    with pytest.raises(NotImplementedError):
        initialize_wandb(ns)


def test_bucket_coalescing_views():
    """predict_stream hands a bucket of query batches to the kernels as ONE tensor: a view when the batches
    are consecutive slices of one buffer, a copy otherwise."""
    import torch
    from nwhead_amd.sharded import _coalesce
    buf = torch.arange(6 * 4 * 3, dtype=torch.float32).reshape(24, 3)
    parts = [buf[k * 4:(k + 1) * 4] for k in range(6)]
    v = _coalesce(parts[1:5])
    assert v.shape == (16, 3) and v.data_ptr() == parts[1].data_ptr() and torch.equal(v, buf[4:20])
    c = _coalesce([parts[0], parts[2]])                        # a gap: copy
    assert c.data_ptr() != parts[0].data_ptr() and torch.equal(c, torch.cat([parts[0], parts[2]]))
    other = [torch.ones(4, 3), torch.zeros(4, 3)]              # different storages: copy
    assert torch.equal(_coalesce(other), torch.cat(other))
    assert torch.equal(_coalesce([parts[3], parts[2]]), torch.cat([parts[3], parts[2]]))   # wrong order: copy


def test_bank_loader_workers_keep_the_row_order():
    """NWNet(..., loader_workers=2): the bank precompute() builds has the rows, in the order, of the single-process loader
    (the reference's loaders are num_workers = 0, support.py:164-165)."""
    import torch
    from nwhead_amd.nwhead.nw import NWNet

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            g = torch.Generator().manual_seed(3)
            self.x = torch.randn(90, 3, 4, 4, generator=g)
            self.targets = [i % 6 for i in range(90)]

        def __len__(self):
            return 90

        def __getitem__(self, i):
            return self.x[i], self.targets[i]

    banks = []
    for workers in (0, 2):
        torch.manual_seed(1)
        feat = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(48, 8))
        net = NWNet(feat, 6, support_dataset=DS(), n_shot_full=9, device="cpu", loader_workers=workers).eval()
        f, y = [], []
        for loader in net.support_eval.support_loaders:
            assert loader.num_workers == workers
            for img, label, _meta in loader:
                f.append(feat(img).detach())
                y.append(label)
        banks.append((torch.cat(f), torch.cat(y)))
    assert torch.equal(banks[0][1], banks[1][1]) and torch.equal(banks[0][0], banks[1][0])
    assert banks[0][1].tolist() == sorted(banks[0][1].tolist())


def test_dense_block_slab_node_is_a_device_path_only():
    """ops.dense_block_nhwc_supported: CPU tensors, a missing weight bank or dropout never select the slab node (the module
    then runs the reference's layer-by-layer sequence); nothing here touches the HIP library."""
    import nwhead_amd.model.backbones as BB
    from nwhead_amd import ops
    block = BB._DenseBlock(2, 64, 4, 32, 0.0)
    x = torch.randn(2, 64, 8, 8).contiguous(memory_format=torch.channels_last)
    assert not ops.dense_block_nhwc_supported(x, list(block.children()), None)

    class _Bank:                                              # would serve every weight: still refused for a CPU tensor
        def has(self, w):
            return True

        def operands(self, w):
            return (object(), object())
    assert not ops.dense_block_nhwc_supported(x, list(block.children()), _Bank())
    y = block(x)                                              # the CPU forward is the reference's sequence
    assert y.shape == (2, 64 + 2 * 32, 8, 8)


def test_nhwc_training_path_is_chosen_by_architecture():
    """DenseNet._nhwc_servable: the channels-last training kernels need channel counts in multiples of 32 -- DenseNet-161
    (growth 48) stays on the NCHW path (its training step used to die with NW_ERR_UNSUPPORTED), -121 / -169 / -201 do not."""
    from nwhead_amd.model import load_model
    assert load_model("densenet121")._nhwc_servable()
    assert load_model("densenet169")._nhwc_servable()
    assert not load_model("densenet161")._nhwc_servable()


def test_device_optimizer_has_no_cpu_path_and_the_harness_picks_by_device():
    """nwhead_amd.optim.SGD: torch's argument checks, torch.optim.SGD's group keys (state_dict()s load either way), and a
    loud refusal of CPU parameters -- the harness uses torch.optim.SGD on CPU (reference train.py:243-247) instead."""
    import pytest
    from nwhead_amd.optim import SGD
    from nwhead_amd.ops import NWHipError
    with pytest.raises(ValueError):
        SGD([torch.nn.Parameter(torch.zeros(3))], lr=0.1, nesterov=True)            # Nesterov needs a momentum
    p = torch.nn.Parameter(torch.zeros(8))
    opt = SGD([p], lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    ref = torch.optim.SGD([torch.nn.Parameter(torch.zeros(8))], lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    assert set(ref.param_groups[0]) <= set(opt.param_groups[0])
    opt.step()                                                                      # no gradient anywhere: nothing to do
    p.grad = torch.ones(8)
    with pytest.raises(NWHipError):
        opt.step()
    assert torch.equal(p.detach(), torch.zeros(8))                                  # ... and nothing was updated
