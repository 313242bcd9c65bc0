"""NWNet / NWHead with the reference's constructor and method surface (nwhead/nw.py:11-289), the
head computed by the HIP kernels in nwhead_amd/csrc/ through ops.nw_head.

Differences that are deliberate (and documented in DESIGN.md):
  * the precomputed bank (full_feat/full_y) stays on the device instead of round-tripping through
    host memory every predict() (reference: nw.py:156,226);
  * 'hnsw' mode is an exact search (hnswlib is not available);
  * there is no CPU execution path: tensors must live on the MI355X.
"""
import torch
import torch.nn as nn

from .. import ops
from .kernel import _ScoreModule, get_kernel
from .support import SupportSetEval, SupportSetTrain


class _ChannelsLast(nn.Module):
    """An inference copy of a backbone kept in channels_last layout; 4-D inputs are converted on the way in."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner.to(memory_format=torch.channels_last)

    def forward(self, x):
        return self.inner(x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x)


class NWHead(nn.Module):
    """forward(x:(B,d), sx:(N,d)|(B,N,d), sy:(N,)|(B,N)) -> (B, n_classes) log-probabilities."""

    def __init__(self, kernel, n_classes, validate_labels=False):
        super().__init__()
        self.kernel = kernel
        self.n_classes = n_classes
        # not in the reference: its F.one_hot (nw.py:276) refuses support labels >= n_classes; here that check costs a device
        # round trip per call, so it is an option (on under NWNet(debug_mode=True)); without it such supports are skipped
        self.validate_labels = bool(validate_labels)

    def forward(self, x, sx, sy, return_weights=False, support_norm2=None, support_cache=None):
        if not isinstance(self.kernel, _ScoreModule):
            # any other callable kernel module, as the reference accepts (nw.py:256-264, :277-283): its scores (torch
            # ops on the device, with their autograd history), then the softmax / aggregation / log tail in HIP
            if return_weights:
                raise NotImplementedError("return_weights needs one of the built-in score functions (get_kernel)")
            b = len(x)
            sxe = sx[None].expand(b, *sx.shape) if sx.dim() == x.dim() else sx       # nw.py:277-279
            scores = self.kernel(x.unsqueeze(1), sxe).squeeze(1)                      # nw.py:281-283
            return ops.nw_aggregate(scores, sy, self.n_classes)
        return ops.nw_head(x, sx, sy, self.n_classes, self.kernel.kind, self.kernel._logit_scale(),
                           return_weights=return_weights, support_norm2=support_norm2,
                           support_cache=support_cache, validate_labels=self.validate_labels)


class NWNet(nn.Module):
    def __init__(self, featurizer, n_classes, support_dataset=None, feat_dim=None, proj_dim=0,
                 kernel_type='euclidean', train_type='random', n_way=None, n_shot=1,
                 n_shot_random=1, n_shot_full=100, n_shot_cluster=1, n_neighbors=10,
                 env_array=None, debug_mode=False, device='cuda:0', return_mask=False, cluster_backend='auto',
                 loader_workers=0, pin_memory=False, knn_per_query=False):
        super().__init__()
        self.knn_per_query = bool(knn_per_query)  # not in the reference: 'knn' / 'hnsw' modes give every query ITS OWN neighbours
        self.cluster_backend = cluster_backend   # not in the reference: where 'cluster' mode's k-means runs (utils.compute_clusters)
        # not in the reference either (its bank loaders are single-process, support.py:164-165): DataLoader workers and
        # pinned staging for the loaders precompute() featurises the bank from; same row order whatever they are
        self.loader_workers, self.pin_memory = int(loader_workers), bool(pin_memory)
        if support_dataset is not None:
            assert hasattr(support_dataset, 'targets'), 'Support set must have .targets attribute'
        if proj_dim > 0:
            assert feat_dim is not None, 'Feature dimension must be specified'
            featurizer = nn.Sequential(featurizer, nn.Linear(feat_dim, proj_dim))
        self.featurizer = featurizer
        self.n_classes, self.train_type, self.n_way = n_classes, train_type, n_way
        self.n_shot, self.n_shot_random, self.n_shot_full = n_shot, n_shot_random, n_shot_full
        self.n_shot_cluster, self.n_neighbors = n_shot_cluster, n_neighbors
        self.env_array, self.debug_mode = env_array, debug_mode
        self.device, self.return_mask = device, return_mask
        # registered twice on purpose: reference state_dicts carry both kernel.* and nwhead.kernel.*
        self.kernel = get_kernel(kernel_type)
        self.nwhead = NWHead(self.kernel, n_classes, validate_labels=bool(debug_mode))
        if support_dataset is not None:
            self.support_train = SupportSetTrain(support_dataset, n_classes, train_type, n_shot,
                                                 n_way=n_way, env_array=env_array)
            self.process_support_eval(support_dataset)

    # ------------------------------------------------------------------ eval-mode BatchNorm folding
    def enable_bn_folding(self, on=True, channels_last=True):
        """Inference (precompute / predict / get_neighbors, featurizer in eval mode) then runs a copy of the
        featurizer whose conv -> BatchNorm pairs are single convolutions (model.fold_batchnorm, SURVEY 8f
        N1), kept in channels_last layout (MIOpen's faster fp32 path: ResNet-18 over 64 images @224
        3.81 -> 3.55 ms folded -> 3.22 ms folded + channels_last, features equal to 3e-7 relative).  The copy
        is rebuilt by precompute() and dropped by train(); it is not part of state_dict()."""
        object.__setattr__(self, '_fold_bn', bool(on))
        object.__setattr__(self, '_fold_cl', bool(channels_last))
        object.__setattr__(self, '_folded', None)

    def _weights_signature(self):
        """Changes whenever a parameter or buffer of the featurizer is written in place (optimizer step,
        load_state_dict, BatchNorm statistics) or replaced: the folded inference copy is rebuilt then.  (torch's fused
        optimizers do not advance version counters; train(), which every training loop passes through, drops the copy.)"""
        ts = list(self.featurizer.parameters()) + list(self.featurizer.buffers())
        return (len(ts), sum(0 if t.is_inference() else t._version for t in ts), sum(t.data_ptr() & 0xffff for t in ts))

    def _eval_featurizer(self, rebuild=False):
        if not getattr(self, '_fold_bn', False) or self.featurizer.training:
            return self.featurizer
        sig = self._weights_signature()
        if rebuild or getattr(self, '_folded', None) is None or getattr(self, '_folded_sig', None) != sig:
            object.__setattr__(self, '_folded_sig', sig)
            from ..model import fold_batchnorm
            from ..model.backbones import Conv1x1Fused, Conv3x3Fused, ScaleShiftReLU
            folded = fold_batchnorm(self.featurizer)
            # DenseNet's folded copy runs its BatchNorm -> ReLU pairs in an NCHW HIP kernel on channel
            # prefixes of the dense-block slab: it stays NCHW (11.2 -> 7.8 ms over 64 images @224)
            if getattr(self, '_fold_cl', False) and not any(isinstance(m, (ScaleShiftReLU, Conv1x1Fused, Conv3x3Fused)) for m in folded.modules()):
                folded = _ChannelsLast(folded)
            object.__setattr__(self, '_folded', folded)
        return self._folded

    def train(self, mode=True):
        if mode:                                         # the weights are about to change
            object.__setattr__(self, '_folded', None)
            self.sharded_bank = None                     # a shard featurised with the old weights must not serve 'full'
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        object.__setattr__(self, '_folded', None)
        self.sharded_bank = None
        return super().load_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------ evaluation bank
    def process_support_eval(self, support_dataset):
        self.support_eval = SupportSetEval(support_dataset, self.n_classes, self.n_shot_random,
                                           self.n_shot_full, n_shot_cluster=self.n_shot_cluster,
                                           n_neighbors=self.n_neighbors, env_array=self.env_array,
                                           knn_per_query=self.knn_per_query,
                                           cluster_backend=self.cluster_backend, loader_workers=self.loader_workers,
                                           pin_memory=self.pin_memory)

    @torch.no_grad()
    def _compute_all_support_feats(self):
        """Featurise every environment's balanced bank in loader order; rows stay on self.device."""
        per_env = []
        featurizer = self._eval_featurizer(rebuild=True)
        for loader in self.support_eval.support_loaders:
            f, y, m = [], [], []
            for img, label, meta in loader:
                f.append(featurizer(img.to(self.device, non_blocking=self.pin_memory)).detach())
                y.append(label.to(self.device))
                m.append(meta.to(self.device))
            per_env.append((torch.cat(f), torch.cat(y), torch.cat(m)))
        feats, labels, meta = (torch.cat([e[k] for e in per_env]) for k in range(3))
        return (feats, labels, meta, [e[0] for e in per_env], [e[1] for e in per_env],
                [e[2] for e in per_env])

    def precompute(self):
        assert not self.featurizer.training
        info = self._compute_all_support_feats()
        self.sharded_bank = None                     # predict('full') serves the bank built here
        self.full_feat, self.full_y = info[0], info[1]
        self.full_cache = ops.SplitBank(self.full_feat, labels=self.full_y)   # norms + split-fp16 rows for predict('full')
        self.full_norm2 = self.full_cache.norm2
        self.support_eval.build_infer_iters(*info)
        self.support_eval.knn.bank = self.support_eval.hnsw.bank = self.full_cache   # neighbour search over the same bank

    @torch.no_grad()
    def precompute_sharded(self, group=None, partial_fn=None, merge_fn=None):
        """'full' inference over a bank sharded across the ranks of `group`: this
        rank featurises ONLY rows [lo, hi) of the balanced, class-sorted bank -- the row order of
        precompute() (environments in order, support.py loader order inside) -- and keeps them resident
        as a ShardedBank; the bank is never gathered.  predict(x, 'full') then exchanges one packed
        partial per query batch.  The other inference modes still need precompute().
        partial_fn / merge_fn: compute hooks for the CPU tests (see sharded.ShardedBank)."""
        import torch.distributed as dist
        from torch.utils.data import DataLoader, Subset
        from ..sharded import ShardedBank, shard_bounds
        assert not self.featurizer.training
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        datasets = self.support_eval.full_datasets
        lo, hi = shard_bounds(sum(len(ds) for ds in datasets), world, rank)
        featurizer = self._eval_featurizer(rebuild=True)
        feats, labels, start = [], [], 0
        for ds in datasets:                                   # this rank's rows of every environment
            a, b = max(lo, start) - start, min(hi, start + len(ds)) - start
            start += len(ds)
            if a >= b:
                continue
            for img, label, _meta in DataLoader(Subset(ds, range(a, b)), batch_size=128, shuffle=False,
                                                num_workers=self.loader_workers, pin_memory=self.pin_memory):
                feats.append(featurizer(img.to(self.device, non_blocking=self.pin_memory)).detach())
                labels.append(label.to(self.device))
        d = feats[0].shape[1] if feats else featurizer(ds[0][0][None].to(self.device)).shape[1]
        feat = torch.cat(feats) if feats else torch.empty(0, d, device=self.device)
        y = torch.cat(labels) if labels else torch.empty(0, dtype=torch.int64, device=self.device)
        self.sharded_bank = ShardedBank(feat, y, self.n_classes, self.kernel.kind, self.kernel._logit_scale(),
                                        group=group, partial_fn=partial_fn, merge_fn=merge_fn)
        return self.sharded_bank

    def predict(self, x, mode='random'):
        qfeat = self._eval_featurizer()(x)
        if mode == 'full' and getattr(self, 'sharded_bank', None) is not None:
            out = self.sharded_bank.predict(qfeat.detach())
            return (out, torch.full((len(x),), True)) if self.return_mask else out
        sfeat, sy = self.support_eval.get_support(mode, x=qfeat)
        if self.debug_mode:
            print('qx shape:', x.shape)
            print('sfeat shape:', [f.shape for f in sfeat] if mode == 'ensemble' else sfeat.shape)
            print('sy:', sy)
        if mode == 'ensemble':
            probs = sum(self.nwhead(qfeat, f.to(x.device), y.to(x.device)).exp() for f, y in zip(sfeat, sy))
            out = torch.log(probs / len(sfeat))
        elif mode == 'full':
            sfeat, sy = sfeat.to(x.device), sy.to(x.device)
            cache = getattr(self, 'full_cache', None)
            if cache is None or not cache.matches(sfeat):   # the bank was replaced or updated since precompute()
                cache = self.full_cache = ops.SplitBank(sfeat, labels=sy)
            out = self.nwhead(qfeat, sfeat, sy, support_cache=cache)
        else:
            out = self.nwhead(qfeat, sfeat.to(x.device), sy.to(x.device))
        if self.return_mask:
            return out, torch.full((len(x),), True)
        return out

    def get_neighbors(self, x):
        """Support indices ordered from nearest to farthest under the configured kernel."""
        qfeat = self._eval_featurizer()(x).detach()
        scores = self.kernel(qfeat, self.full_feat.to(qfeat.device))
        return torch.argsort(scores, dim=-1, descending=True)

    # ------------------------------------------------------------------ training step
    def forward(self, x, y, metadata=None, support_data=None):
        sx, sy, sm = support_data if support_data is not None else self.support_train.get_support(y)
        if sm is None:
            sm = torch.zeros_like(sy)
        sx, sy, sm = sx.to(x.device), sy.to(x.device), sm.to(x.device)
        nq = len(x)
        feats = self.featurizer(torch.cat((x, sx), dim=0))   # joint pass: BN statistics are shared
        qfeat, sfeat = feats[:nq], feats[nq:]
        isin = torch.isin(y, sy)
        if self.debug_mode:
            print('qx shape:', x.shape, 'sx shape:', sx.shape)
            print('qfeat shape:', qfeat.shape, 'sfeat shape:', sfeat.shape)
            print('qy:', y, 'sy:', sy, 'qy in sy:', isin)
            print(f'Percent query dropped: {(1.0 - isin.float().mean().item())*100}%')
            if metadata is not None:
                print('qmeta:', metadata, 'smeta:', sm)
        out = self.nwhead(qfeat, sfeat, sy)
        return (out, isin) if self.return_mask else out
