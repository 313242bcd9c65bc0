"""Host-side index/sampler logic around the head (API of the reference's nwhead/utils.py).

No arithmetic lives here except KNN, which ranks with the HIP score kernel.  The samplers draw from
the GLOBAL ``np.random`` state with the same call sequence as the reference
(utils.py:93,129,136) so a seeded run picks the same supports.
"""
import numpy as np
import torch
from torch.utils.data import Dataset, default_collate

from .. import ops


def get_separated_indices(vals):
    """[0,1,1,2,3] -> [[0],[1,2],[3],[4]]; labels are ranked by sorted value (utils.py:142-159)."""
    if torch.is_tensor(vals):
        vals = vals.detach().cpu().numpy()
    vals = np.asarray(vals)
    uniq, inv = np.unique(vals, return_inverse=True)
    buckets = [[] for _ in range(len(uniq))]
    for pos, k in enumerate(inv.tolist()):
        buckets[k].append(pos)
    return buckets


class DatasetMetadata(Dataset):
    """(x, y) dataset -> (x, y, metadata[idx]) (utils.py:7-19)."""

    def __init__(self, dataset, metadata):
        self.dataset, self.metadata = dataset, metadata
        self.targets = dataset.targets

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        item = self.dataset[idx]
        return item[0], item[1], self.metadata[idx]


class FeatureDataset(Dataset):
    """Precomputed (feature, label, metadata) rows (utils.py:21-32)."""

    def __init__(self, features, targets, metadata):
        self.features, self.targets, self.metadata = features, targets, metadata

    def __len__(self):
        return len(self.features)

    def __getitem__(self, idx):
        return self.features[idx], self.targets[idx], self.metadata[idx]


class FullDataset(Dataset):
    """Class-balanced prefix of a dataset: the first min(n_shot_full, smallest class) items of every
    class, classes in ascending order -> the bank is class-sorted (utils.py:34-54)."""

    def __init__(self, underlying_dataset, n_shot_full):
        self.underlying_dataset = underlying_dataset
        self.indices = get_separated_indices(underlying_dataset.targets)
        keep = min(n_shot_full, min(len(ix) for ix in self.indices))
        self.keys = [i for ix in self.indices for i in ix[:keep]]

    def __len__(self):
        return len(self.keys)

    def __getitem__(self, key):
        return self.underlying_dataset[self.keys[key]]


class InfiniteUniformClassLoader:
    """n_shot random items from every class (or from n_way classes that include the query classes).

    Same draws as utils.py:99-140 for a given ``np.random`` state: one ``choice`` for the extra
    classes (probability 0 on the query classes), then one ``choice(row, n_shot, replace=False)``
    per selected class, query classes last and duplicates kept.
    """

    def __init__(self, dataset, n_shot, n_way=None):
        self.dataset = dataset
        self.indices = get_separated_indices(dataset.targets)
        self.n_classes = len(self.indices)
        self.n_shot, self.n_way = n_shot, n_way
        self.collate_fn = default_collate
        if n_way:
            assert n_way <= self.n_classes

    def __iter__(self):
        return self

    def __next__(self):
        raise NotImplementedError

    def next(self, qy=None):
        rows = self.indices
        if self.n_way:
            assert len(qy) <= self.n_way, "qy must be smaller than n_way"
            qy = qy.detach().cpu().numpy()
            p = np.ones(self.n_classes)
            p[qy] = 0
            p /= p.sum()
            extra = np.random.choice(self.n_classes, size=self.n_way - len(qy), replace=False, p=p)
            rows = [self.indices[c] for c in np.concatenate([extra, qy])]
        picks = np.array([np.random.choice(r, size=self.n_shot, replace=False) for r in rows]).flatten()
        return self.collate_fn([self.dataset[i] for i in picks])


class KNN:
    """Exact nearest supports by Euclidean distance (utils.py:178-193).  As in the reference, the k
    rows of ALL queries are concatenated into one shared (B*k, d) support -- every query attends to every query's
    neighbours.  per_query=True (not in the reference): each query gets ITS OWN k neighbours, supports
    (B, k, d) and labels (B, k) for the head's per-query path."""

    def __init__(self, data, labels, n_neighbors=20, per_query=False):
        self.data, self.labels, self.n_neighbors, self.per_query = data, labels, n_neighbors, bool(per_query)

    bank = None   # the data's ops.SplitBank when there is one (NWNet.precompute): scores from the split-fp16 tile kernel

    def indices(self, x):
        data = self.data.to(x.device)
        scores = ops.nw_scores(x, data, "euclidean", support_cache=self.bank if data.is_cuda else None)
        k = min(self.n_neighbors, scores.shape[1])
        # descending score == ascending distance; ties resolve like a stable argsort would
        if k <= 1024:
            return ops.nw_topk(scores, k)                    # radix select: the other N-k are never sorted
        return torch.argsort(scores, dim=-1, descending=True, stable=True)[:, :k]

    def __call__(self, x):
        idx = self.indices(x)
        if not self.per_query:
            idx = idx.reshape(-1)
        return self.data[idx.to(self.data.device)], self.labels[idx.to(self.labels.device)]


class HNSW(KNN):
    """The reference uses hnswlib's approximate index (utils.py:195-216); hnswlib is not part of
    this image, so 'hnsw' mode is served by the exact search above (a superset in recall)."""


def _class_argmax(values, inv, n_groups):
    """Row index of the largest value inside every group (lowest index on ties), values:(N,), inv:(N,) group ids."""
    N = values.shape[0]
    best = torch.full((n_groups,), -float("inf"), device=values.device).scatter_reduce(0, inv, values, "amax")
    rows = torch.arange(N, device=values.device)
    cand = torch.where(values >= best[inv], rows, torch.full_like(rows, N))
    return torch.full((n_groups,), N, dtype=torch.int64, device=values.device).scatter_reduce(0, inv, cand, "amin")


def kmeans_per_class_device(emb, lab, n_clusters, closest=False, max_iter=100):
    """Per-class k-means on the device, all classes at once (the reference runs sklearn's
    KMeans once per class on the host, utils.py:218-246).

    Lloyd iterations: the assignment step is ONE distance matrix points x all (class, cluster) centroids
    from the HIP scores kernel (ops.nw_scores) restricted to each point's own class; the update step is a
    one-hot matmul (deterministic, unlike atomics).  Initialisation is deterministic maximin seeding inside
    each class (first centre = the point nearest the class mean, every further one = the point farthest
    from the centres chosen so far), so results do not depend on a RNG; an emptied cluster keeps its centre.
    n_clusters == 1 is the class mean, which is also what sklearn returns.
    Returns (centroids (Cn*k, d) fp32 on emb's device, labels (Cn*k,) int64 on the host like the reference)."""
    from .. import ops
    emb = emb.detach().float().contiguous()
    lab = lab.detach().to(emb.device)
    classes, inv = torch.unique(lab, return_inverse=True)
    Cn, k, N = classes.numel(), int(n_clusters), emb.shape[0]
    rows = torch.arange(N, device=emb.device)

    def class_means():
        oh = torch.zeros(Cn, N, device=emb.device).index_put_((inv, rows), torch.ones((), device=emb.device))
        return (oh @ emb) / oh.sum(1, keepdim=True)

    cent = class_means()[:, None, :]                                   # (Cn, 1, d)
    if k > 1:
        def dist_to(one_per_class):                                    # (Cn, d) -> (N,) distance to own class's row
            return -ops.nw_scores(emb, one_per_class.contiguous())[rows, inv]
        chosen = [emb[_class_argmax(-dist_to(cent[:, 0]), inv, Cn)]]
        dmin = dist_to(chosen[0])
        for _ in range(1, k):
            chosen.append(emb[_class_argmax(dmin, inv, Cn)])
            dmin = torch.minimum(dmin, dist_to(chosen[-1]))
        cent = torch.stack(chosen, dim=1)                              # (Cn, k, d)
    assign = None
    for _ in range(max_iter if k > 1 else 0):
        sc = ops.nw_scores(emb, cent.reshape(Cn * k, -1).contiguous())  # (N, Cn*k) = -distance
        own = sc.view(N, Cn, k)[rows, inv]                             # (N, k): this point's class only
        new_assign = own.argmax(1)
        if assign is not None and torch.equal(new_assign, assign):
            break
        assign = new_assign
        oh = torch.zeros(Cn * k, N, device=emb.device).index_put_((inv * k + assign, rows), torch.ones((), device=emb.device))
        cnt = oh.sum(1, keepdim=True)
        upd = (oh @ emb) / cnt.clamp_min(1)
        cent = torch.where(cnt.view(Cn, k, 1) > 0, upd.view(Cn, k, -1), cent)
    cent = cent.reshape(Cn * k, -1).contiguous()
    if closest:                                                        # nearest actual point of the class
        own = ops.nw_scores(emb, cent).view(N, Cn, k)[rows, inv]       # (N, k)
        pick = torch.stack([_class_argmax(own[:, j].contiguous(), inv, Cn) for j in range(k)], dim=1)
        cent = emb[pick.reshape(-1)].contiguous()
    return cent, classes.repeat_interleave(k).cpu()


def compute_clusters(embeddings, labels, n_clusters, closest=False, backend="auto"):
    """Per-class k-means centroids (utils.py:218-246).
    backend 'sklearn': the reference's own call (KMeans(random_state=0) per class on the host) -- what
    pins parity for n_clusters > 1, where the optimum found depends on sklearn's seeding.
    backend 'device': kmeans_per_class_device (HIP distance kernel, deterministic seeding).
    backend 'auto' (default): the device whenever the features are on the GPU (n_clusters == 1: the class mean either way,
    the same centroids; n_clusters > 1 -- round 4 --: another local optimum than sklearn's seeding finds, of the same
    quality: tests/test_kmeans_gpu.py holds its inertia against sklearn's; no host round trip of the bank and 13.8 ms
    instead of 13.5 s on the K3 bank), sklearn for host tensors."""
    if backend not in ("auto", "sklearn", "device"):
        raise ValueError(backend)
    if backend == "device" or (backend == "auto" and embeddings.is_cuda):
        return kmeans_per_class_device(embeddings, labels, n_clusters, closest)
    from sklearn.cluster import KMeans
    emb = embeddings.detach().cpu()
    lab = labels.detach().cpu()
    feats, ys = [], []
    for c in np.unique(lab.numpy()):
        rows = emb[lab == c]
        km = KMeans(n_clusters=n_clusters, random_state=0).fit(rows.numpy())
        cent = torch.tensor(km.cluster_centers_).float()
        if closest:
            cent = rows[torch.cdist(cent, rows).argmin(dim=-1)]
        feats.append(cent)
        ys += [c] * n_clusters
    return torch.cat(feats, dim=0), torch.tensor(ys)
