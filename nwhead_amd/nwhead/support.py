"""Support-set bookkeeping with the reference's class names (nwhead/support.py:7-165): environment
split/merge, the training sampler and the evaluation bank with its inference modes.  The bank
(full_feat / full_y and everything derived from it) stays resident on the device it was computed on.
"""
import numpy as np
import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset, Subset

from .utils import (KNN, HNSW, DatasetMetadata, FeatureDataset, FullDataset,
                    InfiniteUniformClassLoader, compute_clusters)


class SupportSet:
    """Normalises the three accepted inputs (support.py:9-38): one dataset + env_array, a list of
    per-environment datasets, or one plain dataset (single environment 0)."""

    def __init__(self, support_set, n_classes, env_array=None):
        self.y_array = np.array(support_set.targets)
        self.n_classes = n_classes
        is_list = env_array is None and _is_dataset_list(support_set)
        if is_list:
            self.env_array = [e for e, ds in enumerate(support_set) for _ in range(len(ds))]
            wrapped = DatasetMetadata(support_set, self.env_array)
            self.env_datasets = wrapped
            self.env_map = {e: e for e in range(len(wrapped))}
            self.combined_dataset = ConcatDataset(wrapped)
            self.combined_dataset.targets = np.concatenate([env.targets for env in wrapped])
            assert len(self.combined_dataset) == len(self.combined_dataset.targets)
        else:
            self.env_array = env_array if env_array is not None else np.zeros(len(support_set))
            self.combined_dataset = DatasetMetadata(support_set, self.env_array)
            self.env_datasets = self._split_by_env(self.combined_dataset)

    def _split_by_env(self, combined):
        self.env_map, parts = {}, []
        for slot, env in enumerate(np.unique(self.env_array)):
            self.env_map[env] = slot
            rows = (np.asarray(self.env_array) == env).nonzero()[0]
            part = Subset(combined, rows)
            part.targets = self.y_array[rows]
            parts.append(part)
        return parts


def _is_dataset_list(obj):
    try:
        return all(isinstance(d, Dataset) for d in obj)
    except TypeError:
        return False


class SupportSetTrain(SupportSet):
    """Training-time sampler (support.py:58-93)."""

    def __init__(self, support_set, n_classes, train_type, n_shot, n_way=None, env_array=None):
        super().__init__(support_set, n_classes, env_array)
        self.train_type, self.n_shot, self.n_way = train_type, n_shot, n_way
        if train_type == 'random':
            self.train_iter = InfiniteUniformClassLoader(self.combined_dataset, n_shot, n_way)
        else:  # 'irm': one sampler per environment, one environment drawn per step
            self.train_iter = [InfiniteUniformClassLoader(env, n_shot) for env in self.env_datasets]

    def get_support(self, y):
        if self.train_type == 'irm':
            return np.random.choice(self.train_iter).next()
        return self.train_iter.next(y)


class SupportSetEval(SupportSet):
    """Evaluation bank and its six inference modes (support.py:95-165)."""

    MODES = ('random', 'full', 'cluster', 'ensemble', 'knn', 'hnsw')

    def __init__(self, support_set, n_classes, n_shot_random, n_shot_full, n_shot_cluster=3,
                 n_neighbors=20, env_array=None, cluster_backend="auto", loader_workers=0, pin_memory=False,
                 knn_per_query=False):
        super().__init__(support_set, n_classes, env_array)
        self.knn_per_query = bool(knn_per_query)      # utils.KNN(per_query=): each query its own neighbours (not in the reference)
        self.cluster_backend = cluster_backend        # utils.compute_clusters: 'auto' | 'sklearn' | 'device'
        self.n_shot_random, self.n_shot_full = n_shot_random, n_shot_full
        self.n_shot_cluster, self.n_neighbors = n_shot_cluster, n_neighbors
        self.full_datasets = [FullDataset(env, n_shot_full) for env in self.env_datasets]
        # the reference's loaders (support.py:164-165: batch 128, in order, num_workers = 0).  Decoding the bank's images is
        # what bounds precompute() on a real dataset, so the worker count and pinned staging buffers are the caller's to
        # raise (NWNet(..., loader_workers=, pin_memory=)); the row order does not depend on them (shuffle=False)
        self.loader_workers, self.pin_memory = int(loader_workers), bool(pin_memory)
        self.support_loaders = [DataLoader(ds, batch_size=128, shuffle=False, num_workers=self.loader_workers,
                                           pin_memory=self.pin_memory) for ds in self.full_datasets]

    def build_infer_iters(self, sfeat, sy, smeta, sfeat_env, sy_env, smeta_env):
        self.full_feat, self.full_y, self.full_meta = sfeat, sy, smeta
        self.full_feat_sep, self.full_y_sep, self.full_meta_sep = sfeat_env, sy_env, smeta_env
        dev = sfeat.device
        cf, cy = compute_clusters(sfeat, sy, self.n_shot_cluster, backend=self.cluster_backend)
        self.cluster_feat, self.cluster_y = cf.to(dev), cy.to(dev)
        self.random_iter = InfiniteUniformClassLoader(FeatureDataset(sfeat, sy, smeta), self.n_shot_random)
        self.knn = KNN(sfeat, sy, n_neighbors=self.n_neighbors, per_query=self.knn_per_query)
        self.hnsw = HNSW(sfeat, sy, n_neighbors=self.n_neighbors, per_query=self.knn_per_query)

    def get_support(self, mode, x=None):
        if mode not in self.MODES:
            raise NotImplementedError
        try:
            if mode == 'random':
                sfeat, sy, _ = self.random_iter.next()
                return sfeat, sy
            if mode == 'full':
                return self.full_feat, self.full_y
            if mode == 'cluster':
                return self.cluster_feat, self.cluster_y
            if mode == 'ensemble':
                return self.full_feat_sep, self.full_y_sep
            return (self.knn if mode == 'knn' else self.hnsw)(x)
        except AttributeError:
            raise AttributeError('Did you run precompute()?')
