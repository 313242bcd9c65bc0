"""Training / evaluation harness around NWNet on the MI355X: `python -m nwhead_amd.train --dataset synthetic ...`.

The build's counterpart of the reference's train.py, written against the same NWNet
surface and reproducing its epoch order (train.py:287-303):

    eval():  precompute() -> predict(mode) over the validation set for random, full, cluster
    train(): network(img, label) -> NLLLoss -> backward -> SGD(momentum 0.9, nesterov, weight decay)
    MultiStepLR.step(), checkpoint every `log_interval` epochs (+ model.best.h5 by 'full' accuracy)

and its per-step result dictionary (train.py:400-422: loss, acc in percent, batch_size, prob, gt).
Not carried over: torchvision datasets/transforms (see nwhead_amd/data.py), wandb, the FC baseline.
Added: `--resume` (the reference saves optimizer and scheduler state but never loads it back).
"""
import argparse
import json
import os
import random

import numpy as np
import torch
import torch.nn as nn

from .data import build_datasets
from .model import load_model
from .nwhead.nw import NWNet
from .util import metric
from .util.metric import ECELoss, Metric
from .util.utils import latest_checkpoint, load_checkpoint, save_checkpoint

EVAL_MODES = ("random", "full", "cluster")


class TinyNet(nn.Module):
    """Three conv blocks -> 64 features: for smoke tests of the harness, not a reference architecture."""

    def __init__(self, feat_dim=64):
        super().__init__()
        chans = (3, 16, 32, feat_dim)
        self.body = nn.Sequential(*[blk for i in range(3) for blk in (
            nn.Conv2d(chans[i], chans[i + 1], 3, padding=1, bias=False), nn.BatchNorm2d(chans[i + 1]),
            nn.ReLU(inplace=True), nn.MaxPool2d(2))])

    def forward(self, x):
        return self.body(x).mean(dim=(2, 3))


def build_featurizer(arch, small_images):
    if arch == "tiny":
        return TinyNet(64), 64
    if arch == "resnet18":
        return load_model("CIFAR_ResNet18" if small_images else "resnet18"), 512
    if arch == "densenet121":
        return load_model("CIFAR_DenseNet121" if small_images else "densenet121"), 1024
    raise NotImplementedError(arch)


def make_parser():
    p = argparse.ArgumentParser(description="NW head training on MI355X")
    p.add_argument("--models_dir", default="./runs")
    p.add_argument("--data_dir", default="./")
    p.add_argument("--dataset", default="synthetic", help="synthetic | cifar10 | cifar100")
    p.add_argument("--arch", default="resnet18", help="resnet18 | densenet121 | tiny")
    p.add_argument("--gpu_id", type=int, default=0)
    p.add_argument("--workers", type=int, default=0)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--num_epochs", type=int, default=200)
    p.add_argument("--num_steps_per_epoch", type=int, default=10_000_000)
    p.add_argument("--num_val_steps_per_epoch", type=int, default=10_000_000)
    p.add_argument("--scheduler_milestones", nargs="+", type=int, default=(100, 150))
    p.add_argument("--scheduler_gamma", type=float, default=0.1)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--log_interval", type=int, default=25, help="checkpoint every this many epochs")
    p.add_argument("--kernel_type", default="euclidean")
    p.add_argument("--proj_dim", type=int, default=0)
    p.add_argument("--n_shot", type=int, default=1)
    p.add_argument("--n_way", type=int, default=None)
    p.add_argument("--freeze_featurizer", action="store_true")
    p.add_argument("--resume", action="store_true", help="continue from the newest checkpoint of this run")
    p.add_argument("--no_fold_bn", action="store_true",
                   help="evaluate with the training featurizer instead of its folded inference copy (model.fold_batchnorm)")
    # synthetic dataset shape
    p.add_argument("--synthetic_classes", type=int, default=10)
    p.add_argument("--synthetic_per_class", type=int, default=20)
    p.add_argument("--synthetic_size", type=int, default=32)
    p.add_argument("--synthetic_noise", type=float, default=0.35)
    return p


class Trainer:
    def __init__(self, args):
        self.args = args
        if args.seed > 0:
            random.seed(args.seed)
            np.random.seed(args.seed)
            torch.manual_seed(args.seed)
        if not torch.cuda.is_available():
            raise RuntimeError("nwhead_amd has no CPU execution path: an MI355X is required")
        self.device = torch.device("cuda:%d" % args.gpu_id)
        self.run_dir = os.path.join(args.models_dir, "nwhead_%s_%s_lr%g_bs%d_nshot%d_nway%s_seed%d" % (
            args.dataset, args.arch, args.lr, args.batch_size, args.n_shot, args.n_way, args.seed))
        self.ckpt_dir = os.path.join(self.run_dir, "checkpoints")
        os.makedirs(self.ckpt_dir, exist_ok=True)
        with open(os.path.join(self.run_dir, "args.txt"), "w") as fh:
            json.dump({k: v for k, v in vars(args).items()}, fh, indent=2, default=str)

        train_ds, val_ds = build_datasets(args.dataset, args.data_dir, args.synthetic_size,
                                          args.synthetic_per_class, args.synthetic_classes, args.seed,
                                          args.synthetic_noise)
        self.num_classes = train_ds.num_classes
        loader = torch.utils.data.DataLoader
        self.train_loader = loader(train_ds, batch_size=args.batch_size, shuffle=True, num_workers=args.workers)
        self.val_loader = loader(val_ds, batch_size=args.batch_size, shuffle=False, num_workers=args.workers)

        small = args.dataset in ("cifar10", "cifar100") or (args.dataset == "synthetic" and args.synthetic_size <= 64)
        featurizer, feat_dim = build_featurizer(args.arch, small)
        if args.freeze_featurizer:
            for p in featurizer.parameters():
                p.requires_grad = False
        self.network = NWNet(featurizer, self.num_classes, support_dataset=train_ds, feat_dim=feat_dim,
                             proj_dim=args.proj_dim, kernel_type=args.kernel_type, n_shot=args.n_shot,
                             n_way=args.n_way, device=str(self.device)).to(self.device)
        self.network.enable_bn_folding(not args.no_fold_bn)   # precompute()/predict() in the eval epochs
        self.criterion = nn.NLLLoss()
        # the reference's optimizer (train.py:243-247); on the GPU the same update in nw_sgd_step_f32 (optim.SGD: all parameter
        # tensors in a few launches, 0.05 ms for DenseNet-121 where torch's foreach kernels take 0.91 and its fused ones 0.34)
        if self.device.type == "cuda":
            from .optim import SGD
            self.optimizer = SGD(self.network.parameters(), lr=args.lr, momentum=0.9, weight_decay=args.weight_decay, nesterov=True)
        else:
            self.optimizer = torch.optim.SGD(self.network.parameters(), lr=args.lr, momentum=0.9,
                                             weight_decay=args.weight_decay, nesterov=True)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(args.scheduler_milestones),
                                                              gamma=args.scheduler_gamma)
        self.metrics = {k: Metric() for k in ("loss:train", "acc:train")}
        self.val_metrics = {f"{m}:val:{mode}": Metric() for m in ("loss", "acc", "ece") for mode in EVAL_MODES}
        self.start_epoch, self.best_acc1 = 1, 0.0
        if args.resume:
            path = latest_checkpoint(self.ckpt_dir)
            if path is not None:
                ckpt = load_checkpoint(self.network, path, self.optimizer, self.scheduler)
                self.start_epoch = ckpt["epoch"] + 1
                self.best_acc1 = ckpt.get("best_acc1", 0.0)
        self.history = []

    # ------------------------------------------------------------------ one step
    def nw_step(self, batch, is_train=True, mode="random"):
        img, label = batch
        img, label = img.float().to(self.device), label.to(self.device)
        self.optimizer.zero_grad()
        with torch.set_grad_enabled(is_train):
            output = self.network(img, label) if is_train else self.network.predict(img, mode)
            loss = self.criterion(output, label)
            if is_train:
                loss.backward()
                self.optimizer.step()
            acc = metric.acc(output.argmax(-1), label)
        return {"loss": loss.detach().cpu().numpy(), "acc": acc * 100, "batch_size": len(img),
                "prob": output.detach().exp(), "gt": label}

    # ------------------------------------------------------------------ epochs
    def train_epoch(self):
        self.network.train()
        for i, batch in enumerate(self.train_loader):
            res = self.nw_step(batch, is_train=True)
            self.metrics["loss:train"].update_state(res["loss"], res["batch_size"])
            self.metrics["acc:train"].update_state(res["acc"], res["batch_size"])
            if i == self.args.num_steps_per_epoch:
                break

    def eval_epoch(self, mode):
        self.network.eval()
        probs, gts = [], []
        for i, batch in enumerate(self.val_loader):
            res = self.nw_step(batch, is_train=False, mode=mode)
            self.val_metrics[f"loss:val:{mode}"].update_state(res["loss"], res["batch_size"])
            self.val_metrics[f"acc:val:{mode}"].update_state(res["acc"], res["batch_size"])
            probs.append(res["prob"])
            gts.append(res["gt"])
            if i == self.args.num_val_steps_per_epoch:
                break
        ece = (ECELoss()(torch.cat(probs), torch.cat(gts)) * 100).item()
        self.val_metrics[f"ece:val:{mode}"].update_state(ece, 1)
        return self.val_metrics[f"acc:val:{mode}"].result()

    def fit(self):
        a = self.args
        for epoch in range(self.start_epoch, a.num_epochs + 1):
            self.network.eval()
            self.network.precompute()
            accs = {mode: self.eval_epoch(mode) for mode in EVAL_MODES}
            acc1 = accs["full"]
            self.train_epoch()
            self.scheduler.step()
            is_best = acc1 > self.best_acc1
            self.best_acc1 = max(acc1, self.best_acc1)
            if epoch % a.log_interval == 0:
                save_checkpoint(epoch, self.network, self.optimizer, self.ckpt_dir, self.scheduler, is_best=is_best,
                                extra={"best_acc1": self.best_acc1})
            row = {"epoch": epoch, "lr": self.scheduler.get_last_lr()[0]}
            row.update({k: m.result() for k, m in self.metrics.items()})
            row.update({k: m.result() for k, m in self.val_metrics.items()})
            self.history.append(row)
            print("epoch %d: train loss %.4f acc %.2f | val acc random %.2f full %.2f cluster %.2f | lr %g" % (
                epoch, row["loss:train"], row["acc:train"], row["acc:val:random"], row["acc:val:full"],
                row["acc:val:cluster"], row["lr"]), flush=True)
            for m in list(self.metrics.values()) + list(self.val_metrics.values()):
                m.reset_state()
        return self.history


def main(argv=None):
    trainer = Trainer(make_parser().parse_args(argv))
    return trainer.fit()


if __name__ == "__main__":
    main()
