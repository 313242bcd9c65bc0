"""Checkpoints and small helpers of the training harness.  File names and dictionary keys are the
reference's (util/utils.py:33-71): `model.{epoch:04d}.h5` holding epoch / network_state_dict /
optimizer / scheduler, plus `model.best.h5` -- so checkpoints move between the two code bases.  The
reference can only save; `latest_checkpoint` + `load_checkpoint` give the missing resume."""
import argparse
import glob
import os
import re
import shutil

import torch


def summary(network):
    names = [n for n, p in network.named_parameters() if p.requires_grad]
    total = sum(p.numel() for p in network.parameters() if p.requires_grad)
    print(network)
    print("trainable tensors: %d, parameters: %d" % (len(names), total))
    return total


def save_checkpoint(epoch, network, optimizer, model_folder, scheduler=None, is_best=False, extra=None):
    state = {"epoch": epoch, "network_state_dict": network.state_dict(), "optimizer": optimizer.state_dict()}
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    if extra:
        state.update(extra)
    os.makedirs(model_folder, exist_ok=True)
    path = os.path.join(model_folder, "model.%04d.h5" % epoch)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(model_folder, "model.best.h5"))
    return path


def load_checkpoint(network, path, optimizer=None, scheduler=None, verbose=True):
    """Restores what is given; returns the checkpoint dictionary (epoch, extras) for the caller."""
    if verbose:
        print("Loading checkpoint from", path)
    ckpt = torch.load(path, map_location=torch.device("cpu"), weights_only=False)
    network.load_state_dict(ckpt["network_state_dict"])
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    if scheduler is not None and "scheduler" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler"])
    return ckpt


def latest_checkpoint(model_folder):
    """Path of the numbered checkpoint with the highest epoch in `model_folder`, or None."""
    best = None
    for p in glob.glob(os.path.join(model_folder, "model.*.h5")):
        m = re.search(r"model\.(\d+)\.h5$", p)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), p)
    return None if best is None else best[1]


def parse_bool(v):
    if v.lower() == "true":
        return True
    if v.lower() == "false":
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")
