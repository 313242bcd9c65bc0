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


def load_checkpoint(network, path, optimizer=None, scheduler=None, verbose=True, trust_pickle=False):
    """Restores what is given; returns the checkpoint dictionary (epoch, extras) for the caller.
    Checkpoints are read with torch's weights-only unpickler (tensors, numbers, strings, containers: all that
    save_checkpoint writes); trust_pickle=True opts into the full unpickler for files from a trusted source
    that carry other objects."""
    if verbose:
        print("Loading checkpoint from", path)
    ckpt = torch.load(path, map_location=torch.device("cpu"), weights_only=not trust_pickle)
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


class ParseKwargs(argparse.Action):
    """`--flag a=1 b=0.5 c=true d=text` -> {'a': 1, 'b': 0.5, 'c': True, 'd': 'text'} (the reference's command line
    takes its wandb options this way, util/utils.py:87-102)."""

    def __call__(self, parser, namespace, values, option_string=None):
        out = {}
        for item in values:
            key, text = item.split("=")
            digits = text.replace("-", "")
            if digits.isnumeric():
                val = int(text)
            elif digits.replace(".", "").isnumeric():
                val = float(text)
            elif text in ("True", "true"):
                val = True
            elif text in ("False", "false"):
                val = False
            else:
                val = text
            out[key] = val
        setattr(namespace, self.dest, out)


def initialize_wandb(config):
    """Experiment logging to wandb is outside this implementation's scope: the name exists so
    that the reference's import line resolves; calling it says so."""
    raise NotImplementedError("wandb logging is not part of nwhead_amd (DESIGN.md section 7); run with logging off")
