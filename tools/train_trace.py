"""`python -m nwhead_amd.train --arch ARCH` for one short epoch on synthetic images, as a script (for a rocprofv3 kernel trace: which
kernels does the harness's training + validation run?).  usage: python tools/train_trace.py [arch] [image size]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.train import main
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
size = sys.argv[2] if len(sys.argv) > 2 else "64"
with tempfile.TemporaryDirectory() as d:
    main(["--models_dir", d, "--arch", arch, "--dataset", "synthetic", "--num_epochs", "1", "--num_steps_per_epoch", "8",
          "--num_val_steps_per_epoch", "2", "--batch_size", "32", "--synthetic_size", size, "--synthetic_per_class", "12"])
