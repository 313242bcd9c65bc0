#!/bin/bash
# usage (GPU box, repo root): bash tools/conv_cfg_sweep.sh   -- the 1x1 bottleneck layers of DenseNet-121 (42 images) under each tile shape
B=./tools/bench_conv_plain
export NW_BC_MOMENTS=1 NW_BC_PRE=1
for shape in "56 64" "56 128" "56 256" "28 128" "28 256" "28 512" "14 256" "14 512" "14 1024" "7 512" "7 1024"; do
  set -- $shape
  for f in 0 1 2 3 4; do
    r=$(NW_CONV_FORCE_CFG=$f $B 42 $2 $1 $1 128 1 0 | tail -1)
    echo "hw $1 cin $2 cfg $f: $r"
  done
done
