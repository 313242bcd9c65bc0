#!/bin/bash
# usage (GPU box, repo root): bash tools/bn_dgrad_sweep.sh -- the norm1 backward of DenseNet-121's layer shapes (42 images), base and ablations
for b in ${BD_BINS:-bench_bn_dgrad bench_bn_dgrad_noepi}; do
  echo "== $b"
  for shape in "131712 64 256" "131712 224 256" "32928 480 512" "8232 512 1024" "8232 992 1024" "2058 992 1024"; do
    set -- $shape
    ./tools/$b $1 $2 128 $3
  done
done
