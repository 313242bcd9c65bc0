#!/bin/bash
# usage (GPU box, repo root): bash tools/wgrad_abl.sh -- nw_conv_wgrad_kernel on K4's large shapes, base and ablation builds
for shape in "42 224 56 56 128 1" "42 480 28 28 128 1" "42 128 56 56 32 3" "42 128 28 28 32 3" "42 992 14 14 128 1"; do
  for b in bench_wgrad bench_wgrad_NOMFMA bench_wgrad_NOLOAD bench_wgrad_NOCVT bench_wgrad_NOFRAG; do
    echo "$shape $b: $(./tools/$b $shape | tail -1)"
  done
done
