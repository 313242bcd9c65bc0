#!/bin/bash
# usage (GPU box, repo root): bash tools/k4_ab_knob.sh "ENV=VAL" [rounds] -- K4 steps with and without one environment switch, alternated
kv=$1; n=${2:-3}
for i in $(seq $n); do
  echo -n "with $kv: "; env $kv timeout -k 10 200 python tools/k4_step.py 20 | tail -1
  echo -n "default: "; timeout -k 10 200 python tools/k4_step.py 20 | tail -1
done
