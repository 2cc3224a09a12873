#!/bin/bash
# the 128 x 80 producer-wave tile on the UNet's outer-level shapes at batch 8 (320 channels over 8192 pixels: 256 workgroups instead of 320): correctness
# against torch fp64, then its time beside the shipped table's pick, cold-cache like the tuner measures.   usage: tools/tile80_check.sh [tag]
TAG=${1:-tile80}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for shape in "--cin 320 --cout 320 --k 3 --hw 32" "--cin 640 --cout 320 --k 3 --hw 32 --residual 0" "--cin 960 --cout 320 --k 3 --hw 32 --residual 0" "--cin 1280 --cout 320 --k 1 --pad 0 --hw 32 --residual 0" "--cin 320 --cout 320 --k 1 --pad 0 --hw 32" "--cin 72 --cout 160 --k 3 --hw 13 --residual 0" "--cin 640 --cout 640 --k 3 --hw 16"; do
  for cfg in "shipped" "128x64 1 3" "128x80 1 3" "128x80 1 4" "128x80 2 3"; do
    set -- $cfg
    echo "== $cfg: $shape"
    if [ "$1" = shipped ]; then timeout 300 python tools/conv_probe.py $shape --batch 8 --check 1 --iters 30 2>&1 | tail -3
    else MF_FORCE_TILE=$1 MF_FORCE_SPLIT=$2 MF_FORCE_LD=$3 timeout 300 python tools/conv_probe.py $shape --batch 8 --check 1 --iters 30 2>&1 | tail -3; fi
  done
done
} > gpurun_out/${TAG}_check.txt 2>&1
cat gpurun_out/${TAG}_check.txt
