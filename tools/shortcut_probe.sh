#!/bin/bash
# the VAE decoder's two 1x1 shortcut convs (streaming GEMMs: K = 256 / 512, half a million pixels) under every tile / operand path   usage: tools/shortcut_probe.sh
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for shape in "--cin 256 --cout 128 --k 1 --pad 0 --hw 256 --residual 0" "--cin 512 --cout 256 --k 1 --pad 0 --hw 128 --residual 0"; do
  for cfg in "shipped" "256x128 1 0" "128x128 1 0" "128x128 1 2" "128x128 1 3" "128x128 1 4" "128x64 1 2" "128x64 1 3" "128x64 1 4" "64x64 1 2" "64x64 1 3" "256x256 1 0"; do
    set -- $cfg
    echo "== $cfg: $shape"
    if [ "$1" = shipped ]; then timeout 300 python tools/conv_probe.py $shape --batch 8 --iters 20 2>&1 | tail -2
    else MF_FORCE_TILE=$1 MF_FORCE_SPLIT=$2 MF_FORCE_LD=$3 timeout 300 python tools/conv_probe.py $shape --batch 8 --iters 20 2>&1 | tail -2; fi
  done
done
} > gpurun_out/shortcut_probe.txt 2>&1
grep -E "^==|alone" gpurun_out/shortcut_probe.txt
