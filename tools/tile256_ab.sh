#!/bin/bash
# 8-wave 256-wide tiles vs the 4-wave tiles (GPU box)
for shape in "512 512 3 64" "512 512 3 32" "640 640 3 16" "1280 1280 3 8" "320 320 3 32" "1920 640 3 16" "2560 1280 3 8" "1280 1280 3 4" "320 2560 1 32" "1280 320 1 32" "640 5120 1 16" "2560 640 1 16"; do
  set -- $shape
  line="$1->$2 k$3 @$4:"
  for cfg in "default 0" "128x128 1" "256x128 1" "256x128 2" "256x128 3" "256x128 4" "256x128 6" "256x128 8" "256x256 1" "256x256 2" "256x256 4" "256x256 8" "256x256 12"; do
    c=($cfg)
    if [ ${c[0]} = default ]; then e=""; else e="MF_FORCE_TILE=${c[0]} MF_FORCE_SPLIT=${c[1]}"; fi
    r=$(env $e python tools/conv_probe.py --cin $1 --cout $2 --k $3 --pad $(($3/2)) --hw $4 --batch 8 --residual 0 --iters 30 2>/dev/null | grep "launch alone" | sed 's/.*alone: //; s/ us.*//')
    line="$line  ${c[0]}/s${c[1]}=$r"
  done
  echo "$line"
done
