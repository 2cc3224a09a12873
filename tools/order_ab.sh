#!/bin/bash
# A/B of the XCD tile order on UNet shapes (GPU box)
for shape in "640 640 3 16" "1280 1280 3 8" "320 320 3 32" "1920 640 3 16" "2560 1280 3 8" "320 2560 1 32" "1280 320 1 32" "640 5120 1 16" "2560 640 1 16" "1280 1280 3 4"; do
  set -- $shape
  for o in n m; do
    r=$(MF_TILE_ORDER=$o python tools/conv_probe.py --cin $1 --cout $2 --k $3 --pad $(($3/2)) --hw $4 --batch 8 --residual 0 --iters 30 | grep "launch alone")
    echo "$1->$2 k$3 @$4 order=$o: $r"
  done
done
