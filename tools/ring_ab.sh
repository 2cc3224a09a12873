#!/bin/bash
# deep LDS ring x tile x split on UNet shapes (GPU box)
for shape in "640 640 3 16" "1280 1280 3 8" "320 320 3 32" "1920 640 3 16" "2560 1280 3 8" "1280 1280 3 4"; do
  set -- $shape
  for cfg in "64x64 1 2" "128x128 2 deep" "128x128 3 deep" "128x128 4 deep" "128x128 6 deep" "128x64 2 deep" "128x64 3 deep" "128x64 4 deep" "64x64 1 deep" "64x64 2 deep" "64x64 3 deep"; do
    set -- $shape; c=($cfg)
    r=$(MF_RING=${c[2]} MF_FORCE_TILE=${c[0]} MF_FORCE_SPLIT=${c[1]} MF_DBG_TIMES=1 python tools/conv_probe.py --cin $1 --cout $2 --k $3 --pad $(($3/2)) --hw $4 --batch 8 --residual 0 --iters 30 2>&1 | grep "launch alone\|DBG_TIMES" | tail -2 | sed 's/.*alone: //; s/.*WGs tile/WGs tile/; s/; WG start.*//' | tr '\n' ' ')
    echo "$1->$2 k$3 @$4 tile=${c[0]} split=${c[1]} ring=${c[2]}: $r"
  done
done
