for cfg in "640 640 3 16" "320 320 3 32" "1280 1280 3 8" "320 320 1 32" "640 640 1 16" "1280 1280 1 8" "320 2560 1 32" "1280 320 1 32" "960 320 3 32" "1920 640 3 16" "2560 1280 3 8"; do set -- $cfg
 for env in "" "MF_RING=deep" "MF_FORCE_SPLIT=2" "MF_FORCE_SPLIT=4" "MF_FORCE_SPLIT=6" "MF_RING=deep MF_FORCE_SPLIT=2" "MF_RING=deep MF_FORCE_SPLIT=4" "MF_FORCE_SPLIT=8"; do
  r=$(env $env python tools/conv_probe.py --cin $1 --cout $2 --k $3 --pad $(( $3 / 2 )) --hw $4 --batch 8 --residual 0 --iters 20 2>&1 | grep "launch alone" | sed 's/.*alone: //')
  echo "$1->$2 k$3 @$4 [$env]: $r"
 done
done
