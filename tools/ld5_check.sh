#!/bin/bash
# The direct operand path (k_conv_igemm LD 5: no LDS stage) on the UNet's short-K shapes at batch 8, beside ld 2 (registers -> one LDS stage) and ld 3 (producer waves):
# torch-fp64 check and launch time per (shape, tile, split, path).   bash tools/ld5_check.sh   (GPU box)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/ld5_check.txt; : > $OUT
run() { # shape-args, tile, split, ld
  printf "%-58s tile %-7s split %-2s ld %s : " "$1" $2 $3 $4 >> $OUT
  MF_FORCE_TILE=$2 MF_FORCE_SPLIT=$3 MF_FORCE_LD=$4 timeout 300 python tools/conv_probe.py $1 --batch 8 --iters 30 --check 1 2>&1 | grep -E "check|alone" | tr '\n' ' ' | sed 's/ \+/ /g' >> $OUT
  echo >> $OUT
}
S1="--cin 320 --cout 320 --k 1 --pad 0 --hw 32 --residual 0"
S2="--cin 320 --cout 960 --k 1 --pad 0 --hw 32 --residual 0"
S3="--cin 640 --cout 640 --k 1 --pad 0 --hw 16 --residual 0"
S4="--cin 1280 --cout 1280 --k 1 --pad 0 --hw 8 --residual 0"
S5="--cin 1280 --cout 320 --k 1 --pad 0 --hw 32 --residual 1"
S6="--cin 320 --cout 320 --k 3 --hw 32"
S7="--cin 640 --cout 640 --k 3 --hw 16"
for S in "$S1" "$S2" "$S3" "$S4" "$S5" "$S6" "$S7"; do
  for cfg in "64x64 1 2" "64x64 1 3" "64x64 1 5" "128x64 1 2" "128x64 1 3" "128x64 1 5"; do run "$S" $cfg; done
done
run "$S4" 64x64 2 5; run "$S4" 64x64 2 2; run "$S7" 64x64 3 5; run "$S7" 64x64 3 2; run "$S7" 128x64 3 3
cat $OUT
