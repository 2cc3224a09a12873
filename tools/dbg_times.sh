#!/bin/bash
# per-workgroup s_memtime stamps (MF_DEBUG=times) of k_conv_igemm on a few UNet shapes, for the forced configurations given as "tile split ld" triples
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/${1:-dbg}_times.txt; : > $OUT
run() { # shape-args, tile, split, ld
  echo "== $1 | tile $2 split $3 ld $4" >> $OUT
  MF_DEBUG=times MF_FORCE_TILE=$2 MF_FORCE_SPLIT=$3 MF_FORCE_LD=$4 timeout 300 python tools/conv_probe.py $1 --batch 8 --iters 20 2>&1 | grep -E "MF_DEBUG=times|alone" >> $OUT
}
S1="--cin 1280 --cout 1280 --k 3 --hw 4"
S2="--cin 1280 --cout 1280 --k 3 --hw 8"
S3="--cin 1280 --cout 1280 --k 1 --pad 0 --hw 8 --residual 0"
S4="--cin 320 --cout 320 --k 1 --pad 0 --hw 32 --residual 0"
S5="--cin 320 --cout 320 --k 3 --hw 32"
run "$S1" 128x64 12 2; run "$S1" 128x64 12 3; run "$S1" 128x64 6 3
run "$S2" 128x128 6 0; run "$S2" 128x128 6 3; run "$S2" 128x64 3 3; run "$S2" 128x64 6 3
run "$S3" 64x64 1 2; run "$S3" 64x64 1 3; run "$S3" 64x64 2 4; run "$S3" 128x64 2 3
run "$S4" 64x64 1 2; run "$S4" 64x64 1 3; run "$S4" 64x64 1 4; run "$S4" 128x64 1 3
run "$S5" 128x64 3 2; run "$S5" 128x64 3 3; run "$S5" 128x64 1 3; run "$S5" 128x128 1 3; run "$S5" 64x64 1 3
cat $OUT
