# implicit-GEMM layers of the UNet in the f16 + FP6 format against bf16x3 (GPU box): launch time alone + error vs an fp64 convolution
# (1x1 shapes = the transformer blocks' linears at 32^2 / 16^2 / 8^2 tokens per frame: fused q|k|v, to_q, GEGLU proj, ff.net.2, proj_in)
cd $GRAFT_REPO_ROOT
for B in ${BATCHES:-64 8}; do
for s in ${SHAPES:-"320 320 3 1 32" "640 640 3 1 16" "1280 1280 3 1 8" "320 320 1 0 32" "1280 1280 1 0 16" "640 320 3 1 32" "320 2560 1 0 32" "1280 640 3 1 16" "320 960 1 0 32" "1280 320 1 0 32" "640 5120 1 0 16" "640 1920 1 0 16" "2560 640 1 0 16" "1280 10240 1 0 8" "1280 3840 1 0 8"}; do set -- $s
  for prec in bf16x3 f16q; do
    echo -n "B=$B $1->$2 k$3 @$5 $prec: "
    python tools/conv_probe.py --cin $1 --cout $2 --k $3 --pad $4 --hw $5 --batch $B --residual 0 --precision $prec --iters 20 --check 1 2>&1 | grep -E "check|alone|rror" | tr '\n' ' ' | cut -c1-230; echo
  done
done; done
