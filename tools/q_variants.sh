# f16 + FP6 conv kernel variants on its three VAE shapes, same box (GPU box): bash tools/q_variants.sh "<env assignments>" "<env assignments>" ...
# per variant and shape: check vs fp64 conv, launch time alone, and (second run, DBG=1) the s_memtime phase medians
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/q_variants.txt
: > $OUT
for v in "$@"; do
  echo "== variant: [$v]" | tee -a $OUT
  for s in ${SHAPES:-"128,128,256" "256,256,128" "512,512,64"}; do IFS=, read c1 c2 hw <<< "$s"
    env $v python tools/conv_probe.py --cin $c1 --cout $c2 --hw $hw --batch 8 --residual ${RES:-0} --precision f16q --iters 30 --check 1 2>&1 | grep -E "check|alone|rror" | cut -c1-200 | tee -a $OUT
    [ "${DBG:-0}" != 0 ] && env $v MF_DEBUG=times python tools/conv_probe.py --cin $c1 --cout $c2 --hw $hw --batch 8 --residual ${RES:-0} --precision f16q --iters 10 2>&1 | grep -E "DBG" | tail -1 | cut -c1-260 | tee -a $OUT
  done
done
