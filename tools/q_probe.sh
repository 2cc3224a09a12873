# f16 + FP6 conv kernel on its three VAE shapes (GPU box): tools/q_probe.sh [env assignments...]  -> per-phase s_memtime medians + launch time
cd $GRAFT_REPO_ROOT
for s in "128 128 256" "256 256 128" "512 512 64"; do set -- $s
  env "${EXTRA[@]}" MF_DEBUG=times python tools/conv_probe.py --cin $1 --cout $2 --hw $3 --batch 8 --residual 0 --precision f16q --iters 20 --check 1 2>&1 | grep -E "DBG|alone|err|diff" | tail -3 | cut -c1-240
done
