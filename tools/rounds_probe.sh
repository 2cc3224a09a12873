cd $GRAFT_REPO_ROOT
for v in "MF_X=0" "MF_HALO_Q_SP=1"; do for b in 4 8 16 32; do
  echo -n "[$v] batch $b: "; env $v MF_DBG_TIMES=1 python tools/conv_probe.py --cin 512 --cout 512 --hw 64 --batch $b --residual 0 --precision f16q --iters 20 2>&1 | grep -E "DBG|alone" | tail -2 | tr '\n' ' ' | sed 's/span [0-9.]*;//; s/WG start.*//' | cut -c1-260; echo
done; done
