# Same-box A/B of the specialised-workgroup f16 + FP6 halo kernel (MF_HALO_Q_SP=1) against the 8-compute-wave one: the three VAE 3x3 grids without and with
# a residual, and the two upsample + 3x3 layers (four phase launches each); every line carries the check against an fp64 convolution.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/q_sp_ab.txt
: > $OUT
for v in "MF_X=0" "MF_HALO_Q_SP=1" "MF_X=0" "MF_HALO_Q_SP=1"; do
  echo "== variant: [$v]" | tee -a $OUT
  for s in "128,128,256,0,0" "256,256,128,0,0" "512,512,64,0,0" "128,128,256,1,0" "512,512,64,1,0" "512,512,64,0,1" "256,256,128,0,1"; do IFS=, read c1 c2 hw res up <<< "$s"
    echo -n "$s: " | tee -a $OUT
    env $v python tools/conv_probe.py --cin $c1 --cout $c2 --hw $hw --batch 8 --residual $res --upsample $up --precision f16q --iters 30 --check 1 2>&1 | grep -E "check|alone|rror" | tr '\n' ' ' | cut -c1-220 | tee -a $OUT; echo | tee -a $OUT
  done
done
