#!/bin/bash
# the f16 + FP6 halo conv on the 512-channel 64 x 64 level re-fetches each patch once per channel tile (FETCH_SIZE 267 MB per launch against
# 67 MB of input).  Same-box A/B of the blockIdx -> (patch, channel tile) order: 0 = shipped (XCD-contiguous runs, channel tile fastest), 1 = patch fastest,
# 2 = plain launch order.  Per order: HBM-side bytes (FETCH_SIZE x 2, separate pass), L2 hit rate, and the launch time with package power beside it.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; OUT=$R/gpurun_out/halo_order_ab.txt; : > $OUT
cp mere-fusion_amd/libmerefusion_hip.so /tmp/lib_orig.so
ARGS="--cin 512 --cout 512 --hw 64 --batch 8 --residual 0 --precision f16q"
for o in 0 1 2; do
  cp build_ab/libhalo_ord$o.so mere-fusion_amd/libmerefusion_hip.so
  echo "== order $o" >> $OUT
  ( cd /tmp && export TMPDIR=/tmp
    for pass in "mem:FETCH_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
      name=${pass%%:*}; ctrs=${pass#*:}; rm -rf /tmp/pmc_$name
      timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$name -o out -- python $R/tools/conv_probe.py $ARGS --iters 5 > /tmp/pmc_$name.log 2>&1
      python $R/tools/rocprof_pmc_summary.py /tmp/pmc_$name 2>&1 | grep -i -A4 "halo_w" | head -8 | cut -c1-200 >> $OUT
    done )
  for rep in 1; do timeout 120 python tools/conv_probe.py $ARGS --iters 50 --alone-iters 500 2>&1 | grep alone >> $OUT; done
done
cp /tmp/lib_orig.so mere-fusion_amd/libmerefusion_hip.so
cat $OUT
