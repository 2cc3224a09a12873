#!/bin/bash
# Same-box A/B of library builds on the ER-NeRF frame: bash tools/ab_nerf.sh build_ab/libA.so build_ab/libB.so ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/ab_nerf.txt
cp mere-fusion_amd/libmerefusion_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do
  for lib in "$@"; do
    cp $lib mere-fusion_amd/libmerefusion_hip.so
    timeout 200 python bench.py --workload ernerf --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib rep $rep:', d['value'], 'frames/s', d['ms_per_step'], 'ms')" | tee -a gpurun_out/ab_nerf.txt
  done
done
cp /tmp/lib_orig.so mere-fusion_amd/libmerefusion_hip.so
