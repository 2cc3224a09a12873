#!/bin/bash
# HBM traffic of the timed steps themselves (GPU box): two rocprofv3 --pmc passes per workload (FETCH_SIZE, WRITE_SIZE: they do not fit one)
# over `bench.py --extras 0 --profile-iters 0`, --kernel-trace the only trace domain beside them.  usage: tools/pmc_hbm_step.sh [tag]
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export MF_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_cache_pmc.txt
[ -f $R/profiles/${TAG}_tune_cache.txt ] && cp $R/profiles/${TAG}_tune_cache.txt $MF_TUNE_CACHE
cd /tmp && export TMPDIR=/tmp
for WL in musetalk wav2lip ernerf; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmch_${WL}_$CTR
    timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmch_${WL}_$CTR -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 \
        --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 > /tmp/pmch_${WL}_$CTR.log 2>&1 || { echo "pass $WL $CTR failed"; tail -5 /tmp/pmch_${WL}_$CTR.log; }
  done
  python $R/tools/pmc_hbm_summary.py /tmp/pmch_${WL}_FETCH_SIZE /tmp/pmch_${WL}_WRITE_SIZE > $R/gpurun_out/${TAG}_pmc_hbm_$WL.md 2>&1
  head -16 $R/gpurun_out/${TAG}_pmc_hbm_$WL.md | cut -c1-230; tail -n 1 $R/gpurun_out/${TAG}_pmc_hbm_$WL.md
done
