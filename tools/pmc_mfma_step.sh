#!/bin/bash
# MFMA utilisation of the timed steps themselves (GPU box): one rocprofv3 --pmc pass per workload over `bench.py --extras 0 --profile-iters 0`
# (warm-up + timed steps only), --kernel-trace the only trace domain beside it.  usage: tools/pmc_mfma_step.sh [tag]
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export MF_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_cache_pmc.txt
[ -f $R/profiles/${TAG}_tune_cache.txt ] && cp $R/profiles/${TAG}_tune_cache.txt $MF_TUNE_CACHE      # the production launch configurations, no tuning launches
cd /tmp && export TMPDIR=/tmp
for WL in musetalk wav2lip ernerf; do
  rm -rf /tmp/pmcm_$WL
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d /tmp/pmcm_$WL -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --extras 0 --cpu-seconds 0 \
      --profile-iters 0 --pmc-traffic 0 > /tmp/pmcm_$WL.log 2>&1 || { echo "pass $WL failed"; tail -5 /tmp/pmcm_$WL.log; }
  python $R/tools/pmc_mfma_summary.py /tmp/pmcm_$WL > $R/gpurun_out/${TAG}_pmc_mfma_util_$WL.md 2>&1
  head -24 $R/gpurun_out/${TAG}_pmc_mfma_util_$WL.md | cut -c1-260; tail -1 $R/gpurun_out/${TAG}_pmc_mfma_util_$WL.md
  f=$(find /tmp/pmcm_$WL -name "*counter_collection.csv" | head -1); [ -n "$f" ] && head -2 $f > $R/gpurun_out/${TAG}_pmc_csv_head_$WL.txt
done
