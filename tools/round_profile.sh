#!/bin/bash
# Round-end evidence run (GPU box): full test tier, the default bench line, rocprofv3 kernel stats of the SAME timed steps
# (--extras 0 --profile-iters 0: nothing but warm-up + timed steps runs under the profiler, so per-launch averages are the bench's own).
# Outputs under gpurun_out/ (copied into profiles/ by hand afterwards).   usage: tools/round_profile.sh [tag]
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
# one tuning cache for the bench run and the profiled re-runs: the profiles then hold the production launch configurations and no tuning launches
export MF_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_cache.txt
rm -f $MF_TUNE_CACHE
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.txt
python bench.py --dump-layers gpurun_out/${TAG}_layers.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_err.txt
cd /tmp && export TMPDIR=/tmp
for WL in musetalk wav2lip ernerf; do
  rm -rf /tmp/prof_$WL
  STEPS=20; [ $WL = wav2lip ] && STEPS=100; [ $WL = ernerf ] && STEPS=50
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$WL -o $WL -- python $R/bench.py --workload $WL --steps $STEPS --warmup 5 --extras 0 --cpu-seconds 0 --profile-iters 0 > /tmp/prof_$WL.log 2>&1
  DB=$(find /tmp/prof_$WL -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $DB > $R/gpurun_out/${TAG}_kernel_stats_$WL.md 2>&1
done
cd $R
cat gpurun_out/${TAG}_gpu_tests.txt; cut -c1-1500 gpurun_out/${TAG}_bench_line.json; head -30 gpurun_out/${TAG}_kernel_stats_musetalk.md
