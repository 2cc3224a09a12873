#!/bin/bash
# The operating point under the profiler (GPU box): the 8 sessions x 8 frames step (bench.py --batch 64 = what MuseBatcher issues for 8 sessions) --
# rocprofv3 kernel stats by (kernel, grid), the MFMA-utilisation counters and the HBM counters, each in its own pass (--kernel-trace the only trace domain
# beside --pmc).  usage: tools/b64_profile.sh [tag] [batch]
TAG=${1:-r03}; B=${2:-64}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--workload musetalk --batch $B --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0"
rm -rf /tmp/prof_b$B
rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o mt -- python $R/bench.py $ARGS --steps 10 --warmup 3 > /tmp/prof_b$B.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_b$B -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_stats_musetalk_b$B.md 2>&1
grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' /tmp/prof_b$B.log >> $R/gpurun_out/${TAG}_kernel_stats_musetalk_b$B.md
if [ "${PMC:-1}" = 1 ]; then
  rm -rf /tmp/pmcm_b$B
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d /tmp/pmcm_b$B -o p -- python $R/bench.py $ARGS --steps 3 --warmup 1 > /tmp/pmcm_b$B.log 2>&1 || tail -5 /tmp/pmcm_b$B.log
  python $R/tools/pmc_mfma_summary.py /tmp/pmcm_b$B > $R/gpurun_out/${TAG}_pmc_mfma_util_musetalk_b$B.md 2>&1
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmch_b${B}_$CTR
    timeout 900 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmch_b${B}_$CTR -o p -- python $R/bench.py $ARGS --steps 3 --warmup 1 > /tmp/pmch_b${B}_$CTR.log 2>&1 || tail -5 /tmp/pmch_b${B}_$CTR.log
  done
  python $R/tools/pmc_hbm_summary.py /tmp/pmch_b${B}_FETCH_SIZE /tmp/pmch_b${B}_WRITE_SIZE > $R/gpurun_out/${TAG}_pmc_hbm_musetalk_b$B.md 2>&1
fi
head -20 $R/gpurun_out/${TAG}_kernel_stats_musetalk_b$B.md | cut -c1-200
