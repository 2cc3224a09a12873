#!/bin/bash
# rocprofv3 kernel stats of the ER-NeRF bench leg's timed frames   usage: tools/nerf_prof.sh [tag]
TAG=${1:-nerf}; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_nerf
rocprofv3 --kernel-trace --stats -d /tmp/prof_nerf -o ernerf -- python $R/bench.py --workload ernerf --steps 50 --warmup 5 --extras 0 --cpu-seconds 0 --profile-iters 0 > /tmp/prof_nerf.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_nerf -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_stats_ernerf.md 2>&1
head -16 $R/gpurun_out/${TAG}_kernel_stats_ernerf.md | cut -c1-170; grep -E "k_nerf_field_fused|k_loop_march|k_loop_composite" $R/gpurun_out/${TAG}_kernel_stats_ernerf.md | sed -n 4,40p | cut -c1-200
