#!/bin/bash
# Round-end evidence run (GPU box): full test tier, the default bench line, rocprofv3 kernel stats of the same command.
# Outputs under gpurun_out/ (copied into profiles/ by hand afterwards).
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/gpu_tests.txt
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_mt
rocprofv3 --kernel-trace --stats -d /tmp/prof_mt -o mt -- python $R/bench.py --steps 10 --warmup 3 --extras 0 --cpu-seconds 0 > /tmp/prof_mt.log 2>&1
DB=$(find /tmp/prof_mt -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB > $R/gpurun_out/kernel_stats_musetalk.md 2>&1
rm -rf /tmp/prof_w2l
rocprofv3 --kernel-trace --stats -d /tmp/prof_w2l -o w2l -- python $R/bench.py --workload wav2lip --steps 30 --warmup 5 --extras 0 --cpu-seconds 0 > /tmp/prof_w2l.log 2>&1
DB=$(find /tmp/prof_w2l -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB > $R/gpurun_out/kernel_stats_wav2lip.md 2>&1
rm -rf /tmp/prof_nf
rocprofv3 --kernel-trace --stats -d /tmp/prof_nf -o nf -- python $R/bench.py --workload ernerf --steps 30 --warmup 3 --extras 0 --cpu-seconds 0 > /tmp/prof_nf.log 2>&1
DB=$(find /tmp/prof_nf -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB > $R/gpurun_out/kernel_stats_ernerf.md 2>&1
cd $R
cat gpurun_out/gpu_tests.txt; cut -c1-1500 gpurun_out/bench_line.json; head -20 gpurun_out/kernel_stats_musetalk.md
