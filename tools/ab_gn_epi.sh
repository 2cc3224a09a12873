# A/B of the GroupNorm-statistics-from-the-producer paths (GPU box): MF_GN_EPI=2 = only the f16 + FP6 producers (the previous state)
cd $GRAFT_REPO_ROOT
B="python bench.py --workload musetalk --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps 60 --warmup 8"
run() { $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'])" >> gpurun_out/ab_bench.txt; }
for i in 1 2; do
  MF_GN_EPI=2 run old
  run new1024
  MF_COMBINE_BLOCKS=2048 run new2048
  MF_COMBINE_BLOCKS=4096 run new4096
done
cat gpurun_out/ab_bench.txt
