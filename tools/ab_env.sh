# Same-box A/B of the MuseTalk step under environment switches (GPU box): tools/ab_env.sh "A=1 B=2" "C=3" ...  (each argument one arm; "" = defaults)
cd $GRAFT_REPO_ROOT
B="python bench.py --workload musetalk --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps 60 --warmup 8"
for rep in 1 2; do
  for arm in "$@"; do
    env $arm $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[%s]' % '$arm', d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab_env.txt
  done
done
