# Same-box A/B of the MuseTalk step under environment switches, three repeats (GPU box): tools/ab_env_mt.sh "A=1" "" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --workload musetalk --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps 80 --warmup 10"
for rep in 1 2 3; do
  for arm in "$@"; do
    env $arm timeout 400 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[%s]' % '$arm', d['value'], 'frames/s', d['ms_per_step'], 'ms')" | tee -a gpurun_out/ab_env_mt.txt
  done
done
