# Same-box A/B of the ER-NeRF frame under environment switches (GPU box): tools/ab_nerf_env.sh "A=1" "B=2 C=3" ...  (each argument one arm; "" = defaults)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --workload ernerf --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --steps 300 --warmup 30"
for rep in 1 2 3; do
  for arm in "$@"; do
    env $arm timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[%s]' % '$arm', d['value'], 'frames/s', d['ms_per_step'], 'ms')" | tee -a gpurun_out/ab_nerf_env.txt
  done
done
