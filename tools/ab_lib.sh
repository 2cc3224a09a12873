# Same-box A/B of two builds of the library (GPU box): tools/ab_lib.sh build_ab/libA.so build_ab/libB.so
# per build: the f16 + FP6 conv kernel alone on its three VAE shapes (tools/q_probe.sh), then the MuseTalk step twice
cd $GRAFT_REPO_ROOT
B="python bench.py --workload musetalk --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps 60 --warmup 8"
for rep in 1 2; do
  for lib in "$@"; do
    cp $lib mere-fusion_amd/libmerefusion_hip.so
    echo "== $lib (rep $rep)" | tee -a gpurun_out/ab_lib.txt
    [ $rep = 1 ] && EXTRA=() bash tools/q_probe.sh 2>&1 | tee -a gpurun_out/ab_lib.txt
    $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('musetalk', d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab_lib.txt
  done
done
