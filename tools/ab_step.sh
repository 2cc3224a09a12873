# Same-box A/B of library builds on the MuseTalk step (batch 8 and the 8 x 8 operating point): bash tools/ab_step.sh build_ab/libA.so build_ab/libB.so ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp mere-fusion_amd/libmerefusion_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do
  for lib in "$@"; do
    cp $lib mere-fusion_amd/libmerefusion_hip.so
    for b in ${BATCHES:-8 64}; do
      python bench.py --workload musetalk --batch $b --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps $([ $b = 8 ] && echo 60 || echo 12) --warmup $([ $b = 8 ] && echo 8 || echo 3) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib rep $rep batch $b:', d['value'], 'frames/s', d['ms_per_step'], 'ms')" | tee -a gpurun_out/ab_step.txt
    done
  done
done
cp /tmp/lib_orig.so mere-fusion_amd/libmerefusion_hip.so
