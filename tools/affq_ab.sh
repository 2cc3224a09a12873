# Same-box A/B of the conversion kernel variants (MF_AFFQ_VARIANT, mf_aux.hip) INSIDE the MuseTalk step: rocprofv3 kernel stats of the 64-frame and
# batch-8 steps per variant -- does the time the conversion pass gives back stay gained, or do the power-capped convolutions around it take it?
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in 0 2; do for b in 64 8; do
  rm -rf /tmp/p_$v_$b
  MF_AFFQ_VARIANT=$v rocprofv3 --kernel-trace --stats -d /tmp/p_${v}_$b -o mt -- python $R/bench.py --workload musetalk --batch $b --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps 10 --warmup 3 > /tmp/p_${v}_$b.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/p_${v}_$b -name "*.db" | head -1) > $R/gpurun_out/affq_v${v}_b$b.md 2>&1
  grep -o '"ms_per_step": [0-9.]*' /tmp/p_${v}_$b.log >> $R/gpurun_out/affq_v${v}_b$b.md
done; done
cd $R; timeout 900 python -m pytest tests/test_musetalk.py tests/test_musetalk_stress.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/affq_tests.txt
for f in gpurun_out/affq_v*_b*.md; do echo "== $f"; sed -n 5,12p $f | cut -c1-150; tail -2 $f; done
