#!/bin/bash
# GPU box: the render loop's tail launch (k_loop_tail) -- parity tests, then frames/s of the ER-NeRF bench leg with the launch-only chain (MF_NERF_TAIL_AFTER=off, the
# loop as it was before round 6), the adaptive chain (default) and fixed hand-over points.   usage: tools/nerf_tail_ab.sh [tag]
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
OUT=gpurun_out/${TAG}_nerf_tail_ab.txt; : > $OUT
timeout 900 python -m pytest tests/test_ernerf.py -m gpu -q -x -k "tail or launched_rounds or device_loop" 2>&1 | tail -5 | tee -a $OUT
for MODE in off default 0; do
  if [ $MODE = default ]; then unset MF_NERF_TAIL_AFTER; else export MF_NERF_TAIL_AFTER=$MODE; fi
  for REP in 1 2; do
    L=$(timeout 300 python bench.py --workload ernerf --steps 200 --warmup 20 --extras 0 --cpu-seconds 0 --profile-iters 0 2>/dev/null | tail -1)
    echo "MF_NERF_TAIL_AFTER=$MODE rep $REP: $(echo "$L" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms")')" | tee -a $OUT
  done
done
