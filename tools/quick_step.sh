#!/bin/bash
# headline step times only (no extras): MuseTalk batch 8 and Wav2Lip batch 16, three runs each    usage: tools/quick_step.sh [tag]
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; TAG=${1:-q}
for i in 1 2; do
  timeout 300 python bench.py --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('musetalk', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --workload wav2lip --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wav2lip', d['value'], d['ms_per_step'])"
done
