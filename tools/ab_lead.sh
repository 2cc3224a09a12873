#!/bin/bash
# same-box A/B of library builds: producer-wave loop stamps on two UNet shapes, then the MuseTalk and Wav2Lip steps    usage: tools/ab_lead.sh libA.so libB.so ...
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; OUT=gpurun_out/ab_lead.txt; : > $OUT
cp mere-fusion_amd/libmerefusion_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for lib in "$@"; do
  cp $lib mere-fusion_amd/libmerefusion_hip.so
  echo "== $lib (rep $rep)" >> $OUT
  if [ $rep = 1 ]; then
    for cfg in "--cin 1280 --cout 1280 --k 3 --hw 8|128x128|6" "--cin 640 --cout 640 --k 3 --hw 16|128x128|3" "--cin 1280 --cout 1280 --k 3 --hw 4|64x64|6"; do
      IFS='|' read shape tile split <<< "$cfg"
      MF_DEBUG=times MF_FORCE_TILE=$tile MF_FORCE_SPLIT=$split MF_FORCE_LD=3 timeout 200 python tools/conv_probe.py $shape --batch 8 --iters 20 2>&1 | grep -E "MF_DEBUG=times|alone" | tail -2 | cut -c1-200 >> $OUT
    done
  fi
  timeout 300 python bench.py --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 --steps 60 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('musetalk', d['value'], d['ms_per_step'])" >> $OUT
  timeout 300 python bench.py --workload wav2lip --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wav2lip', d['value'], d['ms_per_step'])" >> $OUT
done
done
cp /tmp/lib_orig.so mere-fusion_amd/libmerefusion_hip.so
cat $OUT
