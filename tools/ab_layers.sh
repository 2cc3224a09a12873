#!/bin/bash
# per-op tables (hipEvents, graph off) of the MuseTalk step for two library builds on one box   usage: tools/ab_layers.sh libA.so libB.so
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cp mere-fusion_amd/libmerefusion_hip.so /tmp/lib_orig.so
for lib in "$@"; do
  n=$(basename $lib .so)
  cp $lib mere-fusion_amd/libmerefusion_hip.so
  timeout 300 python bench.py --extras 0 --cpu-seconds 0 --pmc-traffic 0 --dump-layers gpurun_out/layers_$n.json > gpurun_out/line_$n.json 2>/dev/null
done
cp /tmp/lib_orig.so mere-fusion_amd/libmerefusion_hip.so
