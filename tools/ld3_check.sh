#!/bin/bash
# loader-wave path (ld 3) of k_conv_igemm: correctness against torch fp64 on the four tiles, then the tuner on the UNet + VAE at batch 8 with every layer's
# model pick beside the measured winner, then the headline step with that table.   usage: tools/ld3_check.sh [tag]
TAG=${1:-ld3}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for cfg in "128x64 12 3" "64x64 12 3" "64x64 12 4" "128x128 6 3" "128x64 1 3"; do
  set -- $cfg
  for shape in "--cin 1280 --cout 1280 --k 3 --hw 8" "--cin 640 --cout 1280 --k 1 --pad 0 --hw 8 --residual 0" "--cin 320 --cout 320 --k 3 --hw 32" "--cin 72 --cout 200 --k 3 --hw 13 --residual 0"; do
    echo "== tile $1 split $2 ld $3: $shape"
    MF_FORCE_TILE=$1 MF_FORCE_SPLIT=$2 MF_FORCE_LD=$3 timeout 300 python tools/conv_probe.py $shape --batch 8 --check 1 --iters 20 2>&1 | tail -3
  done
done
echo "== reference: model pick"
for shape in "--cin 1280 --cout 1280 --k 3 --hw 8" "--cin 1280 --cout 1280 --k 3 --hw 4"; do timeout 300 python tools/conv_probe.py $shape --batch 8 --check 1 --iters 20 2>&1 | tail -3; done
} > gpurun_out/${TAG}_check.txt 2>&1
rm -f gpurun_out/${TAG}_tune.txt
MF_TUNE_CACHE=gpurun_out/${TAG}_tune.txt MF_DEBUG=tune timeout 1200 python tools/tune_one_batch.py 8 > gpurun_out/${TAG}_tune_log.txt 2>&1
MF_TUNE_CACHE=gpurun_out/${TAG}_tune.txt timeout 600 python bench.py --extras 0 --cpu-seconds 0 --pmc-traffic 0 --dump-layers gpurun_out/${TAG}_layers.json > gpurun_out/${TAG}_line.json 2> gpurun_out/${TAG}_err.txt
timeout 600 python bench.py --extras 0 --cpu-seconds 0 --pmc-traffic 0 > gpurun_out/${TAG}_line_shipped.json 2>> gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_check.txt; grep -c "ld [34]" gpurun_out/${TAG}_tune_log.txt; tail -3 gpurun_out/${TAG}_tune_log.txt; cut -c1-400 gpurun_out/${TAG}_line.json; echo; cut -c1-400 gpurun_out/${TAG}_line_shipped.json
