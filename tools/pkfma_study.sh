# GPU box: the packed-FMA root-cause study.  bash tools/pkfma_study.sh <calls> <variant> ...   ("shipped" = the product library;
# other names = build_ab/lib<name>.so from tools/pkfma_variants.sh / tools/pkfma_variants2.py, built in the container)
cd $GRAFT_REPO_ROOT
N=${1:-40}; shift
O=gpurun_out/pkfma_study.txt
export MF_TUNE_CACHE=$GRAFT_REPO_ROOT/mere-fusion_amd/tune/gfx950.txt        # every build launches the shipped configurations
for v in "$@"; do
  if [ "$v" = repro ]; then
    echo "== standalone reproducer (tools/pkfma_repro.hip)" | tee -a $O
    timeout 300 tools/bin/pkfma_repro 200 2>&1 | tee -a $O
    continue
  fi
  [ "$v" != shipped ] && [ ! -f build_ab/lib$v.so ] && continue
  echo "== library: $v" | tee -a $O
  if [ "$v" = shipped ]; then timeout 600 python tools/unet_copies_probe.py 8 $N 2>&1 | grep "^\[\|^call [01]:" | tee -a $O
  else MF_LIB_PATH=build_ab/lib$v.so timeout 600 python tools/unet_copies_probe.py 8 $N 2>&1 | grep "^\[build\|^call [01]:" | tee -a $O; fi
done
