# GPU box: the packed-FMA root-cause study.  bash tools/pkfma_study.sh [calls]   (after tools/pkfma_variants.sh + the reproducer were built here)
cd $GRAFT_REPO_ROOT
N=${1:-40}
O=gpurun_out/pkfma_study.txt
: > $O
echo "== standalone reproducer (tools/pkfma_repro.hip)" | tee -a $O
timeout 300 tools/bin/pkfma_repro 200 2>&1 | tee -a $O
for v in "" packed packed_nop packed_f32 dbg_fast dbg_fast_nops dbg_fast_scalar; do
  [ -n "$v" ] && [ ! -f build_ab/lib$v.so ] && continue
  echo "== library: ${v:-shipped}" | tee -a $O
  env ${v:+MF_LIB_PATH=build_ab/lib$v.so} timeout 600 python tools/unet_copies_probe.py 8 $N 2>&1 | grep -v Warning | tail -12 | tee -a $O
done
echo "== MF_DEBUG=copies, packed build, eager" | tee -a $O
MF_LIB_PATH=build_ab/libpacked.so MF_DEBUG=copies MF_NO_GRAPH=1 timeout 900 python tools/unet_copies_probe.py 8 6 2>&1 | grep "copies\|call" | head -60 | tee -a $O
