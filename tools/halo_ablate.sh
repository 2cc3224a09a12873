#!/bin/bash
# (needs profiles/r06_halo_ablate_macros.patch applied first: the switches no longer live in the kernels)
# Where the (bf16x3) halo conv's time goes: rebuilds the library with MF_HALO_ABLATE bits (wrong results, timing only) into build_ab/ and times
# one layer with each.  Run the BUILD part here (no GPU): tools/halo_ablate.sh build ; the TIMING part on the GPU box: tools/halo_ablate.sh run
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
SRC="mf_api.cpp mf_conv.hip mf_conv_halo.hip mf_conv_halo2.hip mf_aux.hip mf_wav2lip.hip mf_conv_api.hip mf_mel.hip mf_nn.hip mf_attn.hip mf_whisper.hip mf_musetalk.hip mf_nerf.hip mf_nerf_net.hip mf_nerf_fused.hip mf_nerf_audio.hip"
if [ "$1" = build ]; then
  mkdir -p build_ab
  for m in 1 2 3 4 8 15; do
    ( cd mere-fusion_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DMF_HALO_ABLATE=$m -o $R/build_ab/libmf_ab$m.so $SRC ) &
  done
  wait; ls -la build_ab
else
  for shape in "128 128 256 8" "256 256 128 8" "64 64 96 16"; do
    set -- $shape
    for m in 0 1 2 3 4 8 15; do
      if [ $m = 0 ]; then unset MF_LIB_PATH; else export MF_LIB_PATH=$R/build_ab/libmf_ab$m.so; fi
      echo -n "$1->$2 @$3 B=$4 ablate=$m: "
      timeout 120 python tools/conv_probe.py --cin $1 --cout $2 --k 3 --pad 1 --hw $3 --batch $4 --residual 0 --iters 20 2>&1 | grep -E "conv launch alone" | tr "\n" " "; echo
    done
  done
fi
