# An ablation build of the library: bash tools/ab_build.sh <out name> <source.hip> "<extra hipcc flags>"  ->  build_ab/lib<out name>.so
# (the other objects come from build/obj, i.e. from the last regular build; use with tools/ab_lib.sh or MF_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
name=$1; src=$2; flags=$3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=262144 $flags -c mere-fusion_amd/csrc/$src -o build_ab/$name.o
objs=$(ls build/obj/*.o | grep -v "/${REPLACES:-$src}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ab/lib$name.so $objs build_ab/$name.o
echo build_ab/lib$name.so
