#!/bin/bash
# A/B build of the library with extra compiler flags: tools/build_variant.sh NAME -DFOO=1 ...  -> build_ab/libbbbp_NAME.so (run with BBBP_LIB=...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/bbbp-multi-modal-deep-ensemble-framework_amd/csrc
out=$root/build_ab/$name; mkdir -p $out
objs=""
for f in util gemm conv conv_wino conv_b3 conv_b3c1 rowops engine encoder attention attention_b3 preprocess mlp head forest; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-function "$@" -I $root/include -c $src/$f.hip -o $out/$f.o &
  objs="$objs $out/$f.o"
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_ab/libbbbp_$name.so $objs
echo $root/build_ab/libbbbp_$name.so
