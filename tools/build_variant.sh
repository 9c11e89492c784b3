#!/bin/bash
# Build an experimental variant of the library:  tools/build_variant.sh NAME "-DFM_PF_SUM=3 ..."
# -> build/variants/libfmatch_NAME.so   (run it with: python tools/bench_variant.py build/variants/libfmatch_NAME.so ...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants; mkdir -p $OUT/obj_$NAME; rm -f $OUT/obj_$NAME/*.o
for f in api coarse_prep coarse_dense coarse_max_i8 coarse_screen coarse_select fine fine_tf coarse_tf post dsm_grad; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$ROOT/featurematching_amd/csrc "$@" \
    -c $ROOT/featurematching_amd/csrc/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libfmatch_$NAME.so $OUT/obj_$NAME/*.o -Wl,-rpath,/opt/rocm/lib
echo $OUT/libfmatch_$NAME.so
