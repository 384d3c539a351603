#!/bin/bash
# build_variant.sh NAME "FLAGS" FILE.hip [FILE.hip ...]: a variant of the library for A/B runs -- the named sources recompiled with
# FLAGS into build_dbg/NAME/, linked with the product's other objects into build_dbg/NAME/libmp2gpu.so (select it with MP2G_LIB).
# The product library and its objects are never touched. Works here (cross-compile) and on the GPU box.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
NAME=$1; FLAGS=$2; shift 2
SRC=$R/mapreduce-plonky2_amd/csrc
OUT=$R/build_dbg/$NAME
mkdir -p $OUT
make -s -j8 -C $SRC
OBJS=""
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I$R/include $FLAGS -c $f -o $OUT/$b.o
    OBJS="$OBJS $OUT/$b.o"
  else
    OBJS="$OBJS $SRC/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libmp2gpu.so $OBJS
echo $OUT/libmp2gpu.so
