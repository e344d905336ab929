#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh NAME FILE.hip [-DFLAG ...]
# -> variants/libtspn_NAME.so (select with TSPN_LIB_PATH); other sources are compiled once into /tmp/tspn_objs.
# TSPN_VARIANT_SRC=path/to/other.hip compiles that file IN PLACE OF csrc/FILE.hip (the experimental forms kept
# under tools/probes/, e.g. TSPN_VARIANT_SRC=tools/probes/tspn_bf16_forms.hip tools/build_variant.sh w tspn_bf16.hip).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/temporal-span-proposal-network-vidvrd_amd/csrc
NAME=$1; VFILE=$2; shift 2
VDIR=${TSPN_VARIANT_DIR:-$ROOT/variants}
OBJ=/tmp/tspn_objs; mkdir -p $OBJ $VDIR
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function -Wno-pass-failed -I$ROOT/include -I$CSRC"
objs=""
for f in $CSRC/*.hip; do
  b=$(basename $f .hip)
  if [ "$b.hip" == "$VFILE" ]; then
    hipcc $FLAGS "$@" -c ${TSPN_VARIANT_SRC:-$f} -o $OBJ/${b}_$NAME.o
    objs="$objs $OBJ/${b}_$NAME.o"
  else
    if [ ! -f $OBJ/$b.o ] || [ $f -nt $OBJ/$b.o ] || [ $CSRC/tspn_common.h -nt $OBJ/$b.o ] || [ $ROOT/include/tspn_mi355x.h -nt $OBJ/$b.o ]; then
      hipcc $FLAGS -c $f -o $OBJ/$b.o
    fi
    objs="$objs $OBJ/$b.o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $VDIR/libtspn_$NAME.so
echo $VDIR/libtspn_$NAME.so
