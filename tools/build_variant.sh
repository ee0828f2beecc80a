#!/bin/bash
# Build targetdiff_amd/lib/variant_NAME.so with extra compiler flags on edge16.hip (tuning macros), for tools/ab_variants.sh:
#   tools/build_variant.sh A "-DTD_L2_F16=0"; tools/build_variant.sh B ""
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=$2; FILES=${3:-edge16.hip}
python -m targetdiff_amd.build > /dev/null
OBJ=targetdiff_amd/build
mkdir -p $OBJ/variant_$NAME
cp $OBJ/*.o $OBJ/variant_$NAME/
for F in $FILES; do
  FL="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -x hip -Wall -Wno-unused-function"
  [ "$F" = edge16.hip ] && FL="$FL -fno-slp-vectorize"
  [ "$F" = node.hip ] && FL="$FL -Wno-inline-asm"
  /opt/rocm/bin/hipcc $FL $EXTRA -c targetdiff_amd/csrc/$F -o $OBJ/variant_$NAME/${F%.*}.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script,targetdiff_amd/csrc/exports.map -o targetdiff_amd/lib/variant_$NAME.so $OBJ/variant_$NAME/*.o
echo built targetdiff_amd/lib/variant_$NAME.so "($EXTRA)"
