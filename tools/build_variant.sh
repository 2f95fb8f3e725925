#!/bin/bash
# A/B builds: one translation unit recompiled with extra -D flags and linked with the product's other objects into
# transferia_amd/variants/libtfgpu_<name>.so; TFGPU_LIB_VARIANT=<name> makes bench.py load it (measurement only).
# usage: tools/build_variant.sh NAME "UNIT.hip [UNIT2.hip …]" "-DX=… -DY=…"
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; FLAGS=$3
python -m transferia_amd.build >/dev/null
mkdir -p transferia_amd/variants
OBJS=$(ls transferia_amd/build/*.o)
NEW=""
for U in $UNIT; do
  OBJ=transferia_amd/variants/${U%.*}_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip $FLAGS -Iinclude -Itransferia_amd/csrc -c transferia_amd/csrc/$U -o $OBJ
  OBJS=$(echo "$OBJS" | grep -v "/${U%.*}.o")
  NEW="$NEW $OBJ"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o transferia_amd/variants/libtfgpu_$NAME.so $OBJS $NEW -ldl
echo transferia_amd/variants/libtfgpu_$NAME.so
