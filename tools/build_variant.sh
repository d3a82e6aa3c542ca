#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=..." -- an A/B build of libspmv_hip.so with one kernel file compiled
# differently: spmv-test_amd/lib/libspmv_hip_NAME.so (tools/explore.py takes it as {"lib": "spmv-test_amd/lib/libspmv_hip_NAME.so"})
set -e
NAME=$1; FILE=$2; DEFS=$3
PKG=$(cd "$(dirname "$0")/../spmv-test_amd" && pwd)
make -s -C "$PKG" lib/libspmv_hip.so
OBJS=""
for f in "$PKG"/build/*.o; do
  case "$(basename $f .o)" in
    $(basename $FILE .hip)) ;;
    *) OBJS="$OBJS $f" ;;
  esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden \
  -I"$PKG/../include" -I"$PKG/csrc" $DEFS -c "$PKG/csrc/$FILE" -o "$PKG/build/variant_$NAME.o.tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$PKG/build/variant_$NAME.o.tmp" \
  -Wl,--version-script="$PKG/csrc/exports.map" -o "$PKG/lib/libspmv_hip_$NAME.so"
rm -f "$PKG/build/variant_$NAME.o.tmp"
echo "built $PKG/lib/libspmv_hip_$NAME.so"
