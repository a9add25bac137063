#!/bin/bash
# Diagnostic: another build of ONE source file with extra -D flags, linked with the objects of the current build into
# tools/variants/libsaragan_hip_<name>.so (load it with SARAGAN_LIB=...).  usage: tools/build_variant.sh <name> <file.hip> [-DX=1 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/variants /tmp/sgvar_$name
obj=/tmp/sgvar_$name/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c saragan_amd/csrc/$src -o $obj 2>/dev/null
objs=""
for o in saragan_amd/build/*.o; do
  case "$o" in *-gfx950.o) continue;; esac
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libsaragan_hip_$name.so $objs
echo tools/variants/libsaragan_hip_$name.so
