#!/bin/bash
# usage: bash tools/build_variant.sh <name> <file.hip> <-D flags...>   -> variants/libmgx_<name>.so (other objects from the normal build)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p variants
C=musicgeneration_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $C/$src -o variants/${name}.o
objs=""
for o in $C/build/*.o; do
  if [ "$(basename $o)" != "${src%.hip}.o" ]; then objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmgx_${name}.so variants/${name}.o $objs
echo variants/libmgx_${name}.so
