#!/bin/bash
# usage: bash tools/ab_build.sh NAME SRC.hip [-DFLAG ...]   -> tools/ab/libmgx_NAME.so (one source rebuilt with the flags)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c musicgeneration_amd/csrc/$src.hip -o tools/ab/${src}_$name.o
objs=$(ls musicgeneration_amd/csrc/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libmgx_$name.so $objs tools/ab/${src}_$name.o
rm -f tools/ab/${src}_$name.o
