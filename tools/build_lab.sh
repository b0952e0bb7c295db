#!/bin/bash
# builds libmgx.so and the stand-alone GEMM lab binary (tools/gemm_lab) for gfx950
set -e
cd "$(dirname "$0")/.."
python -m musicgeneration_amd.build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.cpp -o tools/gemm_lab \
    -Lmusicgeneration_amd -lmgx -Wl,-rpath,'$ORIGIN/../musicgeneration_amd'
