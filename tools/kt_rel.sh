#!/bin/bash
# usage (GPU box): bash tools/kt_rel.sh <tag> <lib|cur> "<rel_bench args>"  -- kernel-trace pass over tools/rel_bench.py
tag=$1; lib=$2; args=$3
R=$GRAFT_REPO_ROOT
if [ "$lib" = "cur" ]; then export MGX_LIB=$R/musicgeneration_amd/libmgx.so; else export MGX_LIB=$R/tools/ab/libmgx_$lib.so; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/tools/rel_bench.py $args > $R/gpurun_out/kt_$tag.log 2>&1 || { tail -5 $R/gpurun_out/kt_$tag.log; exit 1; }
python3 $R/tools/profsum.py $R/gpurun_out/kt_$tag 8 | cut -c1-70,100-175
