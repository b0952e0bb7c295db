#!/bin/bash
# usage (GPU box): bash tools/pmc_attn2.sh <tag> <lib|cur> "<counters>" "<attn_bench args>"  -- one SQ counter pass over tools/attn_bench.py
tag=$1; lib=$2; ctr=$3; args=$4
R=$GRAFT_REPO_ROOT
if [ "$lib" = "cur" ]; then export MGX_LIB=$R/musicgeneration_amd/libmgx.so; else export MGX_LIB=$R/tools/ab/libmgx_$lib.so; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/attn_bench.py $args > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmcsum.py $R/gpurun_out/pmc_$tag
