#!/bin/bash
# usage (GPU box): bash tools/kt.sh <tag> <lib|cur> <script.py> "<script args>" [N]
#   rocprofv3 --kernel-trace --stats of tools/<script.py> under tools/ab/libmgx_<lib>.so (cur: the in-tree library);
#   prints the N (default 8) longest kernels.  The program itself follows "--" (no env / bash -c hop under rocprofv3).
tag=$1; lib=$2; script=$3; args=$4; n=${5:-8}
R=$GRAFT_REPO_ROOT
if [ "$lib" = "cur" ]; then export MGX_LIB=$R/musicgeneration_amd/libmgx.so; else export MGX_LIB=$R/tools/ab/libmgx_$lib.so; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/tools/$script $args > $R/gpurun_out/kt_$tag.log 2>&1 || { tail -5 $R/gpurun_out/kt_$tag.log; exit 1; }
python3 $R/tools/profsum.py $R/gpurun_out/kt_$tag $n | cut -c1-70,100-175
