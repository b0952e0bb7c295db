#!/bin/bash
# usage (GPU box): bash tools/pmc1.sh <tag> <lib|cur> "<counters>" <script.py> "<script args>"
#   ONE rocprofv3 --pmc pass (<= 8 SQ_ / 4 TCC_ counters, nothing else on the command line) over tools/<script.py> under
#   tools/ab/libmgx_<lib>.so (cur: the in-tree library); per-kernel means by tools/pmcsum.py
tag=$1; lib=$2; ctr=$3; script=$4; args=$5
R=$GRAFT_REPO_ROOT
if [ "$lib" = "cur" ]; then export MGX_LIB=$R/musicgeneration_amd/libmgx.so; else export MGX_LIB=$R/tools/ab/libmgx_$lib.so; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/$script $args > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmcsum.py $R/gpurun_out/pmc_$tag
