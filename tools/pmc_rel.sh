#!/bin/bash
# usage (GPU box): bash tools/pmc_rel.sh <tag> "<counters>" "<rel_bench args>"  -- one SQ counter pass over tools/rel_bench.py
tag=$1; ctr=$2; args=$3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/rel_bench.py $args > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmcsum.py $R/gpurun_out/pmc_$tag
