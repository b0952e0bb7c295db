#!/bin/bash
# usage: bash tools_pmc.sh <tag> "<counters>" <python script + args>
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/pmc_$tag.log
