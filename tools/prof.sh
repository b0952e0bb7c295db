#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools_prof.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log
