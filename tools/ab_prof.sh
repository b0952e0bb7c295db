#!/bin/bash
# usage (GPU box): bash tools/ab_prof.sh <kernel-name-substring> <script> <variant.so>...   per-variant rocprof kernel averages
pat=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  tag=$(basename $so .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/ab_$tag
  MGX_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/$script > $GRAFT_REPO_ROOT/gpurun_out/ab_$tag.log 2>&1 || { echo "$tag failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/ab_$tag.log; exit 1; }
  echo "== $tag"; python3 $GRAFT_REPO_ROOT/tools/profsum.py $out 12 | grep -E "$pat" | cut -c1-60,100-170
done
