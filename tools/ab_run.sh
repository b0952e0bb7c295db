#!/bin/bash
# usage (GPU box): bash tools/ab_run.sh "B S H dh iters mode" name1 name2 ...   -- tools/attn_bench.py under each tools/ab/libmgx_<name>.so
R=$GRAFT_REPO_ROOT
args=$1; shift
for n in "$@"; do
  if [ "$n" = "cur" ]; then lib=$R/musicgeneration_amd/libmgx.so; else lib=$R/tools/ab/libmgx_$n.so; fi
  echo "== $n"
  MGX_LIB=$lib timeout -k 10 120 python3 $R/tools/attn_bench.py $args || exit 1
done
