#!/bin/bash
# usage (GPU box): bash tools/ab_prof.sh "B S H dh iters mode" PATTERN name1 name2 ...  -- rocprofv3 kernel stats of
# tools/attn_bench.py under each tools/ab/libmgx_<name>.so (cur = the product library); prints kernels matching PATTERN
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
args=$1; pat=$2; shift 2
for n in "$@"; do
  if [ "$n" = "cur" ]; then lib=$R/musicgeneration_amd/libmgx.so; else lib=$R/tools/ab/libmgx_$n.so; fi
  export MGX_LIB=$lib
  rm -rf /tmp/abp_$n
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$n -- python3 $R/tools/attn_bench.py $args > /tmp/abp_$n.log 2>&1 || { tail -5 /tmp/abp_$n.log; exit 1; }
  echo "== $n"
  python3 $R/tools/profsum.py /tmp/abp_$n | grep -E "$pat" | cut -c1-60,100-170
done
