#!/bin/bash
# usage (GPU box): bash tools/ab_kt.sh "<attn_bench args>" <kernel substring> name1 name2 ...  -- rocprofv3 kernel-trace of tools/attn_bench.py under each tools/ab/libmgx_<name>.so (or cur), one line per matching kernel
args=$1; pat=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = "cur" ]; then lib=$R/musicgeneration_amd/libmgx.so; else lib=$R/tools/ab/libmgx_$n.so; fi
  MGX_LIB=$lib timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_ab_$n -- python3 $R/tools/attn_bench.py $args > $R/gpurun_out/kt_ab_$n.log 2>&1 || { tail -5 $R/gpurun_out/kt_ab_$n.log; exit 1; }
  echo "== $n"; python3 $R/tools/profsum.py $R/gpurun_out/kt_ab_$n 6 | grep "$pat" | cut -c1-60,100-175
done
