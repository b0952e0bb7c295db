#!/bin/bash
# usage (GPU box): bash tools/prof_attn.sh [B S H dh]  -- per-kernel times of the attention forward + backward
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pa -- python3 $R/tools/attn_bench.py ${1:-16} ${2:-2048} ${3:-8} ${4:-128} 10 both > $R/gpurun_out/pa.log 2>&1 || { tail -5 $R/gpurun_out/pa.log; exit 1; }
f=$(find $R/gpurun_out/pa -name '*kernel_stats.csv' | sort | sed -n 1p)
python3 $R/tools/profsum.py $f | cut -c1-150 | sed -n 1,12p
