#!/bin/bash
# usage (GPU box, via gpurun): bash tools/prof_all.sh <round tag>   -- rocprofv3 --kernel-trace --stats of every bench
# workload; summaries (tools/profsum.py) and bench lines land in gpurun_out/ under the names profiles/ uses.
set -o pipefail
tag=${1:-r01c}
R=$GRAFT_REPO_ROOT
for wl in train_unified train_separated train_relative decode_unified; do
  extra="--steps 6 --warmup 3"
  [ $wl = decode_unified ] && extra="--seq 512"
  [ $wl = train_unified ] && extra="$extra --no-decode --no-other-configs"
  bash $R/tools/prof.sh ${tag}_$wl --workload $wl $extra > $R/gpurun_out/${tag}_${wl}_bench_line.json || { tail -20 $R/gpurun_out/prof_${tag}_$wl.log; exit 1; }
  python3 $R/tools/profsum.py $R/gpurun_out/prof_${tag}_$wl > $R/gpurun_out/${tag}_${wl}_summary.txt
  cp $(ls $R/gpurun_out/prof_${tag}_$wl/*/*kernel_stats.csv | tail -1) $R/gpurun_out/${tag}_${wl}_kernel_stats.csv
  head -4 $R/gpurun_out/${tag}_${wl}_summary.txt | cut -c1-150
done
# unprofiled lines of the same tree on the same box
cd $R
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_train_unified_bench_line_unprofiled.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload decode_unified > gpurun_out/${tag}_decode_unified_bench_line_unprofiled.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload train_separated --no-cpu-baseline > gpurun_out/${tag}_train_separated_bench_line_unprofiled.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload train_relative --no-cpu-baseline > gpurun_out/${tag}_train_relative_bench_line_unprofiled.json 2>> gpurun_out/${tag}_bench.err || exit 1
cut -c1-140 gpurun_out/${tag}_*_unprofiled.json
