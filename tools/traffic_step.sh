#!/bin/bash
# usage (GPU box): bash tools/traffic_step.sh [tag]   -- HBM traffic of the roofline kernels from the TCC counters, round 2.
# The MFMA kernels are counted inside the bench's own training step (one warm-up + one timed step, the step's real
# mix of shapes); decode attention at context 2112 through tools/decode_one.py.  FETCH_SIZE and WRITE_SIZE do not fit
# one pass (3 + 2 of the 4 TCC slots): one rocprofv3 run per counter, no tracing options beside --pmc.
R=$GRAFT_REPO_ROOT
TAG=${1:-r02c}
cd /tmp && export TMPDIR=/tmp
run() {  # tag counter script args...
  tag=$1; ctr=$2; shift 2
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/tr2_$tag -- python3 "$@" > $R/gpurun_out/tr2_$tag.log 2>&1 || { tail -5 $R/gpurun_out/tr2_$tag.log; exit 1; }
  echo "pass $tag done"
}
STEP="$R/bench.py --workload train_unified --steps 1 --warmup 1 --no-decode --no-other-configs --no-cpu-baseline"
run step_fetch FETCH_SIZE $STEP && \
run step_write WRITE_SIZE $STEP && \
run dec_fetch FETCH_SIZE $R/tools/decode_one.py 2112 5 && \
run dec_write WRITE_SIZE $R/tools/decode_one.py 2112 5 && \
python3 $R/tools/traffic_step_sum.py $R/gpurun_out > $R/gpurun_out/${TAG}_traffic_pmc.json && \
for t in step_fetch step_write dec_fetch dec_write; do
  f=$(find $R/gpurun_out/tr2_$t -name '*counter_collection.csv' | sort | sed -n 1p)
  python3 $R/tools/pmc_compact.py $f > $R/gpurun_out/${TAG}_traffic_${t}_per_kernel.csv
done && cat $R/gpurun_out/${TAG}_traffic_pmc.json
