#!/bin/bash
# usage (GPU box): bash tools/traffic.sh   -- HBM traffic of the roofline kernels from the TCC counters.
# FETCH_SIZE and WRITE_SIZE do not fit one pass (3 + 2 of the 4 TCC slots): one rocprofv3 run per counter.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {  # tag counter script args...
  tag=$1; ctr=$2; shift 2
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/tr_$tag -- python3 "$@" > $R/gpurun_out/tr_$tag.log 2>&1 || { tail -5 $R/gpurun_out/tr_$tag.log; exit 1; }
}
run attn_fetch FETCH_SIZE $R/tools/attn_bench.py 16 2048 8 128 3 fwd && \
run attn_write WRITE_SIZE $R/tools/attn_bench.py 16 2048 8 128 3 fwd && \
run attnb_fetch FETCH_SIZE $R/tools/attn_bench.py 16 2048 8 128 2 bwd && \
run attnb_write WRITE_SIZE $R/tools/attn_bench.py 16 2048 8 128 2 bwd && \
run dec_fetch FETCH_SIZE $R/tools/decode_one.py 2112 5 && \
run dec_write WRITE_SIZE $R/tools/decode_one.py 2112 5 && \
run tn_fetch FETCH_SIZE $R/tools/gemm_tn_step.py 2 && \
run tn_write WRITE_SIZE $R/tools/gemm_tn_step.py 2 && \
python3 $R/tools/traffic_sum.py $R/gpurun_out
