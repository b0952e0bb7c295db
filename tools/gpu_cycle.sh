#!/bin/bash
# usage (GPU box, via gpurun): bash tools/gpu_cycle.sh   -- gpu tests, smoke, headline bench; stops at the first failure
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python3 -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 400 python3 -u bench.py > gpurun_out/bench_train_unified.log 2>&1 || { tail -30 gpurun_out/bench_train_unified.log; exit 1; }
grep '^{' gpurun_out/bench_train_unified.log
