#!/bin/bash
# usage (GPU box): bash tools/pmc_attn.sh <tag> "<script args>"  -- kernel-trace pass + one SQ counter pass over tools/attn_bench.py
tag=$1; args=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/tools/attn_bench.py $args > $R/gpurun_out/kt_$tag.log 2>&1 || { tail -5 $R/gpurun_out/kt_$tag.log; exit 1; }
python3 $R/tools/profsum.py $R/gpurun_out/kt_$tag 8 | cut -c1-70,100-175
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/attn_bench.py $args > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmcsum.py $R/gpurun_out/pmc_$tag
