#!/bin/bash
# PMC passes over the attention micro-benchmark (one variant per run). usage: tools/pmc_attn.sh <variant> <outdir>
set -e
V=$1; OUT=$2; mkdir -p $OUT
export REPEAT=1
export ATTN_VARIANT=$V
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/tools/bench_attn.py 19385 24 3
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $R/$OUT/p2 -- python3 $R/tools/bench_attn.py 19385 24 3
