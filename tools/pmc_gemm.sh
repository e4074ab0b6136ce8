#!/bin/bash
# PMC passes over one GEMM shape (ours and hipBLASLt's kernel side by side). usage: tools/pmc_gemm.sh <outdir> M N K
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/tools/gemm_one.py "$@"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $R/$OUT/p2 -- python3 $R/tools/gemm_one.py "$@"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $R/$OUT/p3 -- python3 $R/tools/gemm_one.py "$@"
