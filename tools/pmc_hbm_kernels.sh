#!/bin/bash
# HBM-side bytes (FETCH_SIZE / WRITE_SIZE, one pass each) of every kernel of one denoising step + one VAE decode.
# usage: tools/pmc_hbm_kernels.sh <outdir under gpurun_out>
set -e
OUT=$1; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null
