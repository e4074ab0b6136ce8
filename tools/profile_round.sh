#!/bin/bash
# The round's evidence run (on the GPU box): kernel-trace stats of bench.py, then HBM-traffic counters of the
# dominant kernel in passes of their own.  usage: tools/profile_round.sh <outdir under gpurun_out>
set -e
OUT=$1; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json
export REPEAT=1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/fetch -- python3 $R/tools/bench_attn.py 19385 24 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/write -- python3 $R/tools/bench_attn.py 19385 24 3
