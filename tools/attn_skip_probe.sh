#!/bin/bash
# Timing-only builds of the third-generation attention sweep with one stream of the tile loop left out
# (FP_V3_SKIP mask, see attn_fwd.hip).  Build here (CPU box):   tools/attn_skip_probe.sh build "0 1 2 4 8 16 32 48 3"
# Run on the GPU box:                                           tools/attn_skip_probe.sh run   "0 1 2 4 8 16 32 48 3"
set -e
cd "$(dirname "$0")/.."
CS=framepack_amd/csrc
mkdir -p $CS/probe
if [ "$1" = build ]; then
  FLAGS=$(sed -n 's/^FLAGS_attn_fwd = //p' $CS/Makefile)
  for m in $2; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -DFP_V3_SKIP=$m -DFP_V3_DIAG=${DIAG:-1} \
        -c $CS/attn_fwd.hip -o $CS/probe/attn_skip$m.o
      # the compiler must not touch AGPRs (they are addressed by number from inline asm): check the assembly
      /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $FLAGS -DFP_V3_SKIP=$m -DFP_V3_DIAG=${DIAG:-1} -S --cuda-device-only \
        $CS/attn_fwd.hip -o $CS/probe/attn_skip$m.s 2>/dev/null
      n=$(awk '/^_ZN.*attn_v3_kernel/{k=1} /amdhsa_kernel.*attn_v3/{k=0} /ASMSTART/{a=1} /ASMEND/{a=0} { if(k && !a && $0 ~ /v_accvgpr|scratch_/) c++ } END{print c+0}' $CS/probe/attn_skip$m.s)
      rm -f $CS/probe/attn_skip$m.s
      if [ "$n" != 0 ]; then echo "FP_V3_SKIP=$m: $n compiler-generated AGPR / scratch instructions - NOT usable"; rm -f $CS/probe/attn_skip$m.o; fi ) &
  done
  wait
  for m in $2; do
    rm -f $CS/probe/libfp_skip$m.so
    [ -f $CS/probe/attn_skip$m.o ] || continue
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $CS/capi.o $CS/norm_kernels.o $CS/probe/attn_skip$m.o $CS/attn_fp8.o \
      $CS/gemm_bf16.o $CS/gemm_fp8.o $CS/vae_kernels.o -o $CS/probe/libfp_skip$m.so
  done
else
  for m in $2; do
    echo "== FP_V3_SKIP=$m"
    [ -f $PWD/$CS/probe/libfp_skip$m.so ] || { echo "(not built)"; continue; }
    FRAMEPACK_HIP_LIB=$PWD/$CS/probe/libfp_skip$m.so REPEAT=1 timeout -k 10 100 python tools/bench_attn.py ${3:-19385} ${4:-24} 10
  done
fi
