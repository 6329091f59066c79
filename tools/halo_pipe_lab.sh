#!/bin/bash
# Compile-time ablations of csrc/conv3x3_pipe_i8.hip: one A/B library per variant under build_ab/ (the product objects + the pipe kernel built with
# -DDLMCQ_PIPE_ABL=bits: 1 no weight DMA, 2 no halo DMA, 4 no epilogue quads, 8 no barriers, 16 no fragment reads, 32 no MFMAs; results are
# garbage, only the time means something).  usage: tools/halo_pipe_lab.sh build "0 4 7 15 31 47"   (here, no GPU)  |  tools/halo_pipe_lab.sh run "0 4 ..." (GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/dlmc-quant_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I$ROOT/include -Wall -Wno-unused-function -fno-slp-vectorize"
if [ "$1" = build ]; then
  mkdir -p $ROOT/build_ab
  OBJS=$(for f in fake_quant observer pack fq_backward rootq weight_fold conv_i8 conv3x3_i8 conv_chain_i8 conv_dw_i8 conv_dwm_i8 conv_dwpw_i8 conv_pw_i8 conv_pwr_i8 conv_stem_i8 conv_stem_pool7_i8 estimator adaround api; do echo $CS/build/$f.o; done)
  for v in $2; do
    ( /opt/rocm/bin/hipcc $FLAGS -DDLMCQ_PIPE_ABL=$v -DDLMCQ_PIPE_STAMP -c $CS/conv3x3_pipe_i8.hip -o /tmp/pipe_abl_$v.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/pipe_abl_$v.o -o $ROOT/build_ab/libdlmcq_pipe_abl$v.so ) &
  done
  wait
  ls -la $ROOT/build_ab/*.so
else
  for v in $2; do
    echo -n "ABL=$v: "; DLMCQ_LAB_TOOLS=1 DLMCQ_LIBRARY=$ROOT/build_ab/libdlmcq_pipe_abl$v.so python3 $ROOT/tools/halo_pipe_ab.py --stamps 2>&1 | grep "C256 14\|stamps" | head -4
  done
fi
