#!/bin/bash
# dev helper: builds a variant of the engine into meltingpot_amd/lib/libmp_engine_<tag>.so
# usage: tools/ab_build.sh <tag> [-DFLAG ...]
cd "$(dirname "$0")/.."; tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o meltingpot_amd/lib/libmp_engine_$tag.so \
  meltingpot_amd/csrc/mp_engine.hip meltingpot_amd/csrc/step_kernels.hip meltingpot_amd/csrc/frame.hip && \
  rm -f meltingpot_amd/lib/*.hipv4* meltingpot_amd/lib/*host-x86* meltingpot_amd/lib/*.hipfb && ls -la meltingpot_amd/lib/
