#!/bin/bash
# builds and runs the conv_pipe ablation probes on the GPU box (gpurun -- bash tools/probe/run_pipeprobe.sh)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/probe
for abl in ${ABLS:-0 1 2 4 8 16 3 19 31}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPH_PIPE_ABL=$abl -x hip tools/probe/pipeprobe.cpp \
    piper-swift_amd/csrc/conv_pipe.hip piper-swift_amd/csrc/context.cpp -o gpurun_out/probe/pipeprobe_$abl 2> gpurun_out/probe/build_$abl.log &
done
wait
for abl in ${ABLS:-0 1 2 4 8 16 3 19 31}; do
  for shape in "64 21504 3,5,7 1,2,3" "128 21504 3,7,11 1,3,5" "32 86016 3,5,7 2,6,12" "256 2688 3,7,11 1,3,5"; do
    timeout -k 5 60 gpurun_out/probe/pipeprobe_$abl $shape
  done
done
