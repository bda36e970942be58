#!/bin/bash
# gpurun -- bash tools/probe/run_winprobe2.sh ["PIPER_HIP_WIN_CFG=4,1,4" …]: phase trace of the stage-0 ResBlock conv launches (medium voice, factor 8)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/win
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_WIN_TRACE -x hip tools/probe/winprobe2.cpp \
  piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o $O/winprobe2 2> $O/build2.log
export PIPER_HIP_TUNING=1
for cfg in "${@:-X=1}"; do
  echo "== $cfg"
  ( export $cfg
    timeout -k 5 60 $O/winprobe2 128 2688 3,5,7 1,1,1
    timeout -k 5 60 $O/winprobe2 128 2688 3,5,7 2,6,12 )
done
