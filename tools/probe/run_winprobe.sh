#!/bin/bash
# gpurun -- bash tools/probe/run_winprobe.sh : phase trace of conv_win_kernel on the three ConvTranspose geometries of the medium voice at factor 8
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/win
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_WIN_TRACE -x hip tools/probe/winprobe.cpp \
  piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o $O/winprobe 2> $O/build.log
export PIPER_HIP_TUNING=1
for cfg in "${@:-default}"; do
  [ "$cfg" != default ] && export $cfg
  echo "== $cfg"
  timeout -k 5 60 $O/winprobe 256 128 16 8 336
  timeout -k 5 60 $O/winprobe 128 64 16 8 2688 avg
  timeout -k 5 60 $O/winprobe 64 32 8 4 21504 avg
done
