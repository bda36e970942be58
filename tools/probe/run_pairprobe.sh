#!/bin/bash
# gpurun -- bash tools/probe/run_pairprobe.sh   (VARIANTS="0 1 2 3" to A/B the inner-loop variants)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/pair
mkdir -p $O
for v in ${VARIANTS:-0}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_PAIR_VARIANT=$v ${TRACE:+-DPH_PAIR_TRACE} ${EXTRA} -x hip tools/probe/pairprobe.cpp \
    piper-swift_amd/csrc/rb_pair.hip piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o $O/pairprobe_$v 2> $O/build_$v.log &
done
wait
for v in ${VARIANTS:-0}; do
  timeout -k 5 120 $O/pairprobe_$v 64 452 3,5,7 1,2,3 2,6,12 check
  timeout -k 5 120 $O/pairprobe_$v 32 452 3,5,7 1,2,3 2,6,12 check
  timeout -k 5 60 $O/pairprobe_$v 32 86016 3,5,7 1,2,3 2,6,12
  timeout -k 5 60 $O/pairprobe_$v 64 21504 3,5,7 1,2,3 2,6,12
done
