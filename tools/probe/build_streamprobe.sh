#!/bin/bash
# builds tools/probe/bin/streamprobe with the phase stamps compiled in (minutes: every tap instantiation); the binary travels with gpurun
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
mkdir -p tools/probe/bin /tmp/sp_obj
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_STREAM_TRACE"
for f in conv conv_short conv_inst_k1 conv_inst_k2 conv_inst_k3 conv_inst_k5 conv_inst_k7 conv_inst_k11 conv_win conv_pipe; do
  /opt/rocm/bin/hipcc $F -c -x hip piper-swift_amd/csrc/$f.hip -o /tmp/sp_obj/$f.o &
done
/opt/rocm/bin/hipcc $F -c -x hip piper-swift_amd/csrc/context.cpp -o /tmp/sp_obj/context.o &
/opt/rocm/bin/hipcc $F -c -x hip tools/probe/streamprobe.cpp -o /tmp/sp_obj/streamprobe.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/sp_obj/*.o -o tools/probe/bin/streamprobe
ls -la tools/probe/bin/streamprobe
