#!/bin/bash
# PMC passes over the conv probes (gpurun -- bash tools/probe/run_pmc.sh): where do the wave cycles of the long-row convs go?
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
export TMPDIR=/tmp
O=gpurun_out/pmc
mkdir -p $O
for abl in ${ABLS:-0 31}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_PIPE_ABL=$abl -x hip tools/probe/pipeprobe.cpp \
    piper-swift_amd/csrc/conv_pipe.hip piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o $O/probe_$abl 2> $O/build_$abl.log &
done
wait
SHAPES=("32 86016 3,5,7 2,6,12" "64 21504 3,5,7 1,2,3" "128 21504 3,7,11 1,3,5")
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P3="GRBM_GUI_ACTIVE"
i=0
for shape in "${SHAPES[@]}"; do
  for abl in ${ABLS:-0 31}; do
    for kind in pipe win; do
      [ "$kind" = win ] && [ "$abl" != 0 ] && continue
      $O/probe_$abl $shape 1 $kind >> $O/timing.txt
      p=1
      for pmc in "$P1" "$P2" "$P3"; do
        d=$O/s${i}_${kind}_abl${abl}_p$p
        timeout -k 10 120 rocprofv3 --pmc $pmc --output-format csv -d $d -- $O/probe_$abl $shape 1 $kind > $d.log 2>&1
        p=$((p+1))
      done
    done
  done
  i=$((i+1))
done
cat $O/timing.txt
