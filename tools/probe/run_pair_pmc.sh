#!/bin/bash
# PMC passes over pairprobe (gpurun -- bash tools/probe/run_pair_pmc.sh)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
export TMPDIR=/tmp
O=gpurun_out/pairpmc
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -x hip tools/probe/pairprobe.cpp \
  piper-swift_amd/csrc/rb_pair.hip piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o $O/pairprobe
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
i=0
for shape in "32 86016 3,5,7 1,2,3 2,6,12" "64 21504 3,5,7 1,2,3 2,6,12"; do
  $O/pairprobe $shape >> $O/timing.txt
  p=1
  for pmc in "$P1" "$P2" "$P3"; do
    d=$O/s${i}_p$p
    timeout -k 10 120 rocprofv3 --pmc $pmc --output-format csv -d $d -- $O/pairprobe $shape > $d.log 2>&1 || echo "pass $p failed" >> $O/timing.txt
    p=$((p+1))
  done
  i=$((i+1))
done
cat $O/timing.txt
