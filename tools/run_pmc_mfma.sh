#!/bin/bash
# MFMA-busy evidence (north-star: "MFMA-busy against gfx950 peak"): separate PMC passes, kernel-trace not combined.
set -e
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-scale-bench --no-profile --steps 40 --warmup 5"
pass() {  # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${name}" -- python3 "$ROOT/bench.py" "$@" > "$OUT/${name}.log" 2>&1
  echo "pmc $name done"
}
pass f32_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $COMMON
pass f32_wave "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" $COMMON
pass high_bf16_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --quality high --precision bf16 $COMMON
pass high_bf16_wave "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" --quality high --precision bf16 $COMMON
find "$OUT" -name "*.db" -delete 2>/dev/null || true
