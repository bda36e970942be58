#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box: gpurun -- bash tools/run_profiles.sh <tag>).
# rocprofv3 passes are separate runs: kernel-trace/stats, then one PMC counter per pass (MI355X_MICROARCH.md §HBM).
set -e
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_stats" -- python3 "$ROOT/bench.py" "$@" > "$OUT/${name}_stats.log" 2>&1
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${name}_fetch" -- python3 "$ROOT/bench.py" "$@" > "$OUT/${name}_fetch.log" 2>&1
  timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${name}_write" -- python3 "$ROOT/bench.py" "$@" > "$OUT/${name}_write.log" 2>&1
  echo "profiled $name"
}
COMMON="--no-cpu-baseline --no-scale-bench --steps 40 --warmup 5"
prof f32 $COMMON
prof high_bf16 --quality high --precision bf16 $COMMON
# kernel-trace only for the two configurations VERDICT r1 #4 names (conv_win / rb_pair where they dominate)
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f64_stats" -- python3 "$ROOT/bench.py" --factor 64 --steps 20 --no-cpu-baseline --no-scale-bench > "$OUT/f64_stats.log" 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/high_f32_stats" -- python3 "$ROOT/bench.py" --quality high --steps 20 --no-cpu-baseline --no-scale-bench > "$OUT/high_f32_stats.log" 2>&1
cd "$ROOT"
timeout -k 10 300 python3 bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
echo "bench full done"
timeout -k 10 200 python3 bench.py --quality high --precision bf16 --no-cpu-baseline > "$OUT/bench_high_bf16.json" 2> "$OUT/bench_high_bf16.err"
timeout -k 10 200 python3 bench.py --quality medium --precision bf16 --no-cpu-baseline > "$OUT/bench_medium_bf16.json" 2>> "$OUT/bench_high_bf16.err"
timeout -k 10 200 python3 bench.py --quality high --no-cpu-baseline --no-scale-bench > "$OUT/bench_high_f32.json" 2>> "$OUT/bench_high_bf16.err"
timeout -k 10 200 python3 bench.py --factor 64 --no-cpu-baseline --no-scale-bench --steps 20 > "$OUT/bench_factor64.json" 2>> "$OUT/bench_high_bf16.err"
echo "bench variants done"
timeout -k 10 100 python3 tools/profile_steps.py > "$OUT/steps_factor8.txt"
timeout -k 10 100 python3 tools/profile_steps.py --factor 64 > "$OUT/steps_factor64.txt"
timeout -k 10 100 python3 tools/profile_steps.py --quality high > "$OUT/steps_high_f32.txt"
timeout -k 10 100 python3 tools/profile_steps.py --batch 8 > "$OUT/steps_factor8_batch8.txt"
timeout -k 10 100 python3 tools/profile_steps.py --quality high --precision bf16 > "$OUT/steps_high_bf16.txt"
timeout -k 10 100 python3 tools/profile_steps.py --quality medium --precision bf16 > "$OUT/steps_medium_bf16.txt"
# keep the merged output small: only the csv summaries travel back
find "$OUT" -name "*.db" -delete 2>/dev/null || true
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null || true  # the per-dispatch rows: only the stats / counter tables are summarised (gpurun merges ≤ 64 MiB back)
du -sh "$OUT"
