#!/bin/bash
# After `gpurun -- bash tools/run_profiles.sh <tag> && bash tools/run_pmc_mfma.sh <tag>`: turn gpurun_out/<tag> into the files committed under profiles/.
set -e
TAG=${1:-r2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
R=gpurun_out/$TAG
python tools/summarize_prof.py $TAG $R/f32_stats $R/f32_fetch $R/f32_write > /dev/null
python tools/summarize_prof.py ${TAG}_high_bf16 $R/high_bf16_stats $R/high_bf16_fetch $R/high_bf16_write > /dev/null
python tools/summarize_prof.py ${TAG}_factor64 $R/f64_stats > /dev/null
python tools/summarize_prof.py ${TAG}_high_f32 $R/high_f32_stats > /dev/null
python tools/summarize_mfma.py $TAG $R/f32_mfma $R/f32_wave "medium fp32 factor 8" > /dev/null
python tools/summarize_mfma.py ${TAG}_high_bf16 $R/high_bf16_mfma $R/high_bf16_wave "high bf16 factor 8" > /dev/null
for f in bench_factor64 bench_high_bf16 bench_high_f32 bench_medium_bf16; do cp $R/$f.json profiles/${TAG}_$f.json; done
cp $R/bench_full.json profiles/${TAG}_bench_full_with_cpu_baseline.json
for f in steps_factor64 steps_factor8 steps_factor8_batch8 steps_high_bf16 steps_high_f32 steps_medium_bf16; do cp $R/$f.txt profiles/${TAG}_$f.txt; done
mkdir -p profiles/${TAG}_raw
for c in f32 f64 high_bf16 high_f32; do cp "$(ls -t $R/${c}_stats/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_raw/${c}_kernel_stats.csv; done
COMMIT=$(git rev-parse --short HEAD)
python - "$COMMIT" <<PY2
import json, sys
for f in ["${TAG}", "${TAG}_high_bf16", "${TAG}_factor64", "${TAG}_high_f32"]:
    p = "profiles/" + f + "_rocprof_summary.json"
    d = json.load(open(p))
    d["commit"] = sys.argv[1]  # the tree the passes were taken on: bench.py prints it as roofline.traffic_profile_commit
    json.dump(d, open(p, "w"), indent=1)
PY2
python - <<PY
import json
for f in ["bench_full_with_cpu_baseline","bench_factor64","bench_high_f32","bench_high_bf16","bench_medium_bf16"]:
    d=json.load(open("profiles/${TAG}_"+f+".json"))
    print(f, d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"])
    for k in d.get("roofline_by_kernel",[]): print("   ", k['kernel'][:62], k['launches'], k['avg_launch_us'], k['achieved'], k['frac'])
PY
