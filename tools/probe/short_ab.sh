#!/bin/bash
# A/B of the short-row conv kernel choices: rocprofv3 kernel-trace of the default bench step (graph replay: true in-graph kernel durations)
# under each environment setting; prints ms/step and the conv kernels' per-instantiation averages.  gpurun -- bash tools/probe/short_ab.sh "A=1" "B=2 C=3" …
export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/short_ab
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  d="$OUT/c$i"; rm -rf "$d"
  echo "== [$i] $cfg"
  ( export $cfg; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-scale-bench --steps 40 --warmup 5 ${BENCH_ARGS} > "$d.log" 2>&1 )
  python3 - "$d" "$d.log" <<'PY'
import sys,glob,csv,json,re
d,log=sys.argv[1],sys.argv[2]
for line in open(log):
    if line.startswith('{'):
        j=json.loads(line); print("  ms_per_step", j["ms_per_step"], "gpu_ms", j.get("gpu_ms_per_step"))
f=glob.glob(d+"/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n=r["Name"]
    if float(r["TotalDurationNs"]) > 0.004 * tot:
        n=re.sub(r"void ph::detail::|\(ph::ConvArgs.*","",n)
        print("  %-60s calls %5s avg %7.2f us  %5.1f%%"%(n[:60], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
done
