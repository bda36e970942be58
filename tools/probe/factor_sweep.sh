#!/bin/bash
# gpurun -- bash tools/probe/factor_sweep.sh : ms/utterance (GPU time) over scale factors, shipped kernel choice vs the lean kernels / consumer-side LayerNorm switched off
export PIPER_HIP_TUNING=1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
one() { python bench.py --factor $1 --no-cpu-baseline --no-scale-bench --steps 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f %.4f' % (j['ms_per_step'], j['gpu_ms_mean']))"; }
for f in ${FACTORS:-12 16 24 32 48}; do
  a=$(one $f); b=$(PIPER_HIP_NO_LEAN=1 one $f); c=$(PIPER_HIP_NO_LN_SELF=1 one $f)
  echo "factor $f: shipped $a | NO_LEAN $b | NO_LN_SELF $c"
done
