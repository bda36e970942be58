export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/cab
for m in kernel dma kernel dma; do
  if [ $m = dma ]; then export PIPER_HIP_COLLECT_DMA=1; else unset PIPER_HIP_COLLECT_DMA; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-scale-bench --steps 200 --warmup 20 > gpurun_out/cab/$m.json 2>gpurun_out/cab/$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/cab/$m.json")); print("$m", d["ms_per_step"], d["gpu_ms_mean"], d["end_to_end_ms"])
PY
done
timeout -k 10 300 python -m pytest tests/test_gpu_voice.py -x -q -m gpu 2>&1 | tail -2
