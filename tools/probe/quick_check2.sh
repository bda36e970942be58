# correctness (op-level + voice parity), then per-launch tables incl. the high voice; stops at the first failure
set -e
mkdir -p gpurun_out/qc
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_voice.py -x -q -m gpu > gpurun_out/qc/tests.txt 2>&1 || { tail -30 gpurun_out/qc/tests.txt; exit 1; }
tail -1 gpurun_out/qc/tests.txt
for f in 8 64; do
  timeout -k 10 120 python tools/profile_steps.py --factor $f > gpurun_out/qc/steps_f$f.txt 2>&1
  grep -q "Memory access fault" gpurun_out/qc/steps_f$f.txt && exit 1
  head -1 gpurun_out/qc/steps_f$f.txt; grep convT gpurun_out/qc/steps_f$f.txt
done
timeout -k 10 120 python tools/profile_steps.py --factor 8 --quality high > gpurun_out/qc/steps_high.txt 2>&1
head -1 gpurun_out/qc/steps_high.txt; grep convT gpurun_out/qc/steps_high.txt
