# correctness first (op-level conv + voice parity), then the per-launch tables; stops at the first failure
set -e
mkdir -p gpurun_out/qc
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_voice.py -x -q -m gpu > gpurun_out/qc/tests.txt 2>&1 || { tail -30 gpurun_out/qc/tests.txt; exit 1; }
tail -3 gpurun_out/qc/tests.txt
for f in 1 8; do
  timeout -k 10 120 python tools/profile_steps.py --factor $f > gpurun_out/qc/steps_f$f.txt 2>&1
  grep -q "Memory access fault" gpurun_out/qc/steps_f$f.txt && exit 1
  head -1 gpurun_out/qc/steps_f$f.txt
done
timeout -k 10 120 python tools/profile_steps.py --factor 8 --batch 8 > gpurun_out/qc/steps_f8b8.txt 2>&1
head -1 gpurun_out/qc/steps_f8b8.txt
timeout -k 10 120 python tools/profile_steps.py --factor 64 > gpurun_out/qc/steps_f64.txt 2>&1
head -1 gpurun_out/qc/steps_f64.txt
