export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/tm
for v in default 0 32 64 128 512 1024; do
  if [ $v = default ]; then unset PIPER_HIP_TM16_BELOW; else export PIPER_HIP_TM16_BELOW=$v; fi
  timeout -k 10 120 python tools/profile_steps.py --factor 8 > gpurun_out/tm/f8_$v.txt 2>&1
  grep -q "Memory access fault" gpurun_out/tm/f8_$v.txt && exit 1
  head -1 gpurun_out/tm/f8_$v.txt
done
for v in default 0 64 1024; do
  if [ $v = default ]; then unset PIPER_HIP_TM16_BELOW; else export PIPER_HIP_TM16_BELOW=$v; fi
  timeout -k 10 120 python tools/profile_steps.py --factor 1 > gpurun_out/tm/f1_$v.txt 2>&1
  head -1 gpurun_out/tm/f1_$v.txt
done
