export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/w32
for m in 1 2 4; do
  PIPER_HIP_KS_WANT_MUL32=$m timeout -k 10 120 python tools/profile_steps.py --factor 64 > gpurun_out/w32/f64_m$m.txt 2>&1
  grep -q "Memory access fault" gpurun_out/w32/f64_m$m.txt && exit 1
  echo "mul32=$m: $(head -1 gpurun_out/w32/f64_m$m.txt)"
  grep -h "enc1\.\|flow3.wn1\|flow3.pre\|ln2_proj\|conv_pre" gpurun_out/w32/f64_m$m.txt
  PIPER_HIP_KS_WANT_MUL32=$m timeout -k 10 120 python tools/profile_steps.py --factor 8 --batch 8 > gpurun_out/w32/b8_m$m.txt 2>&1
  echo "mul32=$m: $(head -1 gpurun_out/w32/b8_m$m.txt)"
  PIPER_HIP_KS_WANT_MUL32=$m timeout -k 10 120 python tools/profile_steps.py --factor 16 > gpurun_out/w32/f16_m$m.txt 2>&1
  echo "mul32=$m: $(head -1 gpurun_out/w32/f16_m$m.txt)"
done
