export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/wm
for m in 1 2 4; do
  for f in 1 8; do
    PIPER_HIP_KS_WANT_MUL=$m timeout -k 10 120 python tools/profile_steps.py --factor $f > gpurun_out/wm/f${f}_m$m.txt 2>&1
    grep -q "Memory access fault" gpurun_out/wm/f${f}_m$m.txt && exit 1
    echo "mul=$m: $(head -1 gpurun_out/wm/f${f}_m$m.txt)"
  done
  grep -h "enc1\.\|flow3.wn1\|flow3.pre\|ln2_proj\|conv_pre" gpurun_out/wm/f8_m$m.txt
done
