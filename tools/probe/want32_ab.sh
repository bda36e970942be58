export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/w32
for rep in 1 2; do for m in 1 2; do
  for cfg in "--factor 8" "--factor 8 --batch 8" "--quality high" "--factor 32" "--factor 64"; do
    echo "mul32=$m [$cfg]: $(PIPER_HIP_KS_WANT_MUL32=$m timeout -k 10 120 python tools/profile_steps.py $cfg 2>&1 | head -1)"
  done
done; done
