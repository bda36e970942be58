export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
for rep in 1 2; do for m in 2048 4096 8192 1000000; do
  for cfg in "--factor 64" "--factor 8 --batch 8" "--factor 8 --batch 16" "--quality high --factor 32"; do
    echo "below=$m [$cfg]: $(PIPER_HIP_TM16_BELOW=$m timeout -k 10 120 python tools/profile_steps.py $cfg 2>&1 | head -1)"
  done
done; done
