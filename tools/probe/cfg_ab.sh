export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
for rep in 1 2 3; do for cfg in "--factor 8" "--factor 64" "--factor 8 --batch 8" "--factor 16" "--quality high --factor 2"; do
  echo "new [$cfg]: $(timeout -k 10 120 python tools/profile_steps.py $cfg 2>&1 | head -1)"
  echo "old [$cfg]: $(PIPER_HIP_WIN_KS_ONE_WAVE=1 timeout -k 10 120 python tools/profile_steps.py $cfg 2>&1 | head -1)"
done; done
