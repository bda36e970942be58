export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
for rep in 1 2; do for w in 1 2 4 8 16 64; do
  for cfg in "--quality high --precision bf16" "--quality medium --precision bf16" "--quality high --precision bf16 --factor 32"; do
    echo "want=$w [$cfg]: $(PIPER_HIP_BF16_WANT_BLOCKS=$w timeout -k 10 120 python tools/profile_steps.py $cfg 2>&1 | head -1)"
  done
done; done
